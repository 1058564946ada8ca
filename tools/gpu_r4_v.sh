#!/bin/bash
out=gpurun_out/r4v; mkdir -p $out
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('value',d['value'],'frac',r['frac'],'traffic',r.get('traffic'),r.get('traffic_note'))"
bash tools/gpu_r4_u.sh
