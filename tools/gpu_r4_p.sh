#!/bin/bash
out=gpurun_out/r4p; mkdir -p $out
( time timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4p/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'ok', d.get('bit_exact_sample'), 'failed', d.get('extras_failed'))
print('gsw', d.get('gsw_reads'))
for k in ('north_star_1M','c3','c3_10M','c5'):
    print(k, {a:b for a,b in d.get(k,{}).items() if a in ('value','leg_wall_s','error','bit_exact_sample')})
print('roofline', d['roofline']['frac'], 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
tail -3 $out/bench.err
