#!/bin/bash
out=gpurun_out/r4w; mkdir -p $out
for sub in 131072 50000 33334 25000 66672 131072 50000; do
  GNX_HOST_SUB=$sub timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sub $sub: host-entry ms', round(d['ms_per_step'],3), 'value %.4g' % d['value'], 'device ms', round(d.get('ms_per_step_device_resident',0),3), d.get('bit_exact_sample'))" | tee -a $out/sub.log
done
