#!/bin/bash
out=gpurun_out/r4z; mkdir -p $out; rm -f $out/wide.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/parity.log 2>&1; grep -E "passed|failed|^E " $out/parity.log | head -5
GNX_WALK_LANE=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_block or long_reads or fast_path" > $out/parity_lane.log 2>&1; grep -E "passed|failed|^E " $out/parity_lane.log | head -5
for env in "GNX_WALK_WIDE=0" "GNX_WALK_WIDE=1" "GNX_WALK_WIDE=0" "GNX_WALK_WIDE=1"; do
 for shp in 1000,1200,100000 800,10000,12000 250,10000,40000 320,10000,32768; do
  env $env timeout 300 python tools/bench_shapes.py affine $shp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); v=d['default']; print('$env $shp default %.3f ms fill %.2f tb %.2f' % (v['ms'], v['fill_ms'], v['tb_ms']), d.get('same_results'))" | tee -a $out/wide.log
 done
done
timeout 400 python tools/stress.py 150 93 2>&1 | tail -1 | tee -a $out/wide.log
