#!/bin/bash
out=gpurun_out/r4x; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/parity.log 2>&1; grep -E "passed|failed|^E " $out/parity.log | head
for shp in 1000,1200,100000 800,10000,12000 3200,10000,3000 480,10000,20000; do
  timeout 300 python tools/bench_shapes.py affine $shp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$shp', ' | '.join('%s %.3f ms fill %.2f tb %.2f' % (k, v['ms'], v['fill_ms'], v['tb_ms']) for k, v in d.items() if isinstance(v, dict)), d.get('same_results'))" | tee -a $out/shapes.log
done
GNX_DEBUG=1 timeout 300 python tools/bench_shapes.py affine 1000,1200,100000 2>&1 | grep "gnx fp\] pairs" | tail -8 | cut -c1-200 | tee -a $out/shapes.log
timeout 400 python tools/stress.py 150 91 2>&1 | tail -2 | tee -a $out/shapes.log
