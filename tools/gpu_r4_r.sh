#!/bin/bash
# small batches of short reads: general path vs fast path
out=gpurun_out/r4r; mkdir -p $out
for shp in 150,1000,1 150,1000,16 150,1000,256 150,1000,2048 150,10000,1 150,10000,16 150,10000,128 150,10000,512 150,10000,2048 150,10000,8192 150,3000,64 150,3000,1024; do
  timeout 120 python tools/bench_shapes.py affine $shp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$shp', ' | '.join('%s %.3f ms (path %d)' % (k, v['ms'], v['path']) for k, v in d.items() if isinstance(v, dict)), d.get('same_results'))" | tee -a $out/small.log
done
