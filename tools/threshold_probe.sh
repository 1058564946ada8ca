#!/bin/bash
# Where does the fast path start to pay?  general vs default vs forced fast path (row_blocks = GNX_FASTPATH=2) on short windows and on small batches of long reads
for shp in 150,192,400000 150,256,400000 150,384,300000 150,512,200000 150,640,200000 150,768,150000 320,384,100000 320,640,100000 800,512,50000 \
           3200,10000,128 3200,10000,256 3200,10000,400 1600,10000,256 1600,10000,800 800,10000,1024 800,10000,1600 20000,100000,16 20000,100000,32 20000,100000,63; do
  python tools/bench_shapes.py affine $shp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$shp', ' | '.join('%s %.3f ms %.2e' % (k, v['ms'], v['cells_per_s']) for k, v in d.items() if isinstance(v, dict)), d.get('same_results'))"
done
