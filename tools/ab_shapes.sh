#!/bin/bash
# Same-box A/B of two builds on tools/bench_shapes.py shapes: bash tools/ab_shapes.sh <other.so> <kind> <n,m,pairs> [...]
other=$1; kind=$2; shift 2
for shape in "$@"; do
  for rep in 1 2; do
    for which in A B; do
      if [ $which = B ]; then export GNX_LIB_PATH=$PWD/$other; else unset GNX_LIB_PATH; fi
      python tools/bench_shapes.py $kind $shape 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$which', '$kind', '$shape', ' '.join('%s: %.3f ms (fill %.3f) %.3e' % (k, v['ms'], v['fill_ms'], v['cells_per_s']) for k, v in d.items() if isinstance(v, dict)), d.get('same_results'))"
    done
  done
done
