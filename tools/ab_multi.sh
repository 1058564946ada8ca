#!/bin/bash
# Same-box comparison of several builds on tools/bench_shapes.py shapes: bash tools/ab_multi.sh <kind> "<lib1.so lib2.so ...>" <n,m,pairs> [...]   ("-" = the in-tree build)
kind=$1; libs=$2; shift 2
for shape in "$@"; do
  for rep in 1 2; do
    for lib in $libs; do
      if [ $lib = - ]; then unset GNX_LIB_PATH; else export GNX_LIB_PATH=$PWD/$lib; fi
      python tools/bench_shapes.py $kind $shape 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); v = d['default']; print('%-22s' % '$lib', '$kind', '$shape', 'default: %.3f ms (fill %.3f)' % (v['ms'], v['fill_ms']), d.get('same_results'))"
    done
  done
done
