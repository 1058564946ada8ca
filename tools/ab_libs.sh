#!/bin/bash
# Same-box A/B of two builds of the library: bash tools/ab_libs.sh <other.so> [bench args...]  (the in-tree build is "A", <other.so> is "B";
# alternating runs, kernel time of the dominant kernel and ms per step from bench.py's line)
other=$1; shift
for rep in 1 2 3; do
  for which in A B; do
    if [ $which = B ]; then export GNX_LIB_PATH=$PWD/$other; else unset GNX_LIB_PATH; fi
    python bench.py --no-cpu --no-host --no-extras --verify 0 --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$which', 'step %.3f ms' % d['ms_per_step'], 'dominant %.3f ms' % d['roofline']['avg_launch_ms'], '%.4e' % d['value'])"
  done
done
