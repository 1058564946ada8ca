#!/bin/bash
out=gpurun_out/r4b; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ticket.py tests/test_const_long.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -15 $out/pytest.log
Q="--no-cpu --no-host --no-extras --series long --pairs 1024 --steps 3 --warmup 1 --verify 2"
for rep in 1 2; do
  for wg in 1 0; do
    GNX_CL_WG=$wg timeout 600 python bench.py $Q 2>>$out/bench.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('WG=$wg', 'step %.2f ms' % d['ms_per_step'], 'sweep %.2f ms' % d['roofline']['avg_launch_ms'], 'frac %.3f' % d['roofline']['frac'], '%.4e' % d['value'], d['bit_exact_sample'])" | tee -a $out/ab.log
  done
done
GNX_CL_WG=1 timeout 600 python bench.py --no-cpu --no-host --no-extras --series long --pairs 2048 --steps 2 --warmup 1 --verify 2 2>>$out/bench.err | cut -c1-300 | tee -a $out/ab.log
tail -3 $out/bench.err
