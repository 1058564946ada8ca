#!/bin/bash
out=gpurun_out/r4h; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
tests/cpp/concurrent_pairs_test.bin 16 1000 8 | tee $out/concurrent_pairs.json
tests/cpp/concurrent_pairs_test.bin 4 1000 1 | tee -a $out/concurrent_pairs.json
for pairs in 1024 2048; do
 for rep in 1 2; do
  for sp in 0 3 4; do
    GNX_CL_WALK_SPEC=$sp timeout 600 python bench.py --no-cpu --no-host --no-extras --series long --pairs $pairs --steps 2 --warmup 1 --verify 2 2>>$out/bench.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('pairs $pairs SPEC=$sp', 'step %.2f ms' % d['ms_per_step'], 'sweep %.2f ms' % d['roofline']['avg_launch_ms'], 'walk+rest %.2f ms' % d['kernel_ms']['traceback_and_rest_per_step'], '%.4e' % d['value'], d['bit_exact_sample'])" | tee -a $out/ab_walk.log
  done
 done
done
GNX_DEBUG=1 python bench.py --no-cpu --no-host --no-extras --steps 1 --warmup 0 --verify 0 2>&1 | grep "gnx fp" | head -12 | tee $out/fp_debug.log
