#!/bin/bash
out=gpurun_out/r4c; mkdir -p $out
export TMPDIR=/tmp
Q="--no-cpu --no-host --no-extras --series long --pairs 1024 --steps 2 --warmup 1 --verify 2"
for rep in 1 2; do
  for wg in 1 0; do
    GNX_CL_WG=$wg timeout 600 python bench.py $Q 2>>$out/bench.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('WG=$wg', 'step %.2f ms' % d['ms_per_step'], 'sweep %.2f ms' % d['roofline']['avg_launch_ms'], 'frac %.3f' % d['roofline']['frac'], '%.4e' % d['value'], d['bit_exact_sample'])" | tee -a $out/ab.log
  done
done
bash tools/pmc_env_ab.sh $out "--no-cpu --no-host --no-extras --series long --pairs 1024 --steps 1 --warmup 0 --verify 0" "cl_sweep" "wg1:GNX_CL_WG=1" "wg0:GNX_CL_WG=0"
# headline sweep: LDS conflicts with the skewed rings
bash tools/pmc_env_ab.sh $out "--no-cpu --no-host --no-extras --steps 1 --warmup 0 --verify 0" "fp_sweep_kernel" "fp:GNX_X=0"
python bench.py --no-cpu --no-extras --steps 5 --warmup 2 2>>$out/bench.err | cut -c1-900
