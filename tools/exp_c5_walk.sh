#!/bin/bash
# C5 walk stage: tiles per round (GNX_CL_WALK_SPEC) x snapshot spacing (GNX_CL_CKC) x pairs per launch.  bash tools/exp_c5_walk.sh > out.txt
for pairs in 1024 2048; do
  for ckc in 224 448; do
    for spec in 0 2 3; do
      r=$(GNX_CL_CKC=$ckc GNX_CL_WALK_SPEC=$spec timeout 600 python bench.py --series long --pairs $pairs --steps 2 --warmup 1 --no-extras --no-cpu --no-host --verify 2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); k=r['kernel_ms']; print('sweep %.2f ms  walk+rest %.2f ms  step %.2f ms  ok %s' % (k['all_fill_kernels_per_step'], k['traceback_and_rest_per_step'], r['ms_per_step'], r['bit_exact_sample']))")
      echo "pairs $pairs ckc $ckc spec $spec: $r"
    done
  done
done
