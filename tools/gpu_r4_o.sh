#!/bin/bash
# split batches (GNX_FP_SPLIT): parity with small batches forced through the split, then A/B of the split fraction
out=gpurun_out/r4o; mkdir -p $out
GNX_FP_SPLIT=50 GNX_FP_SPLIT_MIN=16 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $out/parity.log
for sp in 0 50 70 80 0 70; do
  GNX_FP_SPLIT=$sp timeout 300 python bench.py --no-cpu --no-host --no-extras --steps 10 --warmup 3 --verify 256 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $sp: host-entry ms', round(d['ms_per_step'],3), 'device ms', round(d.get('ms_per_step_device_resident',0),3), 'sweep', d.get('kernel_ms',{}).get('dominant_kernel_per_step'), 'bit_exact', d.get('bit_exact_sample'))" | tee -a $out/ab.log
done
for sp in 0 50 75; do
  GNX_FP_SPLIT=$sp timeout 300 python tools/bench_shapes.py affine 1000,1200,100000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $sp 1000x1200:', round(d['default']['ms'],3), 'ms', d['default']['cells_per_s'], d['same_results'])" | tee -a $out/ab.log
done
