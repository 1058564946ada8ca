#!/bin/bash
# straggler tile trimming A/B (GNX_FP_STRAG_TRIM=0 / 1) on one box
out=gpurun_out/r4l; mkdir -p $out
for t in 0 1; do
  echo "== trim $t" | tee -a $out/ab.log
  GNX_FP_STRAG_TRIM=$t GNX_DEBUG=1 python bench.py --no-cpu --no-host --no-extras --steps 2 --warmup 1 --verify 64 2>&1 | grep -E "gnx fp\] pairs|^\{" | tail -8 | cut -c1-400 | tee -a $out/ab.log
done
for t in 0 1; do
  GNX_FP_STRAG_TRIM=$t python bench.py --no-cpu --no-host --no-extras --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('trim $t', d['ms_per_step'], d.get('value_device_resident'), d['ok'] if 'ok' in d else '')" | tee -a $out/ab.log
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a $out/ab.log
