#!/bin/bash
out=gpurun_out/r4f; mkdir -p $out
export TMPDIR=/tmp
for tag in evbr1 evbr0; do
  GNX_LIB_PATH=$PWD/tools/ab/lib_$tag.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_entry.py -m gpu -x -q -k "not ten_million and not one_million" > $out/pytest_$tag.log 2>&1; echo "$tag pytest rc=$?" | tee -a $out/summary.log
  tail -3 $out/pytest_$tag.log | tee -a $out/summary.log
done
for rep in 1 2 3; do
  for tag in ev0 evbr1 evbr0; do
    GNX_LIB_PATH=$PWD/tools/ab/lib_$tag.so python bench.py --no-cpu --no-host --no-extras --verify 4000 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', 'step %.3f ms' % d['ms_per_step'], 'dev step %.3f ms' % d['ms_per_step_device_resident'], 'sweep %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], 'tb %.3f' % d['kernel_ms']['traceback_and_rest_per_step'], d['bit_exact_sample'])" | tee -a $out/ab.log
  done
done
