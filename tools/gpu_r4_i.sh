#!/bin/bash
out=gpurun_out/r4i; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gsw_reads.py tests/test_gsw_cpp.py tests/test_n2_gsw.py -m gpu -x -q > $out/pytest_gsw.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gsw.log
tail -15 $out/pytest_gsw.log
timeout 900 python tools/bench_gsw.py > $out/gsw_reads.jsonl 2>$out/gsw.err; cut -c1-400 $out/gsw_reads.jsonl; tail -3 $out/gsw.err
for rep in 1 2 3; do
  for tag in intree evpin; do
    if [ $tag = intree ]; then unset GNX_LIB_PATH; else export GNX_LIB_PATH=$PWD/tools/ab/lib_$tag.so; fi
    python bench.py --no-cpu --no-host --no-extras --verify 2000 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', 'step %.3f ms' % d['ms_per_step'], 'dev step %.3f ms' % d['ms_per_step_device_resident'], 'sweep %.3f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], d['bit_exact_sample'])" | tee -a $out/ab_evpin.log
  done
done
