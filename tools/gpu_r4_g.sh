#!/bin/bash
out=gpurun_out/r4g; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
tests/cpp/concurrent_pairs_test.bin 16 1000 8 | tee $out/concurrent_pairs.json
timeout 700 python tools/stress.py 420 41 > $out/stress.log 2>&1; tail -2 $out/stress.log
timeout 400 python tools/stress.py 240 42 >> $out/stress.log 2>&1; tail -2 $out/stress.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-600 $out/bench.json
