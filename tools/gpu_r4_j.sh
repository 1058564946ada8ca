#!/bin/bash
out=gpurun_out/r4j; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
bash tools/switch_matrix.sh > $out/switch_matrix.log 2>&1; cat $out/switch_matrix.log
timeout 1000 python tools/stress.py 600 91 > $out/stress_long.log 2>&1; tail -2 $out/stress_long.log
