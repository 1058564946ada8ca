#!/bin/bash
out=gpurun_out/r4e; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
bash tools/ab_libs.sh tools/ab/lib_noskew.so 2>&1 | tee $out/ab_skew.log
