#!/bin/bash
out=$PWD/gpurun_out/r4y; mkdir -p $out
repo=$PWD
cd /tmp; export TMPDIR=/tmp
GNX_FASTPATH=2 timeout 300 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $repo/tools/bench_shapes.py affine 1000,1200,100000 > /dev/null 2> $out/err.log
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-170
