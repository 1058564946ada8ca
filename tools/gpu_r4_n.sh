#!/bin/bash
out=gpurun_out/r4n; mkdir -p $out
GNX_DEBUG=1 GNX_FASTPATH=2 python tools/bench_shapes.py affine 1000,1200,100000 2>&1 | grep -v "^\[gnx\] general" | tail -40 | cut -c1-300 | tee $out/timeline.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o s -- python $GRAFT_REPO_ROOT/tools/bench_shapes.py affine 1000,1200,100000 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200 | tee $out/kstats.txt
