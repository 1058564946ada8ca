#!/bin/bash
out=gpurun_out/r4q; mkdir -p $out
timeout 600 python tools/pair_latency.py 300 2>&1 | tee $out/latency.jsonl | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o lat -- python $GRAFT_REPO_ROOT/tools/pair_latency.py 100 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*hip_api_stats.csv" | head -1); head -25 "$f" | cut -c1-160 | tee $out/hip_stats.txt
