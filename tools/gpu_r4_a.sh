#!/bin/bash
# round 4, call A: parity suite + smoke + default bench line (value = host-entry rate) on the round's first build
out=gpurun_out/r4a; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $out/smoke.log 2>&1
timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err
tail -5 $out/pytest_gpu.log; tail -1 $out/smoke.log; cut -c1-1500 $out/bench.json; tail -5 $out/bench.err
