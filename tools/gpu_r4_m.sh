#!/bin/bash
# graph aligner's read path: worker pool vs one thread, stage breakdown
out=gpurun_out/r4m; mkdir -p $out
timeout 900 python -m pytest tests/test_gsw_cpp.py tests/test_gsw_reads.py -m gpu -x -q 2>&1 | tail -4 | tee $out/pytest.log
timeout 1500 python tools/bench_gsw.py 2>$out/bench_gsw.err | tee $out/gsw_reads.jsonl | cut -c1-900
tail -5 $out/bench_gsw.err
