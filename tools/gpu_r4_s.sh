#!/bin/bash
out=gpurun_out/r4s; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_host_entry.py -m gpu -x -q -k "not ten_million" > $out/default.log 2>&1; grep -E "passed|failed|^E " $out/default.log | head -20
GNX_FP_SMALL=0 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_const_long.py tests/test_host_entry.py tests/test_n1_gpu.py tests/test_n2_gsw.py tests/test_cpp_host.py tests/test_concurrent_pairs.py -m gpu -x -q -k "not ten_million" > $out/small0.log 2>&1; grep -E "passed|failed|^E " $out/small0.log | head -20
