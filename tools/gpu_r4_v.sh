#!/bin/bash
out=gpurun_out/r4v; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "row_block_shortcut or small_batch_routing" > $out/newtests.log 2>&1; grep -E "passed|failed|^E " $out/newtests.log | head -8
