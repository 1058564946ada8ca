#!/bin/bash
out=gpurun_out/r4k; mkdir -p $out
GNX_DEBUG=2 python bench.py --no-cpu --no-host --no-extras --steps 1 --warmup 0 --verify 0 2>&1 | grep "gnx fp" | head -40 | tee $out/census.log
timeout 600 python -m pytest tests/test_concurrent_pairs.py -m gpu -x -q 2>&1 | tail -3
