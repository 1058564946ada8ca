#!/bin/bash
# Every other measured series in one go (regression check after kernel changes).  Usage on the GPU box: bash tools/bench_all.sh > out.jsonl
for s in const local; do timeout 600 python bench.py --no-cpu --no-host --no-extras --steps 3 --series $s 2>/dev/null; done
GNX_FASTPATH=0 timeout 600 python bench.py --no-cpu --no-host --no-extras --steps 3 2>/dev/null | sed 's/^{"metric": "DP cells\/sec + aligned pairs\/sec, affine-gap 150bp x 10kb batch"/{"metric": "same, general path (GNX_FASTPATH=0)"/'
timeout 600 python bench.py --no-cpu --no-host --no-extras --pairs 1000000 --steps 2 --verify 64 2>/dev/null | sed 's/^{"metric": "DP cells\/sec + aligned pairs\/sec, affine-gap 150bp x 10kb batch"/{"metric": "same, 1 M pairs in one call"/'
timeout 600 python tools/bench_n1.py 2>/dev/null
timeout 600 python tools/bench_n2.py 200000 2>/dev/null
timeout 900 python tools/bench_long.py 2>/dev/null
