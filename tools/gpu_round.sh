#!/bin/bash
# Round-end GPU sequence: parity suite, smoke, bench, rocprofv3 kernel stats + HBM PMC passes (separate runs).
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
tag=${1:-r1}
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $out/smoke.log 2>&1
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
repo=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $repo/bench.py --no-cpu --steps 3 --warmup 1 > $out/stats_bench.json 2> $out/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d $out/pmc_$c -o pmc --output-format csv -- python $repo/bench.py --no-cpu --steps 1 --warmup 0 --verify 0 > $out/pmc_$c.json 2> $out/pmc_$c.err
done
for c in FETCH_SIZE WRITE_SIZE; do
  GNX_FASTPATH=0 timeout 900 rocprofv3 --pmc $c -d $out/pmc_gen_$c -o pmc --output-format csv -- python $repo/bench.py --no-cpu --steps 1 --warmup 0 --verify 0 > $out/pmc_gen_$c.json 2> $out/pmc_gen_$c.err
done
GNX_FASTPATH=0 timeout 600 python $repo/bench.py --no-cpu > $out/bench_general_path.json 2>> $out/bench.err
cd $repo
find $out -name '*.db' -size +20M -delete
ls -la $out $out/stats 2>/dev/null | head -40
tail -3 $out/pytest_gpu.log; cat $out/smoke.log | tail -1; cat $out/bench.json
