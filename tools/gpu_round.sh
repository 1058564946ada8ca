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
# SQ counters of the dominant kernel (three passes; 32768 pairs = 4096 waves of fp_sweep_kernel)
g=0
for grp in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT"; do
  g=$((g+1))
  timeout 900 rocprofv3 --pmc $grp -d $out/pmc_sq$g -o pmc --output-format csv -- python $repo/bench.py --no-cpu --steps 1 --warmup 0 --verify 0 --pairs 32768 > $out/pmc_sq$g.json 2> $out/pmc_sq$g.err
done
cd $repo
# the other measured series (regression view): const / local / general path, N1, N2, long pairs, host-buffer entry point
bash tools/bench_all.sh > $out/all_series.jsonl 2>> $out/bench.err
timeout 300 python tools/bench_host.py 100000 > $out/host_entry.jsonl 2>> $out/bench.err
find $out -name '*.db' -size +20M -delete
ls -la $out $out/stats 2>/dev/null | head -40
tail -3 $out/pytest_gpu.log; cat $out/smoke.log | tail -1; cat $out/bench.json
