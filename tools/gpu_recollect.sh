#!/bin/bash
# After a late change to the kernel sources: the part of tools/gpu_round.sh whose outputs are stamped with the kernel-source hash (bench lines, rocprofv3
# kernel stats, HBM / SQ PMC passes) + the GPU suite, into gpurun_out/<tag>; the other files of an earlier full round are kept by
#   cp -n gpurun_out/<earlier>/* gpurun_out/<tag>/ ; python tools/collect_profiles.py gpurun_out/<tag> r5
tag=${1:-r5c}
repo=$PWD
out=$repo/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c 'import bench; print(bench.kernel_source_hash())' > $out/kernel_source_hash.txt
echo "${GNX_COMMIT:-unknown}" > $out/commit.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $out/smoke.log 2>&1
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 900 python bench.py --series long --no-extras > $out/bench_long.json 2>> $out/bench.err
cd /tmp
Q="--no-cpu --no-host --no-extras --verify 0"
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $repo/bench.py $Q --steps 3 --warmup 1 > $out/stats_bench.json 2> $out/stats.err
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats_long -o stats --output-format csv -- python $repo/bench.py $Q --series long --pairs 1024 --steps 2 --warmup 1 > $out/stats_long_bench.json 2>> $out/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d $out/pmc_fast_$c -o pmc --output-format csv -- python $repo/bench.py $Q --steps 1 --warmup 0 > /dev/null 2> $out/pmc.err
  GNX_FASTPATH=0 timeout 900 rocprofv3 --pmc $c -d $out/pmc_general_$c -o pmc --output-format csv -- python $repo/bench.py $Q --steps 1 --warmup 0 > /dev/null 2>> $out/pmc.err
  timeout 900 rocprofv3 --pmc $c -d $out/pmc_long_$c -o pmc --output-format csv -- python $repo/bench.py $Q --series long --pairs 1024 --steps 1 --warmup 0 > /dev/null 2>> $out/pmc.err
done
g=0
for grp in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  g=$((g+1))
  timeout 900 rocprofv3 --pmc $grp -d $out/pmc_sq_fast$g -o pmc --output-format csv -- python $repo/bench.py $Q --steps 1 --warmup 0 > /dev/null 2>> $out/pmc.err
  timeout 900 rocprofv3 --pmc $grp -d $out/pmc_sq_long$g -o pmc --output-format csv -- python $repo/bench.py $Q --series long --steps 1 --warmup 0 --pairs 1024 > /dev/null 2>> $out/pmc.err
done
cd $repo
timeout 900 python tools/long_pairs.py gpu ${GNX_LONG_CASES:-const_150k affine_340k affine_1M const_300k_2M affine_2M} > $out/long_pairs.jsonl 2>> $out/bench.err
find $out -name '*.db' -size +20M -delete
tail -3 $out/pytest_gpu.log; tail -1 $out/smoke.log; cut -c1-300 $out/bench.json
