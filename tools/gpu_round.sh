#!/bin/bash
# Round-end GPU sequence: parity suite, smoke, bench lines, rocprofv3 kernel stats + HBM / SQ PMC passes (separate runs, as the
# MI355X guide prescribes).  Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]; then, in the build container,
# python tools/collect_profiles.py gpurun_out/<tag> r6
tag=${1:-r6}
repo=$PWD
out=$repo/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c 'import bench; print(bench.kernel_source_hash())' > $out/kernel_source_hash.txt   # what the counters below are taken from
echo "${GNX_COMMIT:-unknown}" > $out/commit.txt                                            # (the box has no .git: pass GNX_COMMIT=$(git rev-parse --short HEAD))
timeout 1500 python -m pytest tests -m gpu -x -q $GNX_PYTEST_EXTRA > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $out/smoke.log 2>&1
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 900 python bench.py --series long --no-extras > $out/bench_long.json 2>> $out/bench.err
# the N > 1 flow of bench.py on this 1-GPU box: two ranks sharing the device over gloo (plumbing check; no scaling claim)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-gpu --dist-backend gloo --pairs 20000 --steps 2 --no-cpu > $out/bench_2ranks_shared_gpu.json 2>> $out/bench.err
cd /tmp
Q="--no-cpu --no-host --no-extras --verify 0"
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $repo/bench.py $Q --steps 3 --warmup 1 > $out/stats_bench.json 2> $out/stats.err
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats_long -o stats --output-format csv -- python $repo/bench.py $Q --series long --pairs 1024 --steps 2 --warmup 1 > $out/stats_long_bench.json 2>> $out/stats.err
# reads of 800 bases (5 row blocks in one launch: fp_sweep_levels_kernel) and 20 kb x 100 kb (125 row blocks)
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats_rb -o stats --output-format csv -- python $repo/tools/bench_shapes.py affine 800,10000,32768 > $out/stats_rb_bench.json 2>> $out/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d $out/pmc_fast_$c -o pmc --output-format csv -- python $repo/bench.py $Q --steps 1 --warmup 0 > /dev/null 2> $out/pmc.err
  GNX_FASTPATH=0 timeout 900 rocprofv3 --pmc $c -d $out/pmc_general_$c -o pmc --output-format csv -- python $repo/bench.py $Q --steps 1 --warmup 0 > /dev/null 2>> $out/pmc.err
  timeout 900 rocprofv3 --pmc $c -d $out/pmc_long_$c -o pmc --output-format csv -- python $repo/bench.py $Q --series long --pairs 1024 --steps 1 --warmup 0 > /dev/null 2>> $out/pmc.err
done
g=0
for grp in "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  g=$((g+1))
  # the headline launch (100 000 pairs = 6.1 rounds of the 2048 wave slots) AND a 2-round launch (32 768 pairs: half of it is ramp and
  # tail -- what round 2's pass measured; DESIGN 4.5 explains 0.90 vs 0.74 with it)
  timeout 900 rocprofv3 --pmc $grp -d $out/pmc_sq_fast$g -o pmc --output-format csv -- python $repo/bench.py $Q --steps 1 --warmup 0 > /dev/null 2>> $out/pmc.err
  timeout 900 rocprofv3 --pmc $grp -d $out/pmc_sq32k_fast$g -o pmc --output-format csv -- python $repo/bench.py $Q --steps 1 --warmup 0 --pairs 32768 > /dev/null 2>> $out/pmc.err
  timeout 900 rocprofv3 --pmc $grp -d $out/pmc_sq_long$g -o pmc --output-format csv -- python $repo/bench.py $Q --series long --steps 1 --warmup 0 --pairs 1024 > /dev/null 2>> $out/pmc.err
done
cd $repo
# the other measured series (regression view)
bash tools/bench_all.sh > $out/all_series.jsonl 2>> $out/bench.err
timeout 600 python tools/bench_host.py 100000 1000000 > $out/host_entry.jsonl 2>> $out/bench.err
timeout 600 python tools/bench_shapes.py affine > $out/shapes_affine.jsonl 2>> $out/bench.err
timeout 600 python tools/bench_shapes.py const > $out/shapes_const.jsonl 2>> $out/bench.err
timeout 600 python tools/bench_shapes.py local > $out/shapes_local.jsonl 2>> $out/bench.err
tools/lds_occupancy.bin > $out/lds_occupancy.txt 2>>$out/bench.err
timeout 600 python tools/bench_gsw.py > $out/gsw_reads.jsonl 2>> $out/bench.err
g++ -std=c++17 -O2 -Iinclude -o tools/bench_cabi.bin tools/bench_cabi.cpp gonomics_amd/libgonomics_align_hip.so -Wl,-rpath,$PWD/gonomics_amd -L/opt/rocm/lib -lamdhip64 2>> $out/bench.err && tools/bench_cabi.bin > $out/cabi_n1_n2.jsonl 2>> $out/bench.err
# round 4: concurrency at the boundary, the C5 sweep's counters with one strip / four strips per workgroup, workgroup occupancy census, stress
g++ -std=c++17 -O2 -pthread -Iinclude -o tests/cpp/concurrent_pairs_test.bin tests/cpp/concurrent_pairs_test.cpp gonomics_amd/libgonomics_align_hip.so -Wl,-rpath,$PWD/gonomics_amd -L/opt/rocm/lib -lamdhip64 2>> $out/bench.err && tests/cpp/concurrent_pairs_test.bin 16 1000 8 > $out/concurrent_pairs.json 2>> $out/bench.err
bash tools/pmc_env_ab.sh gpurun_out/$tag/c5_ab "--no-cpu --no-host --no-extras --series long --pairs 1024 --steps 1 --warmup 0 --verify 0" "cl_sweep" "wg4:GNX_CL_WG=1" "one_strip:GNX_CL_WG=0" > /dev/null 2>> $out/bench.err; cp gpurun_out/$tag/c5_ab/pmc_ab.txt $out/pmc_c5_wg_ab.txt
timeout 300 python tools/pair_latency.py 300 > $out/pair_latency.jsonl 2>> $out/bench.err
# round 5: pairs beyond the static int32 range (snapshot path on moving bases; the 2 Mb / 5 Mb row-panel runs are in profiles/r5_long_pairs.jsonl),
# the latency geometry against the general path, the whole cmd/faChunkAlign command, the graph aligner at genome scale
timeout 900 python tools/long_pairs.py gpu const_150k affine_340k affine_q1_300k affine_1M const_300k_2M affine_2M affine_5M > $out/long_pairs.jsonl 2>> $out/bench.err
# round 6: a one-call process (cmd/cigarToBed): every case in a process of its own after the device has been idle for a while (an allocation waits for the driver's
# clearing of memory that was freed a moment ago -- profiles/r6_alloc_probe.txt -- so what the first call of a process costs depends on what ran before it)
for c in affine_1M affine_2M; do sleep 20; timeout 300 python tools/long_pairs.py gpu $c >> $out/long_pairs_first_call.jsonl 2>> $out/bench.err; done
# counters of the sweep of one long pair (valu_busy, waves per SIMD, HBM bytes): profiles/r6_pmc_long_pair.{txt,json}
bash tools/pmc_long_pair.sh gpurun_out/$tag affine_1M affine_340k const_150k > /dev/null 2>> $out/bench.err
timeout 120 tools/alloc_probe.bin 48 > $out/alloc_probe.txt 2>&1
timeout 600 python tools/lat_crossover.py affine > $out/lat_crossover.jsonl 2>> $out/bench.err
timeout 600 python tools/lat_crossover.py const >> $out/lat_crossover.jsonl 2>> $out/bench.err
timeout 600 python tools/bench_n1_cmd.py 8 30000 3 > $out/n1_cmd.json 2>> $out/bench.err
timeout 900 python tools/bench_gsw_genome.py 100000000 300000 2>> $out/bench.err | grep "^{" > $out/gsw_genome.jsonl
[ -z "$GNX_SKIP_GENOME" ] && timeout 1500 python tools/bench_gsw_genome.py 3000000000 200000 2>> $out/bench.err | grep "^{" >> $out/gsw_genome.jsonl
# a handful of long pairs per batch call: the 16-lane snapshot kernels against the 64-lane kernels + walk farm (the shipped rule: up to 64 pairs)
(timeout 300 python tools/few_long_pairs.py affine 200000 2 4 16; timeout 300 python tools/few_long_pairs.py const 120000 4 32; FLP_EXTRA=80000 timeout 300 python tools/few_long_pairs.py const 20000 16 64 128) > $out/few_long_pairs.jsonl 2>> $out/bench.err
# the walk farm's switches on the long pairs (tiles per round, plain rounds, the one-workgroup walks)
for sw in "GNX_W64_FARM=8" "GNX_W64_FARM=32" "GNX_W64_FARM_PIPE=0" "GNX_W64_FARM=0" "GNX_W64_FARM=0 GNX_W64_SPEC=0"; do
  env $sw timeout 300 python tools/long_pairs.py gpu const_150k affine_340k affine_1M const_300k_2M 2>> $out/bench.err | sed "s/^{/{\"switch\": \"$sw\", /" >> $out/long_pairs_farm_ab.jsonl
done
timeout 600 python tools/gsw_threads.py 2>> $out/bench.err | grep "^{" > $out/gsw_threads.jsonl
tools/wg_occupancy.bin > $out/wg_occupancy.txt 2>> $out/bench.err
timeout 700 python tools/stress.py ${GNX_STRESS_S:-420} 77 > $out/stress.log 2>&1
[ "$GNX_SWITCH_MATRIX" = "1" ] && bash tools/switch_matrix.sh > $out/switch_matrix.log 2>&1
[ "$GNX_SWITCH_MATRIX" = "farm" ] && bash tools/switch_matrix.sh farm > $out/switch_matrix_farm.log 2>&1
[ "$GNX_STRESS_ROUTES" = "1" ] && bash tools/stress_routes.sh > $out/stress_routes.log 2>&1
find $out -name '*.db' -size +20M -delete
tail -3 $out/pytest_gpu.log; tail -1 $out/smoke.log; cat $out/bench.json | cut -c1-400; cat $out/bench_long.json | cut -c1-400
