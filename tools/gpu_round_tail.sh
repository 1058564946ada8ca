#!/bin/bash
# after tools/gpu_round.sh + collect_profiles.py: the bench line again (now that profiles/r6_hbm_traffic.json and r6_pmc_long_pair.json belong to these kernel sources, `traffic` is filled),
# the 64-lane constant-gap kernels against the 16-lane ones for bigger batches, the farm switch matrix, the stress legs of the forced routes
out=gpurun_out/${1:-r6_tail}; mkdir -p $out
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-300 $out/bench.json
(FLP_EXTRA=80000 timeout 900 python tools/few_long_pairs.py const 20000 128 256 512 1024) > $out/c5_w64_crossover.jsonl 2>> $out/bench.err; cat $out/c5_w64_crossover.jsonl | cut -c1-400
bash tools/switch_matrix.sh farm > $out/switch_matrix_farm.log 2>&1; cat $out/switch_matrix_farm.log
bash tools/stress_routes.sh 60 > $out/stress_routes.log 2>&1; cat $out/stress_routes.log
