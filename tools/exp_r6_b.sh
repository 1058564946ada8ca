out=gpurun_out/r6_h; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q --deselect "tests/test_long_range.py::test_long_pairs_equal_the_oracle[affine_1M]" > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
timeout 900 python bench.py --series long --no-extras --pairs 1024 > $out/bench_long_1024.json 2> $out/bench.err; cut -c1-1500 $out/bench_long_1024.json
timeout 900 python bench.py --series long --no-extras > $out/bench_long.json 2>> $out/bench.err; cut -c1-700 $out/bench_long.json
