mkdir -p gpurun_out/r3q
timeout 1200 python -m pytest tests/test_host_entry.py tests/test_gpu_parity.py tests/test_const_long.py -m gpu -x -q > gpurun_out/r3q/pytest.log 2>&1; tail -3 gpurun_out/r3q/pytest.log
bash tools/bench_all.sh > gpurun_out/r3q/all_series.jsonl 2>gpurun_out/r3q/err.log
for k in affine const; do timeout 600 python tools/bench_shapes.py $k > gpurun_out/r3q/shapes_$k.jsonl 2>>gpurun_out/r3q/err.log; done
g++ -std=c++17 -O2 -Iinclude -o tools/bench_cabi.bin tools/bench_cabi.cpp gonomics_amd/libgonomics_align_hip.so -Wl,-rpath,$PWD/gonomics_amd -L/opt/rocm/lib -lamdhip64 && tools/bench_cabi.bin > gpurun_out/r3q/cabi.jsonl
cut -c1-300 gpurun_out/r3q/all_series.jsonl
cat gpurun_out/r3q/cabi.jsonl
