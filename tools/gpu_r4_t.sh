#!/bin/bash
out=gpurun_out/r4t; mkdir -p $out
g++ -std=c++17 -O2 -pthread -Iinclude -o tests/cpp/concurrent_pairs_test.bin tests/cpp/concurrent_pairs_test.cpp gonomics_amd/libgonomics_align_hip.so -Wl,-rpath,$PWD/gonomics_amd -L/opt/rocm/lib -lamdhip64
for k in 1 2 3; do tests/cpp/concurrent_pairs_test.bin 16 1000 8 | tee -a $out/conc.log; done
tests/cpp/concurrent_pairs_test.bin 4 1000 2 | tee -a $out/conc.log
tests/cpp/concurrent_pairs_test.bin 32 500 8 | tee -a $out/conc.log
tests/cpp/concurrent_pairs_test.bin 16 300 2 mixed | tee -a $out/conc.log
