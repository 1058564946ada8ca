#!/bin/bash
# ThreadSanitizer over the read path's worker pool (the C++ mirror compiled into a TSAN test binary; the library itself is not instrumented)
out=gpurun_out/r4u; mkdir -p $out
python - <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gsw_cpp as tc
from test_gsw_reads import make_case
import common
seqs, edges, reads = [], [], []
for seed in (21, 22, 23, 24, 25, 26, 27, 28):
    s2, e2, r2 = make_case(seed, "wide3")
    edges += [(u + len(seqs), v + len(seqs)) for u, v in e2]; seqs += s2; reads += r2
tc.write_case("/tmp/tsan_case.txt", seqs, edges, reads, 16, 1, common.matrices()["HumanChimpTwo"])
PY
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -Iinclude -o /tmp/gsw_tsan.bin tests/cpp/gsw_mirror_test.cpp tests/cpp/gsw_cpu_backend.cpp gonomics_amd/libgonomics_align_hip.so oracle/liboracle.so -Wl,-rpath,$PWD/gonomics_amd -Wl,-rpath,$PWD/oracle -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
GNX_GSW_THREADS=8 GNX_GSW_REPEAT=2 TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" timeout 600 setarch x86_64 -R /tmp/gsw_tsan.bin /tmp/tsan_case.txt /tmp/tsan_out.txt > $out/tsan.log 2>&1; echo "rc=$?" | tee -a $out/tsan.log
grep -c "WARNING: ThreadSanitizer" $out/tsan.log | tee -a $out/tsan_summary.txt
grep -A12 "WARNING: ThreadSanitizer" $out/tsan.log | grep -E "WARNING|#0|#1|#2" | head -40 | tee -a $out/tsan_summary.txt
GNX_GSW_THREADS=1 tests/cpp/gsw_mirror_test.bin /tmp/tsan_case.txt /tmp/ref_out.txt && grep -v "^#" /tmp/ref_out.txt > /tmp/a.txt && grep -v "^#" /tmp/tsan_out.txt > /tmp/b.txt && cmp /tmp/a.txt /tmp/b.txt && echo "tsan run == one-thread run" | tee -a $out/tsan_summary.txt
