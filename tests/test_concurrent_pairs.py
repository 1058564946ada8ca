"""Concurrency at the boundary: 16 host threads x 1000 gnx_align_pair calls from compiled code -- the reference's
worker-pool pattern (genomeGraph/routines.go:12-65) -- give the results of the same calls made serially, every caller its own error,
at >= 8 x the serial rate (the library combines concurrent single-pair calls into device batches)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "concurrent_pairs_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "concurrent_pairs_test.bin")
LIB = os.path.join(ROOT, "gonomics_amd", "libgonomics_align_hip.so")


def _build():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", BIN, SRC, LIB,
                           "-Wl,-rpath," + os.path.join(ROOT, "gonomics_amd"), "-L/opt/rocm/lib", "-lamdhip64"])


def test_concurrent_pairs_builds_and_refuses_without_gpu():
    _build()
    assert subprocess.call([BIN, "2", "2"]) in (0, 1, 2)  # 2 == no HIP device (no CPU fallback)


@pytest.mark.gpu
def test_sixteen_threads_of_single_pair_calls():
    _build()
    r = subprocess.run([BIN, "16", "1000", "8"], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["mismatches"] == 0 and d["calls_with_GNX_EBASE"] > 0
    assert d["combined_batches"] > 0 and d["speedup"] >= 8.0, d
    assert r.returncode == 0
    # every fourth thread with other parameters (align.ConstGap): requests of different parameters share the queue, never a batch
    r = subprocess.run([BIN, "16", "300", "2", "mixed"], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["mismatches"] == 0 and d["combined_batches"] > 0 and r.returncode == 0, d
