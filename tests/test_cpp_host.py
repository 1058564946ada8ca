"""The C++ host mirror (include/gonomics_align.hpp) over the C ABI: builds everywhere, runs on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.bin")
LIB = os.path.join(ROOT, "gonomics_amd", "libgonomics_align_hip.so")


def _build():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-o", BIN, SRC, LIB,
                           "-Wl,-rpath," + os.path.join(ROOT, "gonomics_amd"), "-L/opt/rocm/lib", "-lamdhip64"])


def test_cpp_host_mirror_builds_and_refuses_without_gpu():
    _build()
    rc = subprocess.call([BIN])
    assert rc in (0, 2)  # 2 == "no HIP device" (no CPU fallback); 0 on a GPU box


@pytest.mark.gpu
def test_cpp_host_mirror_runs():
    _build()
    assert subprocess.call([BIN]) == 0
