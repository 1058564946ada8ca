#!/usr/bin/env python3
"""print registers / LDS / scratch of the kernels of the built library whose demangled name contains the argument"""
import sys, os, pathlib, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_kernel_resources as t
ks = t._kernels(pathlib.Path(tempfile.mkdtemp()))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for n, k in sorted(ks.items()):
    if pat in n:
        print("%-90s vgpr %3d agpr %3d lds %6d scratch %4d spill %3d waves/simd %d" % (n[:90], k["vgpr_count"], k["agpr_count"], k["group_segment_fixed_size"], k["private_segment_fixed_size"], k["vgpr_spill_count"], t._waves_per_simd(k)))
