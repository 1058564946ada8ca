#!/usr/bin/env python3
"""Kernel time of the C5 sweep for a library build whose results may be WRONG (ablation builds that drop the hand-over waits):
runs one bench_long-style batch ignoring the return code and prints the library's own fill timing.  Usage: python tools/exp_c5_time.py [pairs]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gonomics_amd import _lib, align  # noqa: E402
import bench  # noqa: E402


def main():
    import torch
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 200 << 30))
    dev = torch.device("cuda", 0)
    n, m = 20000, 100000
    rng = np.random.default_rng(5)
    reads = rng.integers(0, 4, size=(pairs, n), dtype=np.uint8)
    wins = rng.integers(0, 4, size=(pairs, m), dtype=np.uint8)
    p = _lib.make_params(_lib.GNX_CONST_GAP, align.HumanChimpTwoScoreMatrix, -430, 0)
    d_a = torch.from_numpy(reads.reshape(-1)).to(dev); d_b = torch.from_numpy(wins.reshape(-1)).to(dev)
    d_as = torch.arange(pairs, dtype=torch.int64, device=dev) * n; d_bs = torch.arange(pairs, dtype=torch.int64, device=dev) * m
    h_al = np.full(pairs, n, dtype=np.int64); h_bl = np.full(pairs, m, dtype=np.int64)
    d_score = torch.zeros(pairs, dtype=torch.int64, device=dev); d_off = torch.zeros(pairs + 1, dtype=torch.int64, device=dev)
    cap = pairs * 40000
    d_ops = torch.zeros(cap * 16, dtype=torch.uint8, device=dev)
    tot = ctypes.c_int64()
    for it in range(3):
        rc = L.gnx_align_batch_device(ctypes.byref(p), pairs, d_a.data_ptr(), d_as.data_ptr(), 0, d_b.data_ptr(), d_bs.data_ptr(), 0,
                                      h_al.ctypes.data, h_bl.ctypes.data, d_score.data_ptr(), d_ops.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(tot), None)
        torch.cuda.synchronize()
        tm = _lib.get_timing()
        print("rc", rc, "fill_ms %.2f tb_ms %.2f path %s" % (tm["fill_ms"], tm["traceback_ms"], tm["fast_path"]), flush=True)


if __name__ == "__main__":
    main()
