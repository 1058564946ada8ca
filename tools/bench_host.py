#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (gnx_align_batch_windows) on the C2 workload.

bench.py's `value` is measured with the inputs resident in HBM (gnx_align_batch_device); this is the same batch through the
entry point a cgo shim would call with Go slices: H2D of reads + chunk + offsets, kernels, D2H of scores / CIGAR offsets /
CIGAR runs into malloc'ed host arrays, all inside the timed region.  Usage on the GPU box: python tools/bench_host.py [n_pairs]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import common  # noqa: E402
from gonomics_amd import _lib, align  # noqa: E402


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    reads, chunk = common.c2_workload(2, n_pairs)
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 0))
    p = _lib.make_params(_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150)
    a_buf = np.ascontiguousarray(reads.reshape(-1))
    a_start = np.arange(n_pairs, dtype=np.int64) * reads.shape[1]
    a_len = np.full(n_pairs, reads.shape[1], dtype=np.int64)
    b_start = np.zeros(n_pairs, dtype=np.int64)
    b_len = np.full(n_pairs, chunk.shape[0], dtype=np.int64)
    out = []
    for it in range(4):
        t0 = time.perf_counter()
        sc, ops, off = _lib.align_batch_windows(p, a_buf, a_start, a_len, chunk, b_start, b_len)
        dt = time.perf_counter() - t0
        tm = _lib.get_timing()
        out.append((dt, tm["total_ms"]))
    dt, kern = min(out[1:])
    cells = n_pairs * reads.shape[1] * chunk.shape[0]
    print(json.dumps({"series": "host-buffer entry point (PCIe-inclusive), C2 AffineGap", "pairs": n_pairs, "host_call_ms": dt * 1e3,
                      "device_ms_inside": kern, "cells_per_s": cells / dt, "pairs_per_s": n_pairs / dt,
                      "bytes_h2d": int(a_buf.nbytes + chunk.nbytes + 4 * 8 * n_pairs), "bytes_d2h": int(sc.nbytes + off.nbytes + ops.nbytes),
                      "note": "includes the ctypes wrapper's copy of the CIGAR runs into numpy arrays"}))


if __name__ == "__main__":
    main()
