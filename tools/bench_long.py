#!/usr/bin/env python3
"""Measurement for long pairs (not the headline metric): config C5 in miniature -- ONT-style 20 kb reads against 100 kb windows,
ConstGap_highMem semantics -- and the latency of one 10 kb x 10 kb AffineGap pair (what cmd/cigarToBed runs).  Multi-strip
pairs run as pipelined workgroups (one per 160-row strip).  One JSON line per series."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import oracle  # noqa: E402
from gonomics_amd import _lib, align  # noqa: E402


def ont_read(rng, window, n):
    off = int(rng.integers(0, window.shape[0] - n))
    src = window[off:off + n]
    return common.mutate(rng, src, 0.04, 0.06, geo=0.6)[:n]


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 160 << 30))  # the direction matrix of a C5 pair is 0.5 GB: 256 pairs per launch
    rng = np.random.default_rng(5)
    sc = align.HumanChimpTwoScoreMatrix
    # one 10 kb x 10 kb affine pair
    a = rng.integers(0, 4, size=9700).astype(np.uint8)
    b = common.mutate(rng, a, 0.01, 0.003)[:10000]
    p = _lib.make_params(_lib.GNX_AFFINE_GAP, sc, -600, -150)
    _lib.align_batch(p, [a[:500]], [b[:500]])
    t0 = time.perf_counter()
    s1, o1, f1 = _lib.align_batch(p, [a], [b])
    dt = time.perf_counter() - t0
    tm = _lib.get_timing()
    e = oracle.align_batch(oracle.MODE_AFFINE, sc, -600, -150, [a], [b])
    ok = int(s1[0]) == int(e[0][0]) and np.array_equal(o1["run_length"], e[1]["run_length"]) and np.array_equal(o1["op"], e[1]["op"])
    print(json.dumps({"series": "one AffineGap pair %d x %d" % (a.shape[0], b.shape[0]), "bit_exact": bool(ok), "host_call_ms": dt * 1e3,
                      "kernel_ms": {"fill": tm["fill_ms"], "traceback": tm["traceback_ms"]}, "cells_per_s_fill": a.shape[0] * b.shape[0] / (tm["fill_ms"] * 1e-3)}))
    # C5 in miniature
    n, m = 20000, 100000
    windows = [rng.integers(0, 4, size=m).astype(np.uint8) for _ in range(pairs)]
    reads = [ont_read(rng, w, n) for w in windows]
    pc = _lib.make_params(_lib.GNX_CONST_GAP_HIGHMEM, sc, -430)
    for _ in range(2):  # the first call allocates the workspace (hipMalloc of up to 128 GB); report the second
        t0 = time.perf_counter()
        s, ops, off = _lib.align_batch(pc, reads, windows)
        dt = time.perf_counter() - t0
    tm = _lib.get_timing()
    t1 = time.perf_counter()
    e = oracle.align_batch(oracle.MODE_CONST_HIGHMEM, sc, -430, 0, reads[:1], windows[:1])
    cpu_s = time.perf_counter() - t1
    k1 = int(off[1])
    ok = int(s[0]) == int(e[0][0]) and np.array_equal(ops["run_length"][:k1], e[1]["run_length"]) and np.array_equal(ops["op"][:k1], e[1]["op"])
    cells = sum(len(r) * len(w) for r, w in zip(reads, windows))
    same = None
    if len(sys.argv) > 2 and sys.argv[2] == "verify":  # every pair against the sequential-strip run of the same kernels (no hand-over between workgroups)
        os.environ["GNX_NO_PIPE"] = "1"
        s2, ops2, off2 = _lib.align_batch(pc, reads, windows)
        del os.environ["GNX_NO_PIPE"]
        same = bool(np.array_equal(s, s2) and np.array_equal(off, off2) and np.array_equal(ops["run_length"], ops2["run_length"]) and np.array_equal(ops["op"], ops2["op"]))
    print(json.dumps({"series": "C5 miniature: %d x ConstGap_highMem(%d x %d)" % (pairs, n, m), "first_pair_bit_exact": bool(ok), "all_pairs_equal_sequential_strips": same, "host_call_s": dt,
                      "kernel_ms": {"fill": tm["fill_ms"], "traceback": tm["traceback_ms"], "total": tm["total_ms"]}, "launches": tm["n_launches"],
                      "cells_per_s_kernels": cells / (tm["total_ms"] * 1e-3), "cells_per_s_fill": cells / (tm["fill_ms"] * 1e-3),
                      "cpu_oracle_1thread_cells_per_s": n * m / cpu_s}))


if __name__ == "__main__":
    main()
