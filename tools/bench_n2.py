#!/usr/bin/env python3
"""Measurement for the N2 seed-extension DPs (not the headline metric): a batch of gsw-shaped extensions -- read part of
75 bases against a target of read + perfectScore/600 bases -- through gnx_gsw_extend_batch, both sides.  One JSON line per side."""
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


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 32 << 30))
    rng = np.random.default_rng(9)
    sc = np.asarray(align.HumanChimpTwoScoreMatrix, dtype=np.int64)
    ref = rng.integers(0, 4, size=200000).astype(np.uint8)
    starts = rng.integers(0, ref.size - 200, size=pairs)
    betas, alphas = [], []
    for s in starts:
        read = ref[s:s + 75].copy()
        flip = rng.random(75) < 0.02
        read[flip] = (read[flip] + 1) % 4
        betas.append(read)
        alphas.append(ref[s:s + 75 + 12])  # extension = perfectScore/600 + len ~ 75*95/600 + 75
    for side, name in ((_lib.GNX_GSW_LEFT, "LeftDynamicAln"), (_lib.GNX_GSW_RIGHT, "RightDynamicAln")):
        a = [x[::-1].copy() for x in alphas] if side == _lib.GNX_GSW_LEFT else alphas   # left extensions are anchored at the end
        b = [x[::-1].copy() for x in betas] if side == _lib.GNX_GSW_LEFT else betas
        _lib.gsw_extend_batch(side, sc, -600, a[:256], b[:256])
        t0 = time.perf_counter()
        s, ei, ej, ops, off = _lib.gsw_extend_batch(side, sc, -600, a, b)
        dt = time.perf_counter() - t0
        tm = _lib.get_timing()
        ok = True
        for k in range(0, pairs, max(1, pairs // 200)):
            es, er, oi, oj = oracle.gsw_extend(side, sc, -600, a[k], b[k])
            got = [(int(r), int(o)) for r, o in zip(ops["run_length"][off[k]:off[k + 1]], ops["op"][off[k]:off[k + 1]])]
            ok = ok and (int(s[k]), got, int(ei[k]), int(ej[k])) == (es, er, oi, oj)
        t1 = time.perf_counter()
        nref = min(pairs, 2000)
        for k in range(nref):
            oracle.gsw_extend(side, sc, -600, a[k], b[k])
        cpu_pairs_s = nref / (time.perf_counter() - t1)
        cells = sum(len(x) * len(y) for x, y in zip(a, b))
        print(json.dumps({"series": name, "pairs": pairs, "cells": cells, "bit_exact_sample": bool(ok),
                          "host_call_s": dt, "pairs_per_s_host_call": pairs / dt, "kernel_ms": {"fill": tm["fill_ms"], "traceback": tm["traceback_ms"], "total": tm["total_ms"]},
                          "cells_per_s_kernels": cells / (tm["total_ms"] * 1e-3), "pairs_per_s_kernels": pairs / (tm["total_ms"] * 1e-3),
                          "cpu_oracle_pairs_per_s_1thread": cpu_pairs_s}))


if __name__ == "__main__":
    main()
