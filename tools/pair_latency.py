#!/usr/bin/env python3
"""Latency of ONE gnx_align_pair call after another from one thread (what a Go loop over align.AffineGap / ConstGap sees), per pair shape.
Usage: python tools/pair_latency.py [calls]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from gonomics_amd import _lib, align  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 8 << 30))
    rng = np.random.default_rng(1)
    for mode, name, go, ge in ((_lib.GNX_AFFINE_GAP, "AffineGap", -600, -150), (_lib.GNX_CONST_GAP, "ConstGap", -430, 0)):
        p = _lib.make_params(mode, align.HumanChimpTwoScoreMatrix, go, ge)
        for n, m, related in ((150, 150, False), (150, 1000, False), (1000, 1000, False), (150, 10000, False), (3000, 3000, False), (1000, 1000, True), (9673, 10000, True), (20000, 100000, True)):
            a = rng.integers(0, 4, size=n).astype(np.uint8)
            b = rng.integers(0, 4, size=m).astype(np.uint8)
            if related:  # what the named commands align: two versions of one sequence (1 % substitutions, an indel every ~500 bases)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import common
                if m > 2 * n:  # a read inside a window
                    off = (m - n) // 2
                    a = common.mutate(rng, b[off:off + n + n // 50], sub=0.01, indel=0.002, geo=0.5)[:n]
                else:
                    b = common.mutate(rng, a, sub=0.01, indel=0.002, geo=0.5)
                    b = np.concatenate([b, rng.integers(0, 4, size=max(m - len(b), 0)).astype(np.uint8)])[:m]
            if n * m > 5e8:
                calls_here = 10
            else:
                calls_here = calls
            sc, no, ops = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_void_p()
            ts = []
            for k in range(calls_here + 20 if n * m <= 5e8 else calls_here + 2):
                t0 = time.perf_counter()
                rc = L.gnx_align_pair(ctypes.byref(p), a.ctypes.data, n, b.ctypes.data, m, ctypes.byref(sc), ctypes.byref(ops), ctypes.byref(no))
                dt = time.perf_counter() - t0
                _lib.check(rc)
                L.gnx_free(ops)
                if k >= (20 if n * m <= 5e8 else 2):
                    ts.append(dt)
            ts = np.asarray(ts) * 1e6
            tm = _lib.get_timing()
            print(json.dumps({"series": "one %s pair per call, one thread" % name, "n": n, "m": m, "related": related, "us_per_call_median": float(np.median(ts)), "us_p10": float(np.percentile(ts, 10)),
                              "us_p90": float(np.percentile(ts, 90)), "device_ms_of_last_call": tm["total_ms"], "fill_ms": tm["fill_ms"], "traceback_ms": tm["traceback_ms"], "path": tm["fast_path"]}), flush=True)


if __name__ == "__main__":
    main()
