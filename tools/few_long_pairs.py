#!/usr/bin/env python3
"""A handful of long pairs in ONE batch call: the 16-lane snapshot kernels (four pairs per wave, walks of one wave per pair) against the
64-lane kernels + walk farm (GNX_W64=2).  python tools/few_long_pairs.py [affine|const] n pairs...   -> one JSON line per (pairs, route)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from gonomics_amd import _lib, align  # noqa: E402


def main():
    affine = (sys.argv[1] if len(sys.argv) > 1 else "affine") == "affine"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    extra = int(os.environ.get("FLP_EXTRA", "0"))  # random columns appended to every beta (a read against a longer window)
    counts = [int(x) for x in sys.argv[3:]] or [2, 4, 8]
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 0))
    rng = np.random.default_rng(99)
    sc = align.HumanChimpTwoScoreMatrix
    p = _lib.make_params(_lib.GNX_AFFINE_GAP, sc, -600, -150) if affine else _lib.make_params(_lib.GNX_CONST_GAP, sc, -430)
    alphas, betas = [], []
    for _ in range(max(counts)):
        a = rng.integers(0, 4, size=n).astype(np.uint8)
        b = common.mutate(rng, a, 0.03, 0.01, geo=0.4)
        if extra:
            b = np.concatenate([rng.integers(0, 4, size=extra // 2).astype(np.uint8), b, rng.integers(0, 4, size=extra - extra // 2).astype(np.uint8)])
        alphas.append(a); betas.append(b)
    for k in counts:
        ref = None
        for w64 in (None, "2"):
            if w64 is None:
                os.environ.pop("GNX_W64", None)
            else:
                os.environ["GNX_W64"] = w64
            os.environ["GNX_CLONG"] = "2"
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                got = _lib.align_batch(p, alphas[:k], betas[:k])
                dt = time.perf_counter() - t0
                tm = _lib.get_timing()
                best = dt if best is None else min(best, dt)
            same = True
            if ref is None:
                ref = got
            else:
                same = bool(np.array_equal(ref[0], got[0]) and np.array_equal(ref[1]["run_length"], got[1]["run_length"]) and np.array_equal(ref[1]["op"], got[1]["op"]))
            print(json.dumps({"fn": "AffineGap" if affine else "ConstGap", "n": n, "m": int(betas[0].shape[0]), "pairs": k, "GNX_W64": w64 or "default", "route": tm["fast_path"], "call_s": round(best, 4),
                              "sweep_ms": round(tm["fill_ms"], 2), "walk_ms": round(tm["traceback_ms"], 2), "equal_to_first_route": same}), flush=True)


if __name__ == "__main__":
    main()
