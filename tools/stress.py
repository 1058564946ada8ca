#!/usr/bin/env python3
"""Randomised GPU-vs-oracle stress (not part of the pytest suite): many batch shapes, matrices, penalties and
checkerboard sizes, biased towards the fast path (one row block and several).  Usage: python tools/stress.py [seconds] [seed]"""
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
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 16 << 30))
    mats = list(common.matrices().items())
    t0 = time.time()
    rounds = fp_rounds = 0
    while time.time() - t0 < budget:
        name, mx = mats[int(rng.integers(len(mats)))]
        go = int(rng.choice([0, -1, -30, -400, -600, -900]))
        ge = int(rng.choice([-1, -30, -55, -150, -400]))
        kind = int(rng.integers(0, 10))
        n = int(rng.integers(1, 161))
        cnt = int(rng.integers(1, 40))
        for k in ("GNX_FASTPATH", "GNX_FP_MAXIT", "GNX_NO_PIPE", "GNX_WALK_LANE", "GNX_FP_SPEC"):
            os.environ.pop(k, None)
        if kind >= 7:  # reads of several row blocks on the (forced) fast path: mixed numbers of blocks, both orientations, forced straggler rounds
            os.environ["GNX_FASTPATH"] = "2"
            if rng.random() < 0.3:
                os.environ["GNX_FP_MAXIT"] = str(int(rng.choice([0, 1, 3])))
            if rng.random() < 0.2:
                os.environ["GNX_NO_PIPE"] = "1"
            if rng.random() < 0.25:
                os.environ["GNX_WALK_LANE"] = "1"
            elif rng.random() < 0.3:
                os.environ["GNX_FP_SPEC"] = str(int(rng.choice([1, 2, 3])))
            n_top = int(rng.choice([200, 320, 500, 800, 1300]))
            uniform = rng.random() < 0.4
            m = int(rng.integers(300, 5000))
            chunk = rng.integers(0, 5 if rng.random() < 0.3 else 4, size=m + 1700).astype(np.uint8)
            alphas, betas = [], []
            for _ in range(cnt):
                n = int(rng.integers(max(1, n_top - 150), n_top + 1)) if uniform else int(rng.integers(1, n_top + 1))
                off = int(rng.integers(0, m - 1))
                src = chunk[off:off + n + 60]
                a = common.mutate(rng, src, sub=float(rng.choice([0.0, 0.02, 0.15])), indel=float(rng.choice([0.0, 0.01, 0.08])), geo=0.4)
                if len(a) < n:
                    a = np.concatenate([a, rng.integers(0, 4, size=n - len(a)).astype(np.uint8)])
                alphas.append(a[:n])
                betas.append(chunk[:m] if rng.random() < 0.7 else chunk[int(rng.integers(0, 300)):][:m])
            mode = int(rng.choice([0, 0, 2, 3, 3]))
            if mode == 3:
                alphas, betas = betas, alphas
        elif kind <= 2:  # fast-path shape: short alpha (uniform or mixed lengths), long beta (shared chunk or per-pair windows)
            mixed = rng.random() < 0.5
            n_top = n
            m = int(rng.integers(768, 6000))
            chunk = rng.integers(0, 5 if rng.random() < 0.3 else 4, size=m + 400).astype(np.uint8)
            alphas, betas = [], []
            for _ in range(cnt):
                if mixed:
                    n = int(rng.integers(1, n_top + 1))
                off = int(rng.integers(0, m - 1))
                src = chunk[off:off + n + 40]
                a = common.mutate(rng, src, sub=float(rng.choice([0.0, 0.02, 0.15])), indel=float(rng.choice([0.0, 0.01, 0.08])), geo=0.4)
                if len(a) < n:
                    a = np.concatenate([a, rng.integers(0, 4, size=n - len(a)).astype(np.uint8)])
                alphas.append(a[:n])
                betas.append(chunk[:m] if rng.random() < 0.7 else chunk[int(rng.integers(0, 300)):][:m])
            mode = int(rng.choice([0, 0, 2, 3, 3]))
            if mode == 3:  # AffineGapLocal(target, query): the transposed fast path
                alphas, betas = betas, alphas
        elif kind == 6:  # a few long pairs: pipelined strips + wave-cooperative traceback (general path)
            cnt = int(rng.integers(1, 6))
            alphas, betas = common.random_pairs(int(rng.integers(1 << 30)), cnt, 1, 1500, 1024, 2500, related=0.85)
            mode = int(rng.integers(0, 5))
        else:
            alphas, betas = common.random_pairs(int(rng.integers(1 << 30)), cnt, 1, 300, 1, 900, related=0.7)
            mode = int(rng.integers(0, 5))
        cs = int(rng.choice([10000, 10000, 64, 100, 257, 1000]))
        p = _lib.make_params(mode, mx, go, ge, cs, cs)
        try:
            got = _lib.align_batch(p, alphas, betas)
        except _lib.GnxError as ex:
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/stress_fail.npz", mode=mode, go=go, ge=ge, cs=cs, mx=np.asarray(mx),
                     alphas=np.array(alphas, dtype=object), betas=np.array(betas, dtype=object))
            print("ERROR", ex, "kind", kind, "mode", mode, name, go, ge, "cs", cs, "cnt", len(alphas), "n", [len(a) for a in alphas][:8], "m", [len(b) for b in betas][:8])
            sys.exit(1)
        fp_rounds += 1 if _lib.get_timing()["fast_path"] == 1 else 0
        exp = oracle.align_batch(mode, mx, go, ge, alphas, betas, cs, cs, threads=8)
        try:
            common.assert_same(got, exp)
        except AssertionError as e:
            np.savez("gpurun_out/stress_fail.npz", mode=mode, go=go, ge=ge, cs=cs, mx=np.asarray(mx),
                     alphas=np.array(alphas, dtype=object), betas=np.array(betas, dtype=object))
            print("MISMATCH", e, "mode", mode, name, go, ge, "cs", cs, "n", n, "cnt", cnt)
            sys.exit(1)
        rounds += 1
    print("stress ok: %d rounds (%d on the fast path) in %.0f s, seed %d" % (rounds, fp_rounds, time.time() - t0, seed))


if __name__ == "__main__":
    main()
