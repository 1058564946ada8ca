"""Pairs beyond the static int32 range of the kernels' keys: generator, oracle fixtures, GPU timings.

  python tools/long_pairs.py oracle <case> ...   CPU ORACLE on the seeded pair (minutes to half an hour of one core; run in the build
                                                 container) -> tests/golden/long_pairs.json: score, number of runs, sha256 of the runs
  python tools/long_pairs.py gpu [<case> ...]    the same pairs (and the megabase pairs no oracle finishes) through the library, timed:
                                                 one JSON line per pair (profiles/r5_long_pairs.jsonl); consumed / re-scored / fixture checks

Cases are (function, n, m) with the callers' parameters (HumanChimpTwo, -600 / -150 resp. -430, 10 000 x 10 000 checkerboards,
cmd/cigarToBed/cigarToBed.go:86).  tests/test_long_range.py imports gen() and the fixture.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
FIXTURE = os.environ.get("LONG_PAIRS_FIXTURE") or os.path.join(ROOT, "tests", "golden", "long_pairs.json")

# name -> (affine, n, extra columns, seed).  The static bound: 4 * (score - e (i + j)) grows by up to 4 * (100 + 2 * 150) per diagonal step
# (ConstGap: 4 * (100 + 2 * 430)), so the keys pass 2^29 from min(n, m) = 335 544 (ConstGap: 139 810) on.
CASES = {
    "affine_340k": (True, 340000, 0, 501),
    "const_150k": (False, 150000, 30000, 502),
    "affine_1M": (True, 1000000, 0, 1234),
    "const_300k_2M": (False, 300000, 0, 1235),
    "affine_2M": (True, 2000000, 0, 1236),      # 4e12 cells: bottom rows + snapshots of all strips would be 500 GB -> row panels (run_device_mega)
    "affine_5M": (True, 5000000, 0, 1237),      # 2.5e13 cells: the size of the cmd/cigarToBed fixture the reference ships (.MISSING_LARGE_BLOBS:1-3)
    # quirk Q1 on purpose (align/affineGap.go:305): beta lacks the bases of alpha around EVERY row 10 000 k, so the walk crosses each checkerboard
    # edge upwards inside a D run and restarts in the argmax state of the entry cell (29 crossings; the CIGAR re-scores below the score)
    "affine_q1_300k": (True, 300000, 0, 1238),
}
DEFAULT_GPU_CASES = ("const_150k", "affine_340k", "affine_q1_300k", "affine_1M", "const_300k_2M", "affine_2M")
ORACLE_CASES = ("affine_340k", "const_150k", "affine_q1_300k", "affine_1M")


def gen(name):
    import common
    affine, n, extra, seed = CASES[name]
    rng = np.random.default_rng(seed)
    if name == "const_300k_2M":
        win = rng.integers(0, 4, size=2000000).astype(np.uint8)
        a = common.mutate(rng, win[700000:700000 + n + 3000], sub=0.03, indel=0.004, geo=0.5)[:n]
        return affine, a, win
    a = rng.integers(0, 4, size=n).astype(np.uint8)
    if name == "affine_q1_300k":
        keep = np.ones(n, dtype=bool)
        for k in range(1, (n - 5000) // 10000 + 1):
            run = int(rng.integers(3, 120))
            below = int(rng.integers(1, run))  # rows of the run below the edge (alpha[10 000 k] is row 10 000 k + 1)
            keep[10000 * k - (run - below):10000 * k + below] = False
        return affine, a, common.mutate(rng, a[keep], sub=0.01, indel=0.0005, geo=0.4)
    b = common.mutate(rng, a, sub=0.02, indel=0.002, geo=0.4)
    if extra:
        b = np.concatenate([b, rng.integers(0, 4, size=extra).astype(np.uint8)])
    return affine, a, b


def digest(score, ops):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(ops["run_length"]).astype("<i8").tobytes())
    h.update(np.ascontiguousarray(ops["op"]).astype("u1").tobytes())
    return {"score": int(score), "runs": int(ops.shape[0]), "sha256": h.hexdigest()}


def params(affine):
    from gonomics_amd import align
    return (align.HumanChimpTwoScoreMatrix, -600, -150) if affine else (align.HumanChimpTwoScoreMatrix, -430, 0)


def run_oracle(names):
    import oracle
    for name in names:
        affine, a, b = gen(name)
        sc, go, ge = params(affine)
        t0 = time.time()
        s, ops, off = oracle.align_batch(oracle.MODE_AFFINE if affine else oracle.MODE_CONST, sc, go, ge, [a], [b], 10000, 10000, threads=1)
        d = digest(s[0], ops)
        d.update({"n": int(a.shape[0]), "m": int(b.shape[0]), "oracle_s": round(time.time() - t0, 1)})
        fx = json.load(open(FIXTURE)) if os.path.exists(FIXTURE) else {}  # (read again: several of these may run side by side, hours each)
        fx[name] = d
        print(name, d, flush=True)
        with open(FIXTURE, "w") as fh:
            json.dump(fx, fh, indent=1, sort_keys=True)


def gpu_rows(names, reps=None):
    """one row (dict) per case: the pair through the library, timed, with the checks a CIGAR of that size admits"""
    from gonomics_amd import _lib
    from test_const_long import rescore_const
    from test_long_range import rescore_affine
    fx = json.load(open(FIXTURE)) if os.path.exists(FIXTURE) else {}
    _lib.check(_lib.lib().gnx_init(0, 0))
    for name in names:
        affine, a, b = gen(name)
        sc, go, ge = params(affine)
        highmem = os.environ.get("LONG_PAIRS_HIGHMEM") == "1"  # AffineGap_highMem / ConstGap_highMem semantics: no checkerboard quirks
        p = _lib.make_params((_lib.GNX_AFFINE_GAP_HIGHMEM if affine else _lib.GNX_CONST_GAP_HIGHMEM) if highmem else (_lib.GNX_AFFINE_GAP if affine else _lib.GNX_CONST_GAP), sc, go, ge, 10000, 10000)
        best, first = None, 0.0
        _lib.debug_counter(3, reset=True)  # (3 and 4 are one pair of counters: a reset zeroes both)
        ncalls = 0
        for rep in range(reps or (3 if a.shape[0] * b.shape[0] < 2e12 else 2)):  # (the first call of a process also allocates its workspace: ~27 ms per GB)
            t0 = time.perf_counter()
            score, ops, off = _lib.align_batch(p, [a], [b])
            wall = time.perf_counter() - t0
            tm = _lib.get_timing()
            ncalls += 1
            first = wall if rep == 0 else first
            if best is None or wall < best[0]:
                best = (wall, tm)
        wall, tm = best
        q1c, q1n = _lib.debug_counter(3, reset=False) // ncalls, _lib.debug_counter(4, reset=True) // ncalls  # quirk-Q1 restarts of ONE call: that changed the state / all
        ni, nj, total = rescore_affine(a, b, ops, sc, go, ge) if affine else rescore_const(a, b, ops, sc, go)
        row = {"case": name, "fn": "AffineGap(HumanChimpTwo,-600,-150)" if affine else "ConstGap(HumanChimpTwo,-430)", "n": int(a.shape[0]), "m": int(b.shape[0]),
               "cells": int(a.shape[0]) * int(b.shape[0]), "call_s": round(wall, 4), "first_call_s": round(first, 4), "sweep_ms": round(tm["fill_ms"], 2), "walk_ms": round(tm["traceback_ms"], 2),
               "cells_per_s_call": float("%.4g" % (a.shape[0] * b.shape[0] / wall)), "cells_per_s_kernels": float("%.4g" % (a.shape[0] * b.shape[0] / (tm["total_ms"] * 1e-3))),
               "workspace_bytes": int(tm["trace_bytes"]), "route": {2: "snapshot path", 5: "row panels", 6: "snapshot path, 64 lanes per pair"}.get(int(tm["fast_path"]), int(tm["fast_path"])), "launches": int(tm["n_launches"]), "score": int(score[0]), "runs": int(ops.shape[0]),
               "consumes_n_m": (ni, nj) == (a.shape[0], b.shape[0]), "rescored_equals_score": total == int(score[0]), "rescored_minus_score": total - int(score[0]),
               "q1_restarts": int(q1n), "q1_restarts_changed": int(q1c),
               "semantics": "highMem (no checkerboards)" if highmem else "10 000 x 10 000 checkerboards (quirk Q1 can cost the CIGAR a gap open: the reference's own behaviour)"}
        if int(tm["fast_path"]) in (5, 6):
            row["rows_per_lane"], row["snapshot_steps"] = int(_lib.debug_counter(5, reset=False)), int(_lib.debug_counter(6, reset=False))
        if name in fx and not highmem:
            row["equals_oracle"] = digest(score[0], ops) == {k: fx[name][k] for k in ("score", "runs", "sha256")}
        row["ok"] = row_ok(row)
        yield row


def row_ok(r, gap_open=600):
    """what a row must satisfy: the oracle's digest where there is one; always: the CIGAR consumes both sequences and re-scores to the score minus at most one gap
    open per quirk-Q1 restart that changed the walk's state (align/affineGap.go:305: the walk re-enters the tile above in the argmax state X of the entry cell
    instead of the traced gap state -- the part of the gap below the edge is paid as a new gap, a deficit of gapOpen - (X - D) in [0, gapOpen]) -- so exactly
    to the score when the walk reports none (ConstGap: always; highMem semantics: always)"""
    if not r["consumes_n_m"] or not r.get("equals_oracle", True):
        return False
    deficit = -r["rescored_minus_score"]
    return 0 <= deficit <= gap_open * r["q1_restarts_changed"]


def run_gpu(names):
    for row in gpu_rows(names):
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1]
    names = sys.argv[2:]
    if mode == "oracle":
        run_oracle(names or ORACLE_CASES)
    else:
        run_gpu(names or list(DEFAULT_GPU_CASES))
