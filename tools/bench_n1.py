#!/usr/bin/env python3
"""Measurement for the N1 variants (not the headline metric): AffineGapChunk on tandem-repeat-like pairs and one
progressive-alignment round (all x<y group pairs) of multipleAffineGap.  Prints one JSON line per series."""
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
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 32 << 30))
    rng = np.random.default_rng(3)
    chunk, na, nb, pairs = 3, 160, 3000, 4096
    alphas, betas = [], []
    for _ in range(pairs):
        unit = rng.integers(0, 4, size=chunk).astype(np.uint8)
        b = np.tile(unit, nb)
        b[rng.random(b.size) < 0.03] = rng.integers(0, 4)
        a = b[int(rng.integers(0, nb - na)) * chunk:][:na * chunk].copy()
        a[rng.random(a.size) < 0.03] = rng.integers(0, 4)
        alphas.append(a); betas.append(b)
    p = _lib.make_params(_lib.GNX_AFFINE_GAP_HIGHMEM, align.HumanChimpTwoScoreMatrix, -600, -150)
    _lib.affine_gap_chunk_batch(p, chunk, alphas[:64], betas[:64])
    t0 = time.perf_counter()
    sc, ops, off = _lib.affine_gap_chunk_batch(p, chunk, alphas, betas)
    dt = time.perf_counter() - t0
    tm = _lib.get_timing()
    ok = all((int(sc[k]), [(int(r), int(o)) for r, o in zip(ops["run_length"][off[k]:off[k + 1]], ops["op"][off[k]:off[k + 1]])])
             == oracle.affine_gap_chunk(align.HumanChimpTwoScoreMatrix, -600, -150, chunk, alphas[k], betas[k]) for k in range(0, pairs, 257))
    cells = pairs * na * nb
    print(json.dumps({"series": "AffineGapChunk", "pairs": pairs, "chunk_cells_per_pair": na * nb, "chunk_size": chunk,
                      "host_call_s": dt, "fill_ms": tm["fill_ms"], "traceback_ms": tm["traceback_ms"],
                      "chunk_cells_per_s_kernel": cells / (tm["fill_ms"] * 1e-3), "chunk_cells_per_s_call": cells / dt, "bit_exact_sample": ok}))
    # one progressive round over 64 groups of 3 sequences x 400 columns
    groups = []
    base = rng.integers(0, 4, size=400).astype(np.uint8)
    for _ in range(64):
        blk = np.stack([common.mutate(rng, base, sub=0.05, indel=0.0)[:400] for _ in range(3)])
        groups.append(blk)
    prs = [(x, y) for x in range(64) for y in range(x + 1, 64)]
    p = _lib.make_params(_lib.GNX_AFFINE_GAP_HIGHMEM, align.DefaultScoreMatrix, -400, -30)
    _lib.multiple_affine_gap_batch(p, 1, groups, prs[:8])
    t0 = time.perf_counter()
    sc, ops, off = _lib.multiple_affine_gap_batch(p, 1, groups, prs)
    dt = time.perf_counter() - t0
    tm = _lib.get_timing()
    k = 777
    ok = (int(sc[k]), [(int(r), int(o)) for r, o in zip(ops["run_length"][off[k]:off[k + 1]], ops["op"][off[k]:off[k + 1]])]) == \
        oracle.multiple_affine_gap(align.DefaultScoreMatrix, -400, -30, 1, groups[prs[k][0]], groups[prs[k][1]])
    print(json.dumps({"series": "multipleAffineGap round (64 groups x 3 seqs x 400 cols)", "pairs": len(prs), "host_call_s": dt,
                      "fill_ms": tm["fill_ms"], "cells_per_s_call": len(prs) * 400 * 400 / dt, "bit_exact_sample": ok}))


if __name__ == "__main__":
    main()
