#!/usr/bin/env python3
"""gnx_gsw_map_reads: reads/s against the number of driver threads (the worker pool of the read path's host stages).  100 000 reads of 150
bases, 4 x 200 kb graph (round 4's shape) and a 100 Mb variation graph.  Usage: python tools/gsw_threads.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import common
from gonomics_amd import _lib, align
import bench_gsw_genome as bg

def main():
    L = _lib.lib(); _lib.check(L.gnx_init(0, 0))
    rng = np.random.default_rng(9)
    seqs = [rng.integers(0, 4, size=200000).astype(np.uint8) for _ in range(4)]
    reads = []
    for _ in range(100000):
        k = int(rng.integers(0, 4)); o = int(rng.integers(0, 200000 - 170))
        reads.append(common.mutate(rng, seqs[k][o:o + 170], 0.02, 0.01)[:150])
    rcat = np.concatenate(reads); roff = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
    g = _lib.GswGraph(seqs, [], 32, 32)
    for T in (1, 8, 16, 32, 64, 128):
        best = None
        for _ in range(4):
            t0 = time.perf_counter(); gir, _n, _c = g.map_reads((rcat, roff), align.HumanChimpTwoScoreMatrix, threads=T); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(json.dumps({"graph": "4 x 200 kb", "reads": 100000, "threads": T, "call_ms": round(best * 1e3, 2), "reads_per_s": round(100000 / best), "mapped": int((gir["aln_score"] > 0).sum())}), flush=True)
    g.close()
    cat, off, ef, et, n_sites = bg.build_graph(100000000)
    import ctypes
    hh = ctypes.c_void_p()
    _lib.check(L.gnx_gsw_graph_create(cat.ctypes.data, off.ctypes.data, off.shape[0] - 1, ef.ctypes.data, et.ctypes.data, ef.shape[0], 32, 32, ctypes.byref(hh)))
    g = _lib.GswGraph.__new__(_lib.GswGraph); g._h = hh
    rc2, ro2, _ = bg.sample_reads(np.random.default_rng(5), cat, off, n_sites, 100000)
    for T in (16, 64, 128):
        best = None
        for _ in range(3):
            t0 = time.perf_counter(); gir, _n, _c = g.map_reads((rc2, ro2), align.HumanChimpTwoScoreMatrix, threads=T); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(json.dumps({"graph": "100 Mb variation graph", "reads": 100000, "threads": T, "call_ms": round(best * 1e3, 2), "reads_per_s": round(100000 / best), "mapped": int((gir["aln_score"] > 0).sum())}), flush=True)
    g.close()

if __name__ == "__main__":
    main()
