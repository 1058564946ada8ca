#!/usr/bin/env python3
"""Which path is wrong?  One shape, constant gap: default routing vs GNX_CLONG=0 (stored matrix) vs the oracle on a sample; repeated to
catch races.  Usage: python tools/check_shape.py n m pairs [repeats]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gonomics_amd import _lib, align
import bench, oracle
n, m, pairs = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
L = _lib.lib(); _lib.check(L.gnx_init(0, 120 << 30))
reads, chunk = bench.make_workload(3, pairs, read_len=n, chunk_len=m)
p = _lib.make_params(_lib.GNX_CONST_GAP, align.HumanChimpTwoScoreMatrix, -430, 0)
a_start = np.arange(pairs, dtype=np.int64) * n; a_len = np.full(pairs, n, dtype=np.int64)
b_start = np.zeros(pairs, dtype=np.int64); b_len = np.full(pairs, m, dtype=np.int64)
k = min(pairs, 48)
sel = np.linspace(0, pairs - 1, k).astype(np.int64)
exp = oracle.align_batch(1, align.HumanChimpTwoScoreMatrix, -430, 0, [reads[x] for x in sel], [chunk] * k, 10000, 10000, threads=16)
for rep in range(reps):
    for name, env in (("default", {}), ("stored", {"GNX_CLONG": "0"})):
        os.environ.pop("GNX_CLONG", None); os.environ.update(env)
        sc, ops, off = _lib.align_batch_windows(p, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len)
        bad = [int(x) for q, x in enumerate(sel) if sc[x] != exp[0][q] or (off[x + 1] - off[x]) != (exp[2][q + 1] - exp[2][q])]
        print(rep, name, "path", _lib.get_timing()["fast_path"], "mismatching sampled pairs:", bad[:10], "of", k, flush=True)
        if name == "default": d = (sc.copy(), off.copy())
        else:
            diff = np.nonzero((sc != d[0]) | (np.diff(off) != np.diff(d[1])))[0]
            print("   default vs stored differ at", diff[:10], "count", len(diff), flush=True)
