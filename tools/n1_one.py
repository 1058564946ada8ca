import sys, os, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
from gonomics_amd import _lib, align
import bench_n1_cmd as b
recs = b.make_records(8, 30000, 3)
blocks = [np.asarray(r.Seq, dtype=np.uint8)[None, :] for r in recs]
p = _lib.make_params(_lib.GNX_AFFINE_GAP_HIGHMEM, align.HumanChimpTwoScoreMatrix, -300, -40)
for env in (None, "0"):
    if env is None: os.environ.pop("GNX_LAT", None)
    else: os.environ["GNX_LAT"] = env
    for prs in ([(0, 1)], [(x, y) for x in range(7) for y in range(x + 1, 8)]):
        for rep in range(2):
            t0 = time.perf_counter()
            sc, ops, off = _lib.multiple_affine_gap_batch(p, 3, blocks, prs)
            dt = time.perf_counter() - t0
        tm = _lib.get_timing()
        print("GNX_LAT", env, "pairs", len(prs), "call %.2f ms fill %.2f tb %.2f path %d" % (dt * 1e3, tm["fill_ms"], tm["traceback_ms"], tm["fast_path"]), int(sc[0]))
