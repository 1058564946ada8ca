#!/usr/bin/env python3
"""Where the latency geometry (GNX_LAT=2) stops paying against the general path (GNX_LAT=0): pairs of n x m, batch sizes doubling.
Usage: python tools/lat_crossover.py [affine|const]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from gonomics_amd import _lib, align

def main():
    fn = sys.argv[1] if len(sys.argv) > 1 else "affine"
    mode, go, ge = (_lib.GNX_AFFINE_GAP, -600, -150) if fn == "affine" else (_lib.GNX_CONST_GAP, -430, 0)
    p = _lib.make_params(mode, align.HumanChimpTwoScoreMatrix, go, ge)
    rng = np.random.default_rng(4)
    for n, m, counts in ((150, 10000, (64, 256, 512, 1024, 2048, 4096)), (1000, 1000, (32, 128, 256, 512, 1024)), (10000, 10000, (4, 8, 16, 32, 64)), (3000, 3000, (16, 64, 128, 256))):
        for cnt in counts:
            alphas, betas = [], []
            for _ in range(cnt):
                a = rng.integers(0, 4, size=n).astype(np.uint8)
                if m > 2 * n:
                    b = rng.integers(0, 4, size=m).astype(np.uint8); off = int(rng.integers(0, m - n)); a = common.mutate(rng, b[off:off + n + 20], sub=0.02, indel=0.005, geo=0.5)[:n]
                else:
                    b = common.mutate(rng, a, sub=0.02, indel=0.005, geo=0.5)
                    b = np.concatenate([b, rng.integers(0, 4, size=max(m - len(b), 0)).astype(np.uint8)])[:m]
                alphas.append(a); betas.append(b)
            row = {"fn": fn, "n": n, "m": m, "pairs": cnt, "strips128": cnt * ((n + 127) // 128)}
            for name, env in (("lat", "2"), ("other", "0")):
                os.environ["GNX_LAT"] = env
                best = None
                for rep in range(3):
                    _lib.align_batch(p, alphas, betas)
                    tm = _lib.get_timing()
                    if best is None or tm["total_ms"] < best["total_ms"]:
                        best = tm
                row[name + "_ms"] = round(best["total_ms"], 3); row[name + "_fill_ms"] = round(best["fill_ms"], 3); row[name + "_path"] = best["fast_path"]
            print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()
