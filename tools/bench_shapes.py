#!/usr/bin/env python3
"""Kernel time of one batch per shape and path (device-resident inputs): where does the constant-gap path without a stored direction matrix
(GNX_CLONG) / the affine fast path beat the general path?  Usage: python tools/bench_shapes.py [const|affine|local] [n,m,pairs]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gonomics_amd import _lib, align  # noqa: E402
import bench  # noqa: E402


def main():
    import torch
    kind = sys.argv[1] if len(sys.argv) > 1 else "const"
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 120 << 30))
    dev = torch.device("cuda", 0)
    mode, go, ge = (_lib.GNX_CONST_GAP, -430, 0) if kind == "const" else ((_lib.GNX_AFFINE_GAP_LOCAL if kind == "local" else _lib.GNX_AFFINE_GAP), -600, -150)
    p = _lib.make_params(mode, align.HumanChimpTwoScoreMatrix, go, ge)
    shapes = ((150, 10000, 65536), (250, 10000, 40000), (320, 10000, 32768), (480, 10000, 20000), (800, 10000, 12000), (1600, 10000, 8192 if kind == "affine" else 6000), (3200, 10000, 3000), (1000, 1200, 100000)) + (((20000, 100000, 256),) if kind == "affine" else ())
    if kind == "local":
        shapes = ((150, 10000, 65536), (250, 10000, 40000), (800, 10000, 12000), (3200, 10000, 3000))
    if len(sys.argv) > 2:  # one shape: n,m,pairs
        shapes = (tuple(int(x) for x in sys.argv[2].split(",")),)
    for n, m, pairs in shapes:
        reads, chunk = bench.make_workload(3, pairs, read_len=n, chunk_len=m)
        d_reads = torch.from_numpy(reads.reshape(-1)).to(dev); d_chunk = torch.from_numpy(chunk).to(dev)
        h_al = np.full(pairs, n, dtype=np.int64); h_bl = np.full(pairs, m, dtype=np.int64)
        d_as = torch.arange(pairs, dtype=torch.int64, device=dev) * n
        d_bs = torch.zeros(pairs, dtype=torch.int64, device=dev)
        d_score = torch.zeros(pairs, dtype=torch.int64, device=dev); d_off = torch.zeros(pairs + 1, dtype=torch.int64, device=dev)
        cap = pairs * (n + m + 2) if kind == "const" else pairs * 64
        cap = min(cap, 1 << 30)
        d_ops = torch.zeros(cap * 16, dtype=torch.uint8, device=dev)
        tot = ctypes.c_int64()
        row = {"kind": kind, "n": n, "m": m, "pairs": pairs}
        res = {}
        variants = [("general", {"GNX_CLONG": "0", "GNX_FASTPATH": "0"}), ("default", {})]
        if kind == "const":
            variants.append(("snapshot", {"GNX_CLONG": "2"}))  # the snapshot path whatever the routing rule says
        if kind != "const" and n <= 20480 and m >= 16:
            variants.append(("row_blocks", {"GNX_FASTPATH": "2"}))  # the fast path whatever the routing rule says
        for name, env in variants:
            for k in ("GNX_CLONG", "GNX_FASTPATH"):
                os.environ.pop(k, None)
            os.environ.update(env)
            best = None
            for it in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if kind == "local":  # AffineGapLocal(target = chunk, query = read)
                    rc = L.gnx_align_batch_device(ctypes.byref(p), pairs, d_chunk.data_ptr(), d_bs.data_ptr(), 0, d_reads.data_ptr(), d_as.data_ptr(), 0,
                                                  h_bl.ctypes.data, h_al.ctypes.data, d_score.data_ptr(), d_ops.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(tot), None)
                else:
                    rc = L.gnx_align_batch_device(ctypes.byref(p), pairs, d_reads.data_ptr(), d_as.data_ptr(), 0, d_chunk.data_ptr(), d_bs.data_ptr(), 0,
                                                  h_al.ctypes.data, h_bl.ctypes.data, d_score.data_ptr(), d_ops.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(tot), None)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                _lib.check(rc)
                tm = _lib.get_timing()
                if it and (best is None or dt < best[0]):
                    best = (dt, tm)
            res[name] = (d_score.cpu().numpy().copy(), d_off.cpu().numpy().copy())
            row[name] = {"ms": best[0] * 1e3, "fill_ms": best[1]["fill_ms"], "tb_ms": best[1]["traceback_ms"], "path": best[1]["fast_path"],
                         "cells_per_s": pairs * n * m / best[0]}
        row["same_results"] = bool(np.array_equal(res["general"][0], res["default"][0]) and np.array_equal(res["general"][1], res["default"][1]))
        print(json.dumps(row), flush=True)
        del d_ops, d_reads
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
