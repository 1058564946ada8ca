#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (SURVEY 8d's definition of the metric: H2D of reads + kernels + D2H of scores /
CIGARs) next to the device-resident rate of the same batch, in one process:
  gnx_align_batch_windows    reads + one shared 10 kb chunk as host buffers (config C2)
  gnx_align_batch_by_offset  reads against windows of the resident reference (configs C3 / C4), reference set once
  gnx_align_batch_device     inputs and outputs in HBM (what bench.py's `value` measures)
The entry points are called through raw ctypes (no numpy copies of the results inside the timed region); the library's own wall
clock (gnx_timing.host_ms) is reported beside the caller's.  Usage on the GPU box: python tools/bench_host.py [n_pairs ...]"""
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
    sizes = [int(x) for x in sys.argv[1:]] or [100000, 1000000]
    import torch
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 150 << 30))
    p = _lib.make_params(_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150)
    dev = torch.device("cuda", 0)
    for n_pairs in sizes:
        reads, chunk = bench.make_workload(2, n_pairs)
        a_buf = np.ascontiguousarray(reads.reshape(-1))
        a_off = np.arange(n_pairs + 1, dtype=np.int64) * 150
        a_len = np.full(n_pairs, 150, dtype=np.int64)
        b_start = np.zeros(n_pairs, dtype=np.int64)
        b_len = np.full(n_pairs, chunk.shape[0], dtype=np.int64)
        scores = np.zeros(n_pairs, dtype=np.int64)
        cells = n_pairs * 150 * chunk.shape[0]

        def call(which):
            ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
            t0 = time.perf_counter()
            if which == "windows":
                rc = L.gnx_align_batch_windows(ctypes.byref(p), n_pairs, a_buf.ctypes.data, a_buf.shape[0], a_off.ctypes.data, a_len.ctypes.data,
                                               chunk.ctypes.data, chunk.shape[0], b_start.ctypes.data, b_len.ctypes.data,
                                               scores.ctypes.data, ctypes.byref(ops_p), ctypes.byref(off_p))
            else:
                rc = L.gnx_align_batch_by_offset(ctypes.byref(p), n_pairs, a_buf.ctypes.data, a_off.ctypes.data, b_start.ctypes.data, b_len.ctypes.data,
                                                 scores.ctypes.data, ctypes.byref(ops_p), ctypes.byref(off_p))
            dt = time.perf_counter() - t0
            _lib.check(rc)
            total = int(ctypes.cast(off_p, ctypes.POINTER(ctypes.c_int64))[n_pairs])
            L.gnx_free(ops_p); L.gnx_free(off_p)
            return dt, _lib.get_timing(), total

        res = {}
        _lib.set_reference(chunk)
        for which in ("windows", "by_offset"):
            runs = [call(which) for _ in range(5)][1:]
            dt, tm, total = min(runs, key=lambda r: r[0])
            res[which] = {"host_call_ms": dt * 1e3, "library_wall_ms": tm["host_ms"], "first_upload_ms": tm["stage0_ms"], "device_ms": tm["total_ms"],
                          "gather_d2h_ms": tm["fetch_ms"], "cells_per_s": cells / dt, "pairs_per_s": n_pairs / dt,
                          "bytes_h2d": int(a_buf.nbytes + 16 * n_pairs + (chunk.nbytes if which == "windows" else 0)),
                          "bytes_d2h": int(16 * n_pairs + 8 + 16 * total)}
        # device-resident
        d_reads = torch.from_numpy(a_buf).to(dev); d_chunk = torch.from_numpy(chunk).to(dev)
        d_as = torch.from_numpy(a_off[:-1].copy()).to(dev); d_al = torch.from_numpy(a_len).to(dev)
        d_bs = torch.from_numpy(b_start).to(dev); d_bl = torch.from_numpy(b_len).to(dev)
        d_score = torch.zeros(n_pairs, dtype=torch.int64, device=dev); d_off = torch.zeros(n_pairs + 1, dtype=torch.int64, device=dev)
        cap = 48 * n_pairs
        d_ops = torch.zeros(cap * 16, dtype=torch.uint8, device=dev)
        tot = ctypes.c_int64()
        stream = torch.cuda.current_stream().cuda_stream
        best = 1e9
        for it in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(L.gnx_align_batch_device(ctypes.byref(p), n_pairs, d_reads.data_ptr(), d_as.data_ptr(), d_al.data_ptr(), d_chunk.data_ptr(), d_bs.data_ptr(), d_bl.data_ptr(),
                                                a_len.ctypes.data, b_len.ctypes.data, d_score.data_ptr(), d_ops.data_ptr(), cap, d_off.data_ptr(), ctypes.byref(tot), ctypes.c_void_p(stream)))
            torch.cuda.synchronize()
            if it:
                best = min(best, time.perf_counter() - t0)
        res["device_resident"] = {"call_ms": best * 1e3, "cells_per_s": cells / best}
        for which in ("windows", "by_offset"):
            res[which]["vs_device_resident"] = res[which]["cells_per_s"] / res["device_resident"]["cells_per_s"]
        print(json.dumps({"series": "host-buffer entry points (PCIe-inclusive) vs device-resident, C2 AffineGap", "pairs": n_pairs, **res}))
        del d_reads, d_ops
        torch.cuda.empty_cache()


def c3_series(n_pairs=1 << 20, ref_len=3200000000, window=10000):
    """config C3: reads at uniform offsets of a 3.2e9-base reference that is generated on the device and stays there; ONE
    gnx_align_batch_by_offset call per batch of 1 Mi reads (host buffers in, host buffers out)"""
    import test_host_entry as T
    L = _lib.lib()
    seed = 33
    _lib.check(L.gnx_set_reference_synthetic(ref_len, seed))
    reads, starts = T.c3_reads(34, n_pairs, ref_len, seed, window)
    p = _lib.make_params(_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150)
    a_buf = np.ascontiguousarray(reads.reshape(-1))
    a_off = np.arange(n_pairs + 1, dtype=np.int64) * 150
    lens = np.full(n_pairs, window, dtype=np.int64)
    scores = np.zeros(n_pairs, dtype=np.int64)
    best = None
    for it in range(3):
        ops_p, off_p = ctypes.c_void_p(), ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = L.gnx_align_batch_by_offset(ctypes.byref(p), n_pairs, a_buf.ctypes.data, a_off.ctypes.data, starts.ctypes.data, lens.ctypes.data,
                                         scores.ctypes.data, ctypes.byref(ops_p), ctypes.byref(off_p))
        dt = time.perf_counter() - t0
        _lib.check(rc)
        L.gnx_free(ops_p); L.gnx_free(off_p)
        tm = _lib.get_timing()
        if it and (best is None or dt < best[0]):
            best = (dt, tm)
    dt, tm = best
    print(json.dumps({"series": "C3: %d reads of 150 b at uniform offsets of a resident %.1e-base reference, one gnx_align_batch_by_offset call" % (n_pairs, ref_len),
                      "host_call_ms": dt * 1e3, "library_wall_ms": tm["host_ms"], "device_ms": tm["total_ms"], "first_upload_ms": tm["stage0_ms"], "gather_d2h_ms": tm["fetch_ms"],
                      "cells_per_s": n_pairs * 150 * window / dt, "pairs_per_s": n_pairs / dt, "windows_beyond_2GB": int((starts > (1 << 31)).sum())}), flush=True)


if __name__ == "__main__":
    main()
    c3_series()
