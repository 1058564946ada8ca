#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: DP cells/s (+ aligned pairs/s) of the
affine-gap 150 bp x 10 kb batch (config C2: faChunkAlign-style reads vs one 10 kb chunk,
align.AffineGap(read, chunk, HumanChimpTwoScoreMatrix, -600, -150)).

A "step" = one pass of the hot path (fill kernel + traceback kernels, CIGARs emitted) over one batch of
`--pairs` synthetic pairs whose inputs are already resident in HBM.  One process per GPU; for N > 1 the
driver launches this file under torch.distributed.run: the 10 kb chunk is broadcast from rank 0 over
RCCL/xGMI once, every rank aligns its own shard of reads (weak scaling, no data-path collective), the
timed region is bracketed by barrier + synchronize and the max over ranks is reported.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      algorithmic bytes (n + m + ceil(6nm/8) + ceil(6(n+m)/8) + 8 + 16|cigar| per pair)
                / average fill-kernel duration (HIP events on the launch stream, inside the library)
  cpu_baseline  the CPU oracle ("port" of the reference algorithm; the Go reference cannot be built here)
                on all host cores, bounded sample of the same workload
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

READ_LEN = 150
CHUNK_LEN = 10000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def make_workload(seed, n_pairs, chunk=None):
    """C2 generator, vectorised: reads sampled at uniform offsets of one chunk (0.1 % N), 1 % substitutions,
    ~0.2 %/base indel opens (one geometric(0.5)-length indel in ~26 % of the reads)."""
    rng = np.random.default_rng(seed)
    if chunk is None:
        chunk = rng.integers(0, 4, size=CHUNK_LEN).astype(np.uint8)
        chunk[rng.random(CHUNK_LEN) < 0.001] = 4
    off = rng.integers(0, CHUNK_LEN - READ_LEN - 64, size=n_pairs)
    x = np.arange(READ_LEN)[None, :]
    has_indel = rng.random(n_pairs) < 0.26
    pos = rng.integers(10, READ_LEN - 10, size=n_pairs)
    ln = np.minimum(rng.geometric(0.5, size=n_pairs), 32)
    is_del = rng.random(n_pairs) < 0.5
    shift = np.where(has_indel[:, None] & (x >= pos[:, None]), np.where(is_del, ln, -ln)[:, None], 0)
    src = off[:, None] + x + shift
    reads = chunk[np.clip(src, 0, CHUNK_LEN - 1)]
    ins_mask = has_indel[:, None] & (~is_del)[:, None] & (x >= pos[:, None]) & (x < (pos + ln)[:, None])
    reads = np.where(ins_mask, rng.integers(0, 4, size=reads.shape), reads)
    sub = rng.random(reads.shape) < 0.01
    reads = np.where(sub, rng.integers(0, 4, size=reads.shape), reads).astype(np.uint8)
    return np.ascontiguousarray(reads), chunk


def algorithmic_bytes(n, m, pairs, total_ops):
    per_pair = n + m + (n * m * 6 + 7) // 8 + ((n + m) * 6 + 7) // 8 + 8
    return per_pair * pairs + 16 * total_ops


def cpu_baseline(reads, chunk, scores, budget_s=15.0):
    import oracle
    cores = os.cpu_count() or 1
    n = READ_LEN

    last = [None]

    def run(k, threads):
        a_start = np.arange(k, dtype=np.int64) * n
        a_len = np.full(k, n, dtype=np.int64)
        b_start = np.zeros(k, dtype=np.int64)
        b_len = np.full(k, chunk.shape[0], dtype=np.int64)
        t0 = time.perf_counter()
        last[0] = oracle.align_batch_windows(oracle.MODE_AFFINE, scores, -600, -150, reads[:k].reshape(-1), a_start, a_len,
                                             chunk, b_start, b_len, threads=threads)
        return time.perf_counter() - t0

    k = min(4 * cores, reads.shape[0])
    dt = run(k, cores)  # also warms page tables
    while dt < budget_s / 2 and k < reads.shape[0]:  # grow the sample until it is ~budget_s of wall time
        k = min(reads.shape[0], max(k + 1, int(k * min(8.0, 0.9 * budget_s / max(dt, 1e-3)))))
        dt = run(k, cores)
    k_min = min(10000, reads.shape[0])  # the sample doubles as the bit-exactness check of the GPU results: at least 10 000 pairs
    if k < k_min and dt * k_min / k < 3 * budget_s:
        k = k_min
        dt = run(k, cores)
    oracle_results, oracle_k = last[0], k
    cells = k * n * chunk.shape[0]
    k1 = min(8, reads.shape[0])
    dt1 = run(k1, 1)  # context: one thread alone (containers often cap the CPU time of the nominal cores)
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"_oracle": (oracle_results, oracle_k), "cpu_model": model, "value": cells / dt, "unit": "DP cells/s", "cores": cores, "kind": "port",
            "pairs_per_s": k / dt, "single_thread_cells_per_s": k1 * n * chunk.shape[0] / dt1,
            "sample": "%d pairs (150x10000, C2 generator) through oracle/gnx_oracle.c or_align_batch, %d threads, %.1f s"
                      % (k, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=100000, help="pairs per GPU per step (config C2: 100 k)")
    ap.add_argument("--ws-gb", type=float, default=150.0, help="direction-matrix workspace limit per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--verify", type=int, default=32, help="pairs checked bit-exactly against the oracle after timing")
    ap.add_argument("--dist-backend", default="nccl", help="debug: 'gloo' lets several ranks share one GPU (with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="debug: every rank uses cuda:0 (flow check of the N>1 path on a 1-GPU box)")
    ap.add_argument("--series", default="affine", choices=["affine", "const", "local"],
                    help="affine = the headline AffineGap(read, chunk); const = ConstGap(read, chunk, -430); "
                         "local = AffineGapLocal(target=chunk, query=read) (SURVEY 8d second series)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gonomics_amd import _lib, align, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    dev_index = 0 if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend)
    L = _lib.lib()
    ws_gb = args.ws_gb / (world if args.share_gpu else 1)
    _lib.check(L.gnx_init(dev_index, int(ws_gb * (1 << 30))))

    n_pairs = args.pairs
    # rank 0 owns the chunk; everyone gets it by broadcast (RCCL over xGMI when world > 1)
    if rank == 0:
        reads0, chunk_h = make_workload(2, n_pairs)
    else:
        chunk_h = np.zeros(CHUNK_LEN, dtype=np.uint8)
    d_chunk = torch.from_numpy(chunk_h).to(dev)
    if world > 1:
        shard.broadcast_reference(d_chunk, src=0)  # RCCL broadcast over xGMI
        chunk_h = d_chunk.cpu().numpy()
    reads_h = reads0 if rank == 0 else make_workload(2 + 1000 * rank, n_pairs, chunk_h)[0]
    d_reads = torch.from_numpy(reads_h.reshape(-1)).to(dev)
    h_alen = np.full(n_pairs, READ_LEN, dtype=np.int64)
    h_blen = np.full(n_pairs, CHUNK_LEN, dtype=np.int64)
    d_as = torch.arange(n_pairs, dtype=torch.int64, device=dev) * READ_LEN
    d_al = torch.from_numpy(h_alen).to(dev)
    d_bs = torch.zeros(n_pairs, dtype=torch.int64, device=dev)
    d_bl = torch.from_numpy(h_blen).to(dev)
    d_score = torch.zeros(n_pairs, dtype=torch.int64, device=dev)
    d_off = torch.zeros(n_pairs + 1, dtype=torch.int64, device=dev)
    cap = 48 * n_pairs
    d_ops = torch.zeros(cap * 16, dtype=torch.uint8, device=dev)
    if args.series == "affine":
        gmode, omode, go, ge = _lib.GNX_AFFINE_GAP, 0, -600, -150
    elif args.series == "const":
        gmode, omode, go, ge = _lib.GNX_CONST_GAP, 1, -430, 0
    else:
        gmode, omode, go, ge = _lib.GNX_AFFINE_GAP_LOCAL, 3, -600, -150
    params = _lib.make_params(gmode, align.HumanChimpTwoScoreMatrix, go, ge, 10000, 10000)
    total_ops = ctypes.c_int64()
    stream = torch.cuda.current_stream().cuda_stream
    swap = args.series == "local"  # AffineGapLocal(target=chunk, query=read): alpha is the chunk

    def step():
        if swap:
            _lib.check(L.gnx_align_batch_device(ctypes.byref(params), n_pairs, d_chunk.data_ptr(), d_bs.data_ptr(), d_bl.data_ptr(),
                                                d_reads.data_ptr(), d_as.data_ptr(), d_al.data_ptr(),
                                                h_blen.ctypes.data, h_alen.ctypes.data,
                                                d_score.data_ptr(), d_ops.data_ptr(), cap, d_off.data_ptr(),
                                                ctypes.byref(total_ops), ctypes.c_void_p(stream)))
            return
        _lib.check(L.gnx_align_batch_device(ctypes.byref(params), n_pairs, d_reads.data_ptr(), d_as.data_ptr(), d_al.data_ptr(),
                                            d_chunk.data_ptr(), d_bs.data_ptr(), d_bl.data_ptr(),
                                            h_alen.ctypes.data, h_blen.ctypes.data,
                                            d_score.data_ptr(), d_ops.data_ptr(), cap, d_off.data_ptr(),
                                            ctypes.byref(total_ops), ctypes.c_void_p(stream)))

    try:  # one untimed sizing call: const-gap CIGARs have hundreds of runs per pair
        step()
    except _lib.GnxError as e:
        if e.code != _lib.GNX_ECAPACITY:
            raise
        cap = int(total_ops.value * 1.05) + 1024
        d_ops = torch.zeros(cap * 16, dtype=torch.uint8, device=dev)
    for _ in range(args.warmup):
        step()
    fill_ms, tb_ms, launches = [], [], 0
    dom_ms, dom_launches, fast_path = 0.0, 0, 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = _lib.get_timing()
        fill_ms.append(tm["fill_ms"]); tb_ms.append(tm["traceback_ms"]); launches += tm["n_launches"]
        dom_ms += tm["dominant_ms"]; dom_launches += tm["dominant_launches"]; fast_path = tm["fast_path"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- after the timed region: verification + baselines (rank 0) ----
    ok = True
    if args.verify > 0:
        import oracle
        k = min(args.verify, n_pairs)
        sc = d_score[:k].cpu().numpy()
        off = d_off[:k + 1].cpu().numpy()
        ops = d_ops[:int(off[-1]) * 16].cpu().numpy().view(_lib.CIGAR_DTYPE)
        a_start = np.arange(k, dtype=np.int64) * READ_LEN
        if swap:
            exp = oracle.align_batch_windows(omode, align.HumanChimpTwoScoreMatrix, go, ge, chunk_h, np.zeros(k, np.int64), h_blen[:k],
                                             reads_h.reshape(-1), a_start, h_alen[:k], threads=min(k, os.cpu_count() or 1))
        else:
            exp = oracle.align_batch_windows(omode, align.HumanChimpTwoScoreMatrix, go, ge, reads_h.reshape(-1),
                                             a_start, h_alen[:k], chunk_h, np.zeros(k, np.int64), h_blen[:k], threads=min(k, os.cpu_count() or 1))
        ok = bool(np.array_equal(sc, exp[0]) and np.array_equal(off, exp[2]) and np.array_equal(ops["run_length"], exp[1]["run_length"])
                  and np.array_equal(ops["op"], exp[1]["op"]))
    if world > 1:
        f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        ok = bool(f.item())
        # final gather of scores / CIGAR offsets / CIGAR blob on rank 0, in input order (outside the timed region)
        n_ops_local = int(d_off[n_pairs].item())
        gathered = shard.gather_results(d_score, d_ops[: n_ops_local * 16], d_off, dst=0)
        if rank == 0:
            ok = ok and gathered[0].numel() == n_pairs * world and int(gathered[2][-1].item()) * 16 == gathered[1].numel()
    if rank == 0:
        cells_per_step = n_pairs * READ_LEN * CHUNK_LEN * world
        value = cells_per_step * args.steps / dt
        fill_avg_ms = dom_ms / max(dom_launches, 1)  # average duration of the dominant kernel's launches (HIP events)
        pairs_per_launch = n_pairs * args.steps / max(dom_launches, 1)
        abytes = algorithmic_bytes(READ_LEN, CHUNK_LEN, pairs_per_launch, total_ops.value * pairs_per_launch / n_pairs)
        achieved = abytes / (fill_avg_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per fill launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
        try:
            with open(os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")) as fh:
                traffic = json.load(fh)["fast_path" if fast_path else "general_path"]["hbm_bytes_per_pair"] * pairs_per_launch
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "DP cells/sec + aligned pairs/sec, affine-gap 150bp x 10kb batch" if args.series == "affine"
                      else "DP cells/sec, series=%s (not the headline metric)" % args.series,
            "value": value, "unit": "DP cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "C2 faChunkAlign-style: %d x (150 bp read vs one 10 kb chunk) per GPU, %s, "
                                   "HumanChimpTwoScoreMatrix, %s, score + full CIGAR"
                                   % (n_pairs, {"affine": "align.AffineGap", "const": "align.ConstGap", "local": "align.AffineGapLocal(target=chunk, query=read)"}[args.series],
                                      "gapPen -430" if args.series == "const" else "gapOpen -600, gapExtend -150"),
                       "pairs_per_gpu": n_pairs, "read_len": READ_LEN, "chunk_len": CHUNK_LEN, "parallelism": "pairs sharded x%d" % world},
            "pairs_per_s": n_pairs * world * args.steps / dt,
            "bit_exact_sample": ok, "bit_exact_pairs_checked": int(min(args.verify, n_pairs)),
            "kernel_ms": {"all_fill_kernels_per_step": float(np.mean(fill_ms)), "traceback_and_rest_per_step": float(np.mean(tb_ms)),
                          "dominant_kernel_per_step": dom_ms / args.steps, "fast_path": bool(fast_path)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (PMC, profiles/r1_hbm_traffic.json)",
                         "kernel": "fp_sweep_kernel<%d, %s> (fast-path forward sweep)" % (19 if READ_LEN <= 152 else 20, "true" if args.series == "local" else "false") if fast_path
                                   else "fill_affine_kernel (full direction matrix)",
                         "avg_launch_ms": fill_avg_ms,
                         "algorithmic_bytes_per_launch": abytes,
                         "cells_per_s_kernel": pairs_per_launch * READ_LEN * CHUNK_LEN / (fill_avg_ms * 1e-3)},
        }
        try:  # second ceiling: VALU issue, 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6e12 lane-ops/s
            with open(os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")) as fh:
                fpj = json.load(fh)["fast_path"]
                vi = fpj.get("valu_insts_per_pair") if fast_path else None
            if vi:
                lane_ops = vi * 64.0 * pairs_per_launch / (fill_avg_ms * 1e-3)
                out["roofline_valu"] = {"bound": "valu", "achieved": lane_ops / 1e12, "peak": 78.6, "unit": "T lane-ops/s", "frac": lane_ops / 78.6e12,
                                        "valu_insts_per_pair": vi, "valu_busy": fpj.get("valu_busy"),
                                        "source": "SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE (rocprofv3 --pmc, profiles/r1_pmc_sq.csv)"}
        except (OSError, KeyError, ValueError):
            pass
        if not args.no_cpu and args.series == "affine" and world == 1:
            cb = cpu_baseline(reads_h, chunk_h, align.HumanChimpTwoScoreMatrix)
            exp, k = cb.pop("_oracle")
            # the oracle results of the baseline sample double as the bit-exactness check of the first k GPU results
            sc = d_score[:k].cpu().numpy()
            off = d_off[:k + 1].cpu().numpy()
            ops = d_ops[:int(off[-1]) * 16].cpu().numpy().view(_lib.CIGAR_DTYPE)
            ok_k = bool(np.array_equal(sc, exp[0]) and np.array_equal(off, exp[2]) and np.array_equal(ops["run_length"], exp[1]["run_length"])
                        and np.array_equal(ops["op"], exp[1]["op"]))
            ok = ok and ok_k
            out["bit_exact_sample"] = ok
            out["bit_exact_pairs_checked"] = int(k)
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = value / cb["value"]
        if args.series != "affine":
            out["roofline"]["note"] = "algorithmic-byte model is the affine one; use cells_per_s_kernel for this series"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit("bit-exactness check against the oracle FAILED")


if __name__ == "__main__":
    main()
