#!/usr/bin/env python3
"""The workload cmd/faChunkAlign really runs (cmd/faChunkAlign/faChunkAlign.go:18-29 -> align.AllSeqAffineChunk, align/multiAlign.go:70-78):
a progressive multiple alignment of G tandem-repeat-like sequences by chunks -- every round aligns ALL pairs of groups
(multipleAffineGapChunk, affineGap_highMem.go:308-353) and merges the best one.  Usage: python tools/bench_n1_cmd.py [G] [bases] [chunk]
One JSON line: whole-command wall time, per-round times, kernel times, the CPU oracle on a sample of first-round pairs beside it."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_records(G, bases, chunk, seed=8):
    """G versions of one tandem repeat (unit = chunk bases): 3 % substitutions, whole units lost / gained every ~150 units"""
    from gonomics_amd.fasta import Fasta
    rng = np.random.default_rng(seed)
    unit = rng.integers(0, 4, size=chunk).astype(np.uint8)
    units = bases // chunk
    recs = []
    for g in range(G):
        u = np.tile(unit, (units + 40, 1))
        keep = rng.random(units + 40) > 1 / 150.0
        u = u[keep][:units]
        while u.shape[0] < units:
            u = np.concatenate([u, unit[None, :]])
        s = u.reshape(-1).copy()
        mut = rng.random(s.size) < 0.03
        s[mut] = rng.integers(0, 4, size=int(mut.sum()))
        recs.append(Fasta("seq%d" % g, s))
    return recs


def run(G=8, bases=30000, chunk=3, go=-300, ge=-40, cpu_pairs=2, cpu_threads=0):
    from gonomics_amd import _lib, align, cmds, fasta
    import oracle
    L = _lib.lib()
    recs = make_records(G, bases, chunk)
    tmp = tempfile.mkdtemp()
    fin, fout = os.path.join(tmp, "in.fa"), os.path.join(tmp, "out.fa")
    fasta.Write(fin, recs)
    rounds = []
    inner = align.multipleAffineGapBatch

    def timed_batch(groups, pairs, sm, go_, ge_, cs):
        t0 = time.perf_counter()
        res = inner(groups, pairs, sm, go_, ge_, cs)
        tm = _lib.get_timing()
        cells = sum((len(groups[x][0].Seq) // cs) * (len(groups[y][0].Seq) // cs) for x, y in pairs)
        rounds.append({"pairs": len(pairs), "chunk_cells": cells, "call_s": time.perf_counter() - t0, "fill_ms": tm["fill_ms"], "traceback_ms": tm["traceback_ms"], "path": tm["fast_path"]})
        return res

    align.multipleAffineGapBatch = timed_batch
    try:
        t0 = time.perf_counter()
        out = cmds.faChunkAlign(fin, chunk, go, ge, fout)
        wall = time.perf_counter() - t0
    finally:
        align.multipleAffineGapBatch = inner
    cells = sum(r["chunk_cells"] for r in rounds)
    fill_s = sum(r["fill_ms"] for r in rounds) * 1e-3
    # CPU beside it: the oracle's multipleAffineGapChunk on first-round pairs (singleton groups), one pair per thread
    blocks = [np.asarray(r.Seq, dtype=np.uint8)[None, :] for r in recs]
    if cpu_pairs <= 0:
        return {"command_s": wall, "per_round": rounds}
    prs = [(x, y) for x in range(G - 1) for y in range(x + 1, G)][:cpu_pairs]
    import concurrent.futures as cf
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=len(prs)) as ex:
        exp = list(ex.map(lambda xy: oracle.multiple_affine_gap(align.HumanChimpTwoScoreMatrix, go, ge, chunk, blocks[xy[0]], blocks[xy[1]]), prs))
    cpu_s = time.perf_counter() - t0
    cpu_cells = len(prs) * (bases // chunk) ** 2
    sc, ops, off = _lib.multiple_affine_gap_batch(_lib.make_params(_lib.GNX_AFFINE_GAP_HIGHMEM, align.HumanChimpTwoScoreMatrix, go, ge), chunk, blocks, prs)
    ok = all((int(sc[k]), [(int(r), int(o)) for r, o in zip(ops["run_length"][off[k]:off[k + 1]], ops["op"][off[k]:off[k + 1]])]) == exp[k] for k in range(len(prs)))
    n = m = bases // chunk
    # SURVEY 8d's byte model per pair of the FIRST round, in chunk cells: bases of both members + 6 bits per cell + the path + score + runs
    abytes = sum(r["chunk_cells"] for r in rounds) * 6 / 8.0
    return {"workload": "cmd/faChunkAlign: AllSeqAffineChunk of %d sequences x %d bases, chunk %d, HumanChimpTwo, gapOpen %d gapExtend %d (multi-fasta in, multi-fasta out)" % (G, bases, chunk, go, ge),
            "command_s": wall, "rounds": len(rounds), "alignments": sum(r["pairs"] for r in rounds), "chunk_cells": cells,
            "value": cells / wall, "unit": "chunk cells/s (whole command: file in, file out)",
            "dp_calls_s": sum(r["call_s"] for r in rounds), "fill_kernels_s": fill_s, "traceback_kernels_s": sum(r["traceback_ms"] for r in rounds) * 1e-3,
            "per_round": rounds, "aligned_columns": int(len(out[0].Seq)),
            "roofline": {"bound": "hbm", "kernel": {0: "fill_affine_kernel<.., SCORED>", 3: "lat_fill_kernel<.., SCORED>"}.get(rounds[0]["path"], str(rounds[0]["path"])),
                         "achieved": abytes / fill_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": abytes / fill_s / 8e12, "direction_bits_per_chunk_cell": 6,
                         "algorithmic_bytes": abytes, "cells_per_s_kernel": cells / fill_s, "traffic": None,
                         "note": "the fill also READS a materialised score matrix (4 or 2 B per chunk cell, written by score_matrix_kernel): not part of the algorithmic bytes"},
            "cpu_baseline": {"value": cpu_cells / cpu_s, "unit": "chunk cells/s", "cores": len(prs), "kind": "port",
                             "sample": "%d first-round pairs (%d x %d chunk cells each) through the oracle's multipleAffineGapChunk, one pair per thread, %.1f s" % (len(prs), n, m, cpu_s),
                             "whole_command_extrapolated_s_per_core": cells / (cpu_cells / cpu_s / len(prs))},
            "bit_exact_sample": bool(ok), "bit_exact_pairs_checked": len(prs)}


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*(a[:3]))))
