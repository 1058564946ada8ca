#!/usr/bin/env python3
"""Measurement for rows N4 / N2-callers (not the headline metric): seed index of a synthetic linear genome on the device, seed search for a
batch of 150-base reads, and the whole read path (seeds -> traversals -> DPs -> giraf) for a smaller batch.  One JSON line per series."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from gonomics_amd import _lib, align, genomeGraph as gg  # noqa: E402


def main():
    L = _lib.lib()
    _lib.check(L.gnx_init(0, 32 << 30))
    rng = np.random.default_rng(9)
    n_nodes, node_len, seed_len, step = 24, 2000000, 32, 32
    seqs = [rng.integers(0, 4, size=node_len).astype(np.uint8) for _ in range(n_nodes)]
    t0 = time.perf_counter()
    keys, locs = _lib.seed_index_build(seqs, seed_len, step)
    t_build = time.perf_counter() - t0
    print(json.dumps({"series": "IndexGenomeIntoMap on the device: %d nodes x %d bases, seedLen %d, step %d" % (n_nodes, node_len, seed_len, step),
                      "kmers": int(keys.shape[0]), "host_call_s": t_build, "bases_per_s": n_nodes * node_len / t_build}), flush=True)
    n_reads = 100000
    reads = []
    for _ in range(n_reads):
        k = int(rng.integers(0, n_nodes)); o = int(rng.integers(0, node_len - 160))
        r = seqs[k][o:o + 150].copy()
        r[rng.random(150) < 0.01] = rng.integers(0, 4)
        reads.append(r if rng.random() < 0.5 else (3 - r[::-1]).astype(np.uint8))
    _lib.seed_find_batch(keys, locs, seqs, reads[:1000], seed_len)  # index + nodes resident, warm
    t0 = time.perf_counter()
    hits = _lib.seed_find_batch(keys, locs, seqs, reads, seed_len)
    t_find = time.perf_counter() - t0
    print(json.dumps({"series": "seed search (hash lookup + exact-match extension) for %d reads of 150 bases, both strands" % n_reads, "hits": int(sum(len(h) for h in hits)),
                      "host_call_s": t_find, "reads_per_s": n_reads / t_find, "note": "host_call_s includes the Python binding's list building"}), flush=True)
    # whole read path on a small graph object (Python bookkeeping dominates: this is the mirror of the reference's per-read logic, not a tuned mapper)
    g = gg.GenomeGraph()
    for k in range(4):
        gg.AddNode(g, gg.Node(k, seqs[k][:200000]))
    index = gg.SeedIndex(g.Nodes, seed_len, step)
    sub = []
    for _ in range(2000):
        k = int(rng.integers(0, 4)); o = int(rng.integers(0, 200000 - 170))
        sub.append(gg.FastqBig("r", common.mutate(rng, g.Nodes[k].Seq[o:o + 170], 0.02, 0.01)[:150]))
    t0 = time.perf_counter()
    out = gg.GswBatchToGiraf(g, sub, index, seed_len, align.HumanChimpTwoScoreMatrix)
    dt = time.perf_counter() - t0
    print(json.dumps({"series": "GswBatchToGiraf: %d reads against a 4 x 200 kb graph (device seeds + rounds of batched device DPs)" % len(sub), "mapped": int(sum(o.AlnScore > 0 for o in out)),
                      "host_call_s": dt, "reads_per_s": len(sub) / dt}), flush=True)
    # the same read path through the C++ mirror (include/gonomics_genomegraph.hpp) on the same graph, 10x the reads: what a compiled host pays
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gsw_cpp as tc
    tc._build()
    more = list(sub)
    for _ in range(18000):
        k = int(rng.integers(0, 4)); o = int(rng.integers(0, 200000 - 170))
        more.append(gg.FastqBig("r", common.mutate(rng, g.Nodes[k].Seq[o:o + 170], 0.02, 0.01)[:150]))
    most = list(more)
    for _ in range(80000):
        k = int(rng.integers(0, 4)); o = int(rng.integers(0, 200000 - 170))
        most.append(gg.FastqBig("r", common.mutate(rng, g.Nodes[k].Seq[o:o + 170], 0.02, 0.01)[:150]))
    # the same read path as ONE C-ABI call per batch from Python (gnx_gsw_graph_create + gnx_gsw_map_reads: the compiled driver inside the library)
    t0 = time.perf_counter()
    ng = gg.NativeGraph(g, seed_len, step)
    t_graph = time.perf_counter() - t0
    for name, batch in (("2000", sub), ("20000", more), ("100000", most)):
        seqs = [r.Seq for r in batch]
        seqs = (np.concatenate(seqs), np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.int64))  # as the C ABI takes them
        best, gir = None, None
        for _ in range(4):
            t0 = time.perf_counter()
            gir, _nodes, _cig = ng.map_reads_raw(seqs, align.HumanChimpTwoScoreMatrix)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row = {"series": "gnx_gsw_map_reads from Python (one C-ABI call per batch, arrays out): %s reads, same graph (fastest of 4 calls)" % name,
               "mapped": int((gir["aln_score"] > 0).sum()), "graph_and_index_s": t_graph, "host_call_s": best, "reads_per_s": len(batch) / best,
               "note": "reads handed over concatenated, as the C ABI takes them"}
        if name == "2000":
            nat = ng.GswBatchToGiraf(sub, align.HumanChimpTwoScoreMatrix)
            row["equals_python_mirror"] = all(a.key() == b.key() for a, b in zip(nat, out))
        print(json.dumps(row), flush=True)
    ng.handle.close()
    with tempfile.TemporaryDirectory() as td:
        for name, batch in (("2000", sub), ("20000", more), ("100000", most)):
            tc.write_case(os.path.join(td, "case.txt"), [n.Seq for n in g.Nodes], [], [r.Seq for r in batch], seed_len, step, align.HumanChimpTwoScoreMatrix)
            rows1 = None
            for threads in ("1", ""): # one host thread (round 3's loop), then the worker pool (GNX_GSW_THREADS unset: min(hardware threads, 16))
                env = dict(os.environ, GNX_GSW_REPEAT="4") # the batch five times in one process: the fastest call (index resident, workers started)
                env.pop("GNX_GSW_THREADS", None)
                if threads:
                    env["GNX_GSW_THREADS"] = threads
                subprocess.check_call([tc.BIN, os.path.join(td, "case.txt"), os.path.join(td, "out.txt")], env=env)
                rows, timing = tc.read_out(os.path.join(td, "out.txt"))
                if rows1 is None:
                    rows1 = rows
                same = None
                if name == "2000":
                    same = all(rows[k] == out[k].key()[:8] + (len(out[k].key()[8]),) for k in range(len(sub)))
                keys = ("seed_device", "seed_host", "tasks", "dp_pack", "dp_device", "dp_merge", "advance", "finish")
                print(json.dumps({"series": "GswBatchToGiraf through the C++ mirror: %s reads, same graph, %d host thread%s (fastest of 5 calls in one process)" % (name, int(timing[3]), "" if int(timing[3]) == 1 else "s"),
                                  "mapped": int(sum(r[7] > 0 for r in rows)), "host_threads": int(timing[3]),
                                  "index_ms": timing[0], "seeds_traversals_dps_ms": timing[1], "dp_rounds": int(timing[2]), "reads_per_s": len(batch) / (timing[1] / 1e3),
                                  "stages_ms": {k: round(v, 3) for k, v in zip(keys, timing[4:12])},
                                  "equals_python_mirror": same, "equals_one_thread": rows == rows1}), flush=True)
            # CPU baseline beside it: the SAME read path (device seeds, the C++ mirror's loop) with the extension DPs on the
            # CPU oracle (or_gsw_extend, literal restatement of search.go:234-321) on every host thread -- the reference's -t worker pool
            subprocess.check_call([tc.BIN, os.path.join(td, "case.txt"), os.path.join(td, "out_cpu.txt"), "reads", "cpu"], env=dict(os.environ, GNX_GSW_REPEAT="2"))
            rows_c, timing_c = tc.read_out(os.path.join(td, "out_cpu.txt"))
            print(json.dumps({"series": "cpu_baseline: the same read path, extension DPs on the CPU oracle (all host threads): %s reads" % name,
                              "host_threads": os.cpu_count(), "seeds_traversals_dps_ms": timing_c[1], "reads_per_s": len(batch) / (timing_c[1] / 1e3),
                              "equals_gpu_results": rows_c == rows, "gpu_vs_cpu": timing_c[1] / timing[1],
                              "kind": "port (oracle/gnx_oracle.c); seeds and index stay on the device in both runs"}), flush=True)
    census(rng)


def census(rng):
    """how often do the declared deviations of the parity contract come into play?  A variation graph in
    the style of config C3's gsw workload: a 600 kb backbone cut at a variant every ~1 kb (70 % SNPs: two alleles of 1 base; 30 %
    indels: alleles of 1 .. 6 bases), 3000 reads of 150 bases sampled from random haplotypes with 1 % substitutions."""
    for k in gg.STATS:
        gg.STATS[k] = 0
    g = gg.GenomeGraph()
    prev, paths = None, []
    hap_nodes = []
    nid = 0
    pos = 0
    while pos < 600000:
        ln = int(rng.integers(600, 1400))
        node = gg.Node(nid, rng.integers(0, 4, size=ln).astype(np.uint8)); gg.AddNode(g, node); nid += 1
        if prev is not None:
            for u in prev:
                gg.AddEdge(u, node)
        hap_nodes.append([node])
        pos += ln
        snp = rng.random() < 0.7
        a1 = gg.Node(nid, rng.integers(0, 4, size=1 if snp else int(rng.integers(1, 7))).astype(np.uint8)); gg.AddNode(g, a1); nid += 1
        a2 = gg.Node(nid, rng.integers(0, 4, size=1 if snp else int(rng.integers(1, 7))).astype(np.uint8)); gg.AddNode(g, a2); nid += 1
        gg.AddEdge(node, a1); gg.AddEdge(node, a2)
        hap_nodes.append([a1, a2])
        prev = [a1, a2]
    seed_len, step = 32, 32
    index = gg.SeedIndex(g.Nodes, seed_len, step)
    reads = []
    for _ in range(3000):
        k0 = int(rng.integers(0, len(hap_nodes) - 8))
        hap = np.concatenate([c[int(rng.integers(0, len(c)))].Seq for c in hap_nodes[k0:k0 + 8]])
        o = int(rng.integers(0, hap.shape[0] - 160))
        r = hap[o:o + 150].copy()
        r[rng.random(150) < 0.01] = rng.integers(0, 4)
        reads.append(gg.FastqBig("r", r if rng.random() < 0.5 else (3 - r[::-1]).astype(np.uint8)))
    t0 = time.perf_counter()
    out = gg.GswBatchToGiraf(g, reads, index, seed_len, align.HumanChimpTwoScoreMatrix, on_panic="mark")
    dt = time.perf_counter() - t0
    panics = sum(isinstance(o, gg.GoPanic) for o in out)
    print(json.dumps({"series": "census of the parity contract's declared deviations: 3000 reads of 150 bases on a 600 kb variation graph (a variant every ~1 kb)",
                      "reads": len(reads), "nodes": len(g.Nodes), "reads_the_go_code_panics_on": int(panics),
                      "mapped": int(sum((not isinstance(o, gg.GoPanic)) and o.AlnScore > 0 for o in out)), **gg.STATS, "host_call_s": dt,
                      "reading": "reads_with_more_than_100_seeds: order among equal TotalLength undefined without a Go toolchain (sort.Slice); branching traversals: "
                                 "where Go's shared backing arrays alias an earlier sibling's route -- MODELLED since round 4 (Go-slice classes in the mirrors and in the restatement): no longer a deviation; "
                                 "panics: getLeftTargetBases with a short Prev node (search.go:139)",
                      "declared_deviations_in_play": int(gg.STATS.get("reads_with_more_than_100_seeds", 0))}), flush=True)


if __name__ == "__main__":
    main()
