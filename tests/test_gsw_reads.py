"""Rows N2 (callers) and N4 of SURVEY 8f: the graph aligner's read path -- seed index, seed search, traversals, per-read driver --
(gonomics_amd/genomeGraph.py; device: csrc/seed_kernels.hip.h + the gsw DP kernels) against the independent literal restatement
tests/pyref_gsw.py.  The reference's tests of this path only log: PARITY UNPINNED; the contract is in the module docstring."""
import numpy as np
import pytest

import common
import pyref_gsw as ref
from gonomics_amd import genomeGraph as gg

MX = common.matrices()


def make_case(seed, kind):
    """(node sequences, edges, reads): 'linear' = chromosomes without edges; 'snp' = a backbone with SNP / indel bubbles"""
    rng = np.random.default_rng(seed)
    seqs, edges = [], []
    if kind == "linear":
        for _ in range(3):
            s = rng.integers(0, 4, size=int(rng.integers(300, 900))).astype(np.uint8)
            s[rng.random(s.shape[0]) < 0.004] = 4
            seqs.append(s)
        paths = [[k] for k in range(3)]
    else:
        prev = None
        path_a, path_b = [], []
        for b in range(6):
            k = len(seqs)
            seqs.append(rng.integers(0, 4, size=int(rng.integers(40, 160))).astype(np.uint8))
            if prev is not None:
                for u in prev:
                    edges.append((u, k))
            path_a.append(k); path_b.append(k)
            if b < 5:
                alt1 = rng.integers(0, 4, size=int(rng.integers(1, 4))).astype(np.uint8)
                alt2 = rng.integers(0, 4, size=int(rng.integers(1, 6))).astype(np.uint8)
                seqs.append(alt1); seqs.append(alt2)
                edges.append((k, k + 1)); edges.append((k, k + 2))
                path_a.append(k + 1); path_b.append(k + 2)
                prev = [k + 1, k + 2]
        paths = [path_a, path_b]
    reads = []
    for r in range(40):
        pth = paths[int(rng.integers(0, len(paths)))]
        hap = np.concatenate([seqs[k] for k in pth])
        L = int(rng.integers(36, 120))
        if hap.shape[0] <= L + 2:
            L = hap.shape[0] - 2
        o = int(rng.integers(0, hap.shape[0] - L))
        rd = common.mutate(rng, hap[o:o + L + 10], sub=0.03, indel=0.02, geo=0.5, alphabet=4)[:L]
        if r % 7 == 0:
            rd = rd.copy(); rd[int(rng.integers(0, len(rd)))] = 4
        if r % 2:
            rd = np.asarray([3 - int(x) if x < 4 else 4 for x in rd[::-1]], dtype=np.uint8)
        reads.append(rd)
    return seqs, edges, reads


def build(seqs, edges):
    g = gg.GenomeGraph()
    for k, s in enumerate(seqs):
        gg.AddNode(g, gg.Node(k, s))
    for u, v in edges:
        gg.AddEdge(g.Nodes[u], g.Nodes[v])
    return g


def seed_keys(seeds):
    return [s.key() for s in seeds]


@pytest.mark.parametrize("kind", ["linear", "snp"])
def test_host_statement_matches_the_restatement(kind):
    """CPU only: two-bit words with the N quirk, match counting, index, seeds and their order"""
    seqs, edges, reads = make_case(5, kind)
    g = build(seqs, edges)
    nodes = ref.make_graph(seqs, edges)
    for k, n in enumerate(g.Nodes):
        assert n.SeqTwoBit.Seq == nodes[k]["tb"][0]
    for seed_len, step in ((16, 1), (20, 7), (32, 32)):
        idx = gg.IndexGenomeIntoMap(g.Nodes, seed_len, step)
        assert idx == ref.index_genome(nodes, seed_len, step)
        for rd in reads[:12]:
            big = gg.FastqBig("r", rd)
            r2 = ref.make_read(rd)
            assert [t.Seq for t in big.rainbows()[0]] == [w for w, _ in r2["rb"]]
            got = gg.seed_map_host(idx, g.Nodes, big, seed_len)
            assert seed_keys(got) == ref.seed_map(idx, nodes, r2, seed_len)
    a = [gg.SeedDev(0, k, 0, 1, True, int(v)) for k, v in enumerate(np.random.default_rng(1).integers(1, 9, size=57))]
    b = [((0, k, 0, 1, True, s.TotalLength),) for k, s in enumerate(a)]
    gg.heapSortSeeds(a)
    ref.heap_sort(b)
    assert seed_keys(a) == b


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "snp"])
def test_device_index_and_seed_search(gpu_lib, kind):
    seqs, edges, reads = make_case(6, kind)
    g = build(seqs, edges)
    nodes = ref.make_graph(seqs, edges)
    for seed_len, step in ((16, 1), (24, 5), (32, 32)):
        full = ref.index_genome(nodes, seed_len, step)
        index = gg.SeedIndex(g.Nodes, seed_len, step)
        ks = sorted(full)
        assert [int(x) for x in index.keys] == [k for k in ks for _ in full[k]]
        assert [int(x) for x in index.locs] == [v for k in ks for v in full[k]]  # the map's insertion order within a key
        bigs = [gg.FastqBig("r%d" % k, rd) for k, rd in enumerate(reads)]
        got = gg.seed_map_batch(index, g.Nodes, bigs, seed_len)
        for k, rd in enumerate(reads):
            assert seed_keys(got[k]) == ref.seed_map(full, nodes, ref.make_read(rd), seed_len), "read %d" % k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "snp"])
def test_reads_to_giraf(gpu_lib, kind):
    """GraphSmithWatermanToGiraf for a batch: device seeds + rounds of batched device DPs == the sequential restatement on the CPU oracle"""
    seqs, edges, reads = make_case(7, kind)
    g = build(seqs, edges)
    nodes = ref.make_graph(seqs, edges)
    sc = MX["HumanChimpTwo"]
    seed_len = 16
    index = gg.SeedIndex(g.Nodes, seed_len, 1)
    full = ref.index_genome(nodes, seed_len, 1)
    bigs = [gg.FastqBig("r%d" % k, rd) for k, rd in enumerate(reads)]
    got = gg.GswBatchToGiraf(g, bigs, index, seed_len, sc)
    mapped = 0
    for k, rd in enumerate(reads):
        r2 = ref.make_read(rd)
        exp = ref.read_to_giraf(nodes, r2, ref.seed_map(full, nodes, r2, seed_len), sc)
        assert got[k].key() == ref.giraf_key(exp), "read %d" % k
        mapped += exp["AlnScore"] > 0
    assert mapped >= len(reads) // 2
