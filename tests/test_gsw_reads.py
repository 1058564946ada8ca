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
    """(node sequences, edges, reads): 'linear' = chromosomes without edges; 'snp' = a backbone with SNP / indel bubbles of 1 .. 5
    bases (the reference PANICS when a left extension that has collected bases enters such a node: search.go:139);
    'wide' = bubbles whose alleles are long enough (>= 2 x extension) for the reference's expression to stay in bounds"""
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
        n_alt = 3 if kind == "wide3" else 2  # "wide3": three alleles per bubble -- every traversal across one has >= 3 sibling branches, the case in
        #                                      which a later sibling's DP writes through the route slice an earlier best alignment still points into
        paths = [[] for _ in range(n_alt)]
        for b in range(6):
            k = len(seqs)
            seqs.append(rng.integers(0, 4, size=int(rng.integers(40, 160) if kind == "snp" else rng.integers(420, 520))).astype(np.uint8))
            if prev is not None:
                for u in prev:
                    edges.append((u, k))
            for pth in paths:
                pth.append(k)
            if b < 5:
                if kind == "snp":
                    alts = [rng.integers(0, 4, size=int(rng.integers(1, 4))).astype(np.uint8), rng.integers(0, 4, size=int(rng.integers(1, 6))).astype(np.uint8)]
                else:  # long alleles that differ by a few substitutions and an indel
                    alt1 = rng.integers(0, 4, size=int(rng.integers(420, 520))).astype(np.uint8)
                    alts = [alt1] + [common.mutate(rng, alt1, sub=0.02, indel=0.005, geo=0.5, alphabet=4) for _ in range(n_alt - 1)]
                prev = []
                for a, alt in enumerate(alts):
                    seqs.append(alt); edges.append((k, k + 1 + a)); paths[a].append(k + 1 + a); prev.append(k + 1 + a)
    reads = []
    for r in range(40):
        pth = paths[int(rng.integers(0, len(paths)))]
        hap = np.concatenate([seqs[k] for k in pth])
        L = int(rng.integers(36, 120))
        if hap.shape[0] <= L + 2:
            L = hap.shape[0] - 2
        o = int(rng.integers(0, hap.shape[0] - L))
        if kind in ("wide", "wide3") and r % 3 == 0:  # start a few bases after a node border: the left extension collects bases, then enters the Prev node
            cuts = np.cumsum([len(seqs[k]) for k in pth])[:-1]
            o = int(cuts[int(rng.integers(0, len(cuts)))]) + int(rng.integers(1, 12))
            o = min(o, hap.shape[0] - L - 1)
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


def test_left_target_known_answers():
    """getLeftTargetBases as Go parses it (search.go:139): n.Seq[refEnd - Min(len(seq)+refEnd, extension) - len(seq) : refEnd] ++ seq.
    Expected values worked out by hand from that expression, not from any of the mirrors."""
    n = gg.Node(0, np.arange(300, dtype=np.int64).astype(np.uint8) % 4)
    seq = np.asarray([4] * 20, dtype=np.uint8)
    # first node of a traversal (seq empty): 300 - min(300, 100) - 0 = 200 -> 100 bases
    t = gg._left_target(n, 100, 300, np.zeros(0, np.uint8))
    assert len(t) == 100 and list(t) == list(n.Seq[200:300])
    # 20 bases collected, long Prev node: 300 - min(320, 100) - 20 = 180 -> 120 = extension + len(seq) bases of the node, then seq
    t = gg._left_target(n, 100, 300, seq)
    assert len(t) == 140 and list(t[:120]) == list(n.Seq[180:300]) and list(t[120:]) == [4] * 20
    # ... and the restatement takes the same 120
    nodes = ref.make_graph([n.Seq], [])
    # short Prev node (refEnd = 50): 50 - min(70, 100) - 20 = -40 -> Go panics
    with pytest.raises(gg.GoPanic):
        gg._left_target(gg.Node(1, n.Seq[:50]), 100, 50, seq)
    # node exactly long enough: refEnd = 120: 120 - min(140, 100) - 20 = 0 -> the whole node
    t = gg._left_target(gg.Node(2, n.Seq[:120]), 100, 120, seq)
    assert len(t) == 140
    # one base short: refEnd = 119 -> -1 -> panic
    with pytest.raises(gg.GoPanic):
        gg._left_target(gg.Node(3, n.Seq[:119]), 100, 119, seq)
    assert nodes is not None


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "snp", "wide", "wide3"])
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
    got = gg.GswBatchToGiraf(g, bigs, index, seed_len, sc, on_panic="mark")
    mapped = panics = 0
    for k, rd in enumerate(reads):
        r2 = ref.make_read(rd)
        try:
            exp = ref.read_to_giraf(nodes, r2, ref.seed_map(full, nodes, r2, seed_len), sc)
        except IndexError:  # the Go code panics on this read (search.go:139): the mirror must say so, not invent an alignment
            assert isinstance(got[k], gg.GoPanic), "read %d" % k
            panics += 1
            continue
        assert got[k].key() == ref.giraf_key(exp), "read %d" % k
        mapped += exp["AlnScore"] > 0
    assert mapped + panics >= len(reads) // 2
    assert (panics > 0) == (kind == "snp")
    if panics:  # default behaviour: what the Go process does
        with pytest.raises(gg.GoPanic):
            gg.GswBatchToGiraf(g, bigs, index, seed_len, sc)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "wide", "wide3"])
def test_wrap_pair_giraf(gpu_lib, kind):
    """WrapPairGiraf for a batch of read pairs (the paired reads of config C3 go through it, toGiraf.go:117-140): both mates in one batched
    call, flags as the Go code sets them (Fwd: +8 +16 +16, Rev: nothing; uint8), against the sequential restatement"""
    seqs, edges, reads = make_case(11, kind)
    g = build(seqs, edges)
    nodes = ref.make_graph(seqs, edges)
    sc = MX["HumanChimpTwo"]
    seed_len = 16
    index = gg.SeedIndex(g.Nodes, seed_len, 1)
    full = ref.index_genome(nodes, seed_len, 1)
    pairs = [(gg.FastqBig("p%d/1" % k, reads[2 * k]), gg.FastqBig("p%d/2" % k, reads[2 * k + 1])) for k in range(len(reads) // 2)]
    got = gg.WrapPairGirafBatch(g, pairs, index, seed_len, sc)
    flags_seen = set()
    for k, (fw, rv) in enumerate(got):
        r1, r2 = ref.make_read(reads[2 * k]), ref.make_read(reads[2 * k + 1])
        ef, er, ff, rf = ref.wrap_pair(nodes, r1, r2, ref.seed_map(full, nodes, r1, seed_len), ref.seed_map(full, nodes, r2, seed_len), sc)
        assert fw.key() == ref.giraf_key(ef) and rv.key() == ref.giraf_key(er), "pair %d" % k
        assert (fw.Flag, rv.Flag) == (ff, rf), "pair %d" % k
        flags_seen.add((ff, rf))
    assert len(flags_seen) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear", "snp", "wide", "wide3"])
def test_native_read_path_equals_python_mirror(gpu_lib, kind):
    """gnx_gsw_graph_create / gnx_gsw_map_reads (the whole read path of a batch in one C-ABI call: the compiled per-read driver on a pool
    of host threads around the device's seed search and DP rounds) == the Python mirror, read for read and pair for pair; and a batch
    against the graph again after another index took the device's place."""
    seqs, edges, reads = make_case(13, kind)
    g = build(seqs, edges)
    sc = MX["HumanChimpTwo"]
    index = gg.SeedIndex(g.Nodes, 16, 1)
    bigs = [gg.FastqBig("r%d" % k, rd) for k, rd in enumerate(reads)]
    py = gg.GswBatchToGiraf(g, bigs, index, 16, sc, on_panic="mark")
    ng = gg.NativeGraph(g, 16, 1)
    assert gg._edge_list(g) == [tuple(e) for e in edges]
    for threads in (1, 5):
        nat = ng.GswBatchToGiraf(bigs, sc, threads=threads, on_panic="mark")
        assert len(nat) == len(py)
        for k, (a, b) in enumerate(zip(nat, py)):
            if isinstance(b, gg.GoPanic):
                assert isinstance(a, gg.GoPanic), "read %d" % k
            else:
                assert a.key() == b.key() and a.Flag == b.Flag and a.MapQ == b.MapQ, "read %d (%d threads)" % (k, threads)
    assert any(isinstance(b, gg.GoPanic) for b in py) == (kind == "snp")
    # the Python mirror's seed search made ITS index resident in between (gnx_seed_index_set): the handle notices and uploads its own again
    pairs = [(bigs[2 * k], bigs[2 * k + 1]) for k in range(len(bigs) // 2)]
    pyp = gg.WrapPairGirafBatch(g, pairs, index, 16, sc, on_panic="mark")
    natp = ng.WrapPairGirafBatch(pairs, sc, on_panic="mark")
    for k, (pa, pb) in enumerate(zip(natp, pyp)):
        for a, b in zip(pa, pb):
            if isinstance(b, gg.GoPanic):
                assert isinstance(a, gg.GoPanic), "pair %d" % k
            else:
                assert a.key() == b.key() and a.Flag == b.Flag, "pair %d" % k
    gir, nodes, cig = ng.map_reads_raw([b.Seq for b in bigs], sc)
    assert gir.shape[0] == len(bigs) and int(gir["n_nodes"].sum()) == nodes.shape[0] and int(gir["n_cigar"].sum()) == cig.shape[0]
    ng.handle.close()


def test_native_graph_bad_arguments():
    """argument checks of the graph entry points that need no device (a bad edge, a base >= 5) -- and no CPU fallback behind them"""
    from gonomics_amd import _lib
    with pytest.raises(_lib.GnxError):
        _lib.GswGraph([np.zeros(40, np.uint8)], [(0, 3)], 16, 1)
    with pytest.raises(_lib.GnxError):
        _lib.GswGraph([np.full(40, 7, np.uint8)], [], 16, 1)


def test_edge_list_of_a_graph():
    """NativeGraph hands the edges over in an order of AddEdge calls that rebuilds every node's Next AND Prev list (traversals try a node's
    edges in list order); a graph whose lists fit neither order is refused, not reordered."""
    seqs, edges, _ = make_case(3, "wide3")
    g = build(seqs, edges)
    assert gg._edge_list(g) == [tuple(e) for e in edges]
    # edges added in an order that is neither "by source" nor "by target": still one sequence of AddEdge calls
    g2 = gg.GenomeGraph()
    for k in range(4):
        gg.AddNode(g2, gg.Node(k, np.zeros(40, np.uint8)))
    for u, v in ((1, 3), (0, 3), (1, 2), (0, 2)):
        gg.AddEdge(g2.Nodes[u], g2.Nodes[v])
    got = gg._edge_list(g2)
    nxt, prv = {k: [] for k in range(4)}, {k: [] for k in range(4)}
    for u, v in got:
        nxt[u].append(v); prv[v].append(u)
    assert nxt[0] == [3, 2] and nxt[1] == [3, 2] and prv[3] == [1, 0] and prv[2] == [1, 0]
    # lists that no sequence of AddEdge calls produces: 0 -> 3 before 0 -> 2 (Next of 0), 0 -> 2 ... 1 -> 2 ... and a cycle through the Prev lists
    g3 = gg.GenomeGraph()
    for k in range(4):
        gg.AddNode(g3, gg.Node(k, np.zeros(40, np.uint8)))
    for u, v in ((0, 2), (1, 2), (1, 3), (0, 3)):
        gg.AddEdge(g3.Nodes[u], g3.Nodes[v])
    assert gg._edge_list(g3) == [(0, 2), (1, 2), (1, 3), (0, 3)]
    g3.Nodes[0].Next.reverse()   # now 0 -> 3 before 0 -> 2, but 0 -> 2 before 1 -> 2 before 1 -> 3 before 0 -> 3: a cycle
    with pytest.raises(ValueError):
        gg._edge_list(g3)


def test_edge_list_of_the_reference_constructors_graphs():
    """the reference's own in-memory constructors do not add edges node by node -- VariantGraph adds (altAllele -> match) before
    (refAllele -> match) (genomeGraph/graphTools.go:100-108: the match node after a SNP has Prev = [k+2, k+1]), cmd/cigarToBed-style builders add
    (i-1 -> i) before (i-2 -> i).  The edge order handed to the library (here _edge_list; shim/genomeGraph/routines_hip.go hipEdgeOrder is its
    transliteration) must rebuild BOTH lists of every node exactly."""
    def rebuilt(g):
        order = gg._edge_list(g)
        nxt, prv = {k: [] for k in range(len(g.Nodes))}, {k: [] for k in range(len(g.Nodes))}
        for u, v in order:
            nxt[u].append(v); prv[v].append(u)
        ids = {id(n): k for k, n in enumerate(g.Nodes)}
        for k, n in enumerate(g.Nodes):
            assert nxt[k] == [ids[id(e.Dest)] for e in n.Next], (k, nxt[k])
            assert prv[k] == [ids[id(e.Dest)] for e in n.Prev], (k, prv[k])
        return order
    # two SNPs in a row, as VariantGraph builds them: match0, ref1, alt2, match3, ref4, alt5, match6
    g = gg.GenomeGraph()
    for k in range(7):
        gg.AddNode(g, gg.Node(k, np.zeros(30, np.uint8)))
    for u, v in ((0, 1), (0, 2), (2, 3), (1, 3), (3, 4), (3, 5), (5, 6), (4, 6)):
        gg.AddEdge(g.Nodes[u], g.Nodes[v])
    assert [gg_id for gg_id in ([g.Nodes.index(e.Dest) for e in g.Nodes[3].Prev])] == [2, 1]
    order = rebuilt(g)
    assert order.index((2, 3)) < order.index((1, 3)) and order.index((5, 6)) < order.index((4, 6))
    # (i-1 -> i) before (i-2 -> i)
    g2 = gg.GenomeGraph()
    for k in range(6):
        gg.AddNode(g2, gg.Node(k, np.zeros(30, np.uint8)))
    for i in range(1, 6):
        gg.AddEdge(g2.Nodes[i - 1], g2.Nodes[i])
        if i >= 2:
            gg.AddEdge(g2.Nodes[i - 2], g2.Nodes[i])
    rebuilt(g2)
