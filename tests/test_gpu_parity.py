"""Parity tests proper: the HIP path (through the C ABI) against the reference's golden vectors and
against the CPU oracle on seeded inputs.  Bit-exact: scores and CIGARs (integer work, no tolerance)."""
import os

import numpy as np
import pytest

import common
import oracle
from gonomics_amd import align, dna, fasta

pytestmark = pytest.mark.gpu

T = common.tables()
MX = common.matrices()


def _route(r):
    return [(c.RunLength, c.Op) for c in r]


# ---- the reference's own tests, re-stated against the drop-in API -------------------------------------
def test_affine_gap_highmem_view(gpu_lib):  # align/affineGap_test.go:45-55
    t = T["affineAlignTests"]
    for c in t["cases"]:
        a, b = dna.StringToBases(c["seqOne"]), dna.StringToBases(c["seqTwo"])
        _, cigar = align.AffineGap_highMem(a, b, align.DefaultScoreMatrix, -400, -30)
        assert align.View(a, b, cigar) == c["aln"]


def test_affine_gap_lowmem(gpu_lib):  # align/affineGap_test.go:57-81
    t = T["affineAlignTests"]
    for c in t["cases"]:
        a, b = dna.StringToBases(c["seqOne"]), dna.StringToBases(c["seqTwo"])
        hs, hr = align.AffineGap_highMem(a, b, align.DefaultScoreMatrix, -400, -30)
        ls, lr = align.AffineGap(a, b, align.DefaultScoreMatrix, -400, -30)
        cs, cr = align.AffineGap_customizeCheckersize(a, b, align.DefaultScoreMatrix, -400, -30, 3, 3)
        assert ls == hs and cs == hs
        assert lr == hr
        assert all(cr[k] == hr[k] for k in range(len(cr)))
        # and exactly what the reference algorithm gives at checkersize 3
        assert (cs, _route(cr)) == oracle.align_one(oracle.MODE_AFFINE, MX["Default"], -400, -30, a, b, 3, 3)


def test_affine_gap_local(gpu_lib):  # align/affineGap_test.go:120-155
    for c in T["affineLocalTests"]["cases"]:
        tgt, qry = dna.StringToBases(c["target"]), dna.StringToBases(c["query"])
        score, cig = align.AffineGapLocal(tgt, qry, align.DefaultScoreMatrix, c["gapOpen"], c["gapExtend"])
        assert score == c["score"] and align.PrintCigar(cig) == c["cigar"]


def test_go_affine_gap_local_engine(gpu_lib):  # align/affineGap_test.go:157-192
    cases = T["affineLocalEngineTests"]["cases"]
    inputs, outputs = align.GoAffineGapLocalEngine(align.DefaultScoreMatrix, -600, -150)
    for c in cases:
        test = align.TargetQueryPair(Target=dna.StringToBases(c["target"]), Query=dna.StringToBases(c["query"]))
        inputs.send(test)
        test = outputs.recv()
        assert test.Score == c["score"] and align.PrintCigar(test.Cigar) == c["cigar"]
    # FIFO order when several are in flight
    for c in cases:
        inputs.send(align.TargetQueryPair(Target=dna.StringToBases(c["target"]), Query=dna.StringToBases(c["query"])))
    inputs.close()
    got = [(r.Score, align.PrintCigar(r.Cigar)) for r in outputs]
    assert got == [(c["score"], c["cigar"]) for c in cases]


def test_const_gap_view(gpu_lib):  # align/view_test.go:28-38
    t = T["constAlignTests"]
    for c in t["cases"]:
        a, b = dna.StringToBases(c["seqOne"]), dna.StringToBases(c["seqTwo"])
        _, cigar = align.ConstGap(a, b, align.DefaultScoreMatrix, -430)
        assert align.View(a, b, cigar) == c["aln"]
        _, hc = align.ConstGap_highMem(a, b, align.DefaultScoreMatrix, -430)
        assert hc == cigar


def test_global_alignment_cmd(gpu_lib):  # cmd/globalAlignment/globalAlignment_test.go:13-40 + faOut fixture
    t = T["globalAlignmentGraph"]
    a, b = dna.StringToBases(t["toad"]), dna.StringToBases(t["ahsoka"])
    _, aln = align.ConstGap(a, b, align.HumanChimpTwoScoreMatrix, -430)
    assert len(aln) == 3 and align.PrintCigar(aln) == "3M3D3M"
    d = os.path.join(common.DATA, "globalAlignment")
    fa1, fa2 = fasta.Read(os.path.join(d, "chelsea.fa"))[0], fasta.Read(os.path.join(d, "eric.fa"))[0]
    _, aln = align.ConstGap(fa1.Seq, fa2.Seq, align.HumanChimpTwoScoreMatrix, -430)
    v = align.View(fa1.Seq, fa2.Seq, aln).split("\n")
    assert ">" + fa1.Name + "\n" + v[0] + "\n>" + fa2.Name + "\n" + v[1] + "\n" == open(os.path.join(d, "faOut_test.fa")).read()


@pytest.mark.parametrize("idx", [1, 2])
def test_global_alignment_anchor_tsv(gpu_lib, idx):  # cmd/globalAlignmentAnchor test: score + %v cigar
    cases = common.anchor_cases(idx)
    params = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150, 10000, 10000)
    res = align.AlignBatch(params, [c[0] for c in cases], [c[1] for c in cases])  # the cmd's loop, batched
    for (a, b, score, cig), (s, route) in zip(cases, res):
        assert s == score and align.FormatCigar(route) == cig
        s1, r1 = align.AffineGap_customizeCheckersize(a, b, align.HumanChimpTwoScoreMatrix, -600, -150, 10000, 10000)
        assert (s1, r1) == (s, route)


def test_cigar_to_bed(gpu_lib):  # cmd/cigarToBed/cigarToBed_test.go: 9x15 and 9673x10000
    d = os.path.join(common.DATA, "cigarToBed")
    for sub, f1, f2, fi, fd, ins, dele, exp in [
        ("sethvsraven", "seth.fa", "raven.fa", 1, 1, "affineGap_sethvsraven_ins.bed", "affineGap_sethvsraven_del.bed", (-1070, 3)),
        ("firstTest", "testRegion10kb_PanTro6.fa", "testRegion10kb_hg38.fa", 119320000, 116703287,
         "affineGap_PanTro6vshg38_ins.bed", "affineGap_PanTro6vshg38_del.bed", (790738, 19)),
    ]:
        a = dna.AllToUpper(fasta.Read(os.path.join(d, sub, f1))[0].Seq)
        b = dna.AllToUpper(fasta.Read(os.path.join(d, sub, f2))[0].Seq)
        s, aln = align.AffineGap(a, b, align.HumanChimpTwoScoreMatrix, -600, -150)
        gi, gd = common.cigar_to_beds(_route(aln), fi, fd, "chr1")
        assert gi == open(os.path.join(d, sub, ins)).read()
        assert gd == open(os.path.join(d, sub, dele)).read()
        assert (s, len(aln)) == exp
        assert (s, _route(aln)) == oracle.align_one(oracle.MODE_AFFINE, MX["HumanChimpTwo"], -600, -150, a, b)


def test_appendix_b_quirks(gpu_lib):
    for kind, mx, go, ge, cs, sa, sb, score, high, low in common.QUIRK_CASES:
        a, b = dna.StringToBases(sa), dna.StringToBases(sb)
        if kind == "affine":
            hs, hr = align.AffineGap_highMem(a, b, MX[mx], go, ge)
            ls, lr = align.AffineGap_customizeCheckersize(a, b, MX[mx], go, ge, cs, cs)
        else:
            hs, hr = align.ConstGap_highMem(a, b, MX[mx], go)
            ls, lr = align.ConstGap_customizeCheckersize(a, b, MX[mx], go, cs, cs)
        assert (hs, align.PrintCigar(hr)) == (score, high)
        assert (ls, align.PrintCigar(lr)) == (score, low)


# ---- seeded fuzz against the oracle -----------------------------------------------------------------
def _gpu_batch(gpu_lib, mode, mx, go, ge, alphas, betas, ci=10000, cj=10000):
    return gpu_lib.align_batch(gpu_lib.make_params(mode, mx, go, ge, ci, cj), alphas, betas)


MODES = [(0, "affine"), (1, "const"), (2, "affine_highmem"), (3, "affine_local"), (4, "const_highmem")]


@pytest.mark.parametrize("mode,name", MODES)
def test_fuzz_small(gpu_lib, mode, name):
    lo = 0 if mode >= 2 else 1
    alphas, betas = common.random_pairs(1000 + mode, 3000, lo, 48, lo, 48)
    for mx, go, ge in (("Default", -400, -30), ("HumanChimpTwo", -600, -150), ("HoxD55", -400, -30)):
        got = _gpu_batch(gpu_lib, mode, MX[mx], go, ge, alphas, betas)
        exp = oracle.align_batch(mode, MX[mx], go, ge, alphas, betas, threads=8)
        common.assert_same(got, exp, "%s %s" % (name, mx))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("cs", [2, 3, 5, 7, 16])
def test_fuzz_multitile_quirks(gpu_lib, mode, cs):
    alphas, betas = common.random_pairs(2000 + cs, 2000, 1, 60, 1, 60)
    got = _gpu_batch(gpu_lib, mode, MX["Default"], -400, -30, alphas, betas, cs, cs)
    exp = oracle.align_batch(mode, MX["Default"], -400, -30, alphas, betas, cs, cs, threads=8)
    common.assert_same(got, exp)
    high = oracle.align_batch(mode + (2 if mode == 0 else 3), MX["Default"], -400, -30, alphas, betas, threads=8)
    assert not np.array_equal(got[2], high[2]) or not np.array_equal(got[1]["run_length"], high[1]["run_length"])


@pytest.mark.parametrize("mode,name", MODES)
def test_fuzz_multistrip(gpu_lib, mode, name):
    """alpha longer than one 160-row strip (row buffer hand-over between strips), ragged batch."""
    lo = 0 if mode >= 2 else 1
    alphas, betas = common.random_pairs(3000 + mode, 96, lo, 700, lo, 500, related=0.8)
    got = _gpu_batch(gpu_lib, mode, MX["HumanChimpTwo"], -600, -150, alphas, betas)
    exp = oracle.align_batch(mode, MX["HumanChimpTwo"], -600, -150, alphas, betas, threads=8)
    common.assert_same(got, exp, name)
    if mode < 2:
        got = _gpu_batch(gpu_lib, mode, MX["HumanChimpTwo"], -600, -150, alphas, betas, 100, 100)
        exp = oracle.align_batch(mode, MX["HumanChimpTwo"], -600, -150, alphas, betas, 100, 100, threads=8)
        common.assert_same(got, exp, name + " checkersize 100")


def test_c1_1kb_pair(gpu_lib):  # config C1: two ~1 kb records, ConstGap + AffineGap
    rng = np.random.default_rng(1)
    a = rng.integers(0, 4, size=1000).astype(np.uint8)
    b = common.mutate(rng, a, sub=0.05, indel=0.01, geo=0.3)
    s, r = align.ConstGap(a, b, align.HumanChimpTwoScoreMatrix, -430)
    assert (s, _route(r)) == oracle.align_one(oracle.MODE_CONST, MX["HumanChimpTwo"], -430, 0, a, b)
    s, r = align.AffineGap(a, b, align.HumanChimpTwoScoreMatrix, -600, -150)
    assert (s, _route(r)) == oracle.align_one(oracle.MODE_AFFINE, MX["HumanChimpTwo"], -600, -150, a, b)


def test_c2_reads_vs_chunk(gpu_lib):  # config C2 at a size the oracle finishes in seconds
    reads, chunk = common.c2_workload(2, 512)
    n = reads.shape[0]
    a_start = np.arange(n, dtype=np.int64) * 150
    a_len = np.full(n, 150, dtype=np.int64)
    b_start = np.zeros(n, dtype=np.int64)
    b_len = np.full(n, chunk.shape[0], dtype=np.int64)
    for mode in (gpu_lib.GNX_AFFINE_GAP, gpu_lib.GNX_CONST_GAP):
        p = gpu_lib.make_params(mode, align.HumanChimpTwoScoreMatrix, -600 if mode == 0 else -430, -150)
        got = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len)
        exp = oracle.align_batch_windows(mode, MX["HumanChimpTwo"], -600 if mode == 0 else -430, -150,
                                         reads.reshape(-1), a_start, a_len, chunk, b_start, b_len, threads=8)
        common.assert_same(got, exp)
    # second series: AffineGapLocal(target=chunk, query=read)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, align.HumanChimpTwoScoreMatrix, -600, -150)
    got = gpu_lib.align_batch_windows(p, chunk, b_start[:64], b_len[:64], reads.reshape(-1), a_start[:64], a_len[:64])
    exp = oracle.align_batch_windows(oracle.MODE_AFFINE_LOCAL, MX["HumanChimpTwo"], -600, -150,
                                     chunk, b_start[:64], b_len[:64], reads.reshape(-1), a_start[:64], a_len[:64], threads=8)
    common.assert_same(got, exp)


def test_errors(gpu_lib):
    a, b = dna.StringToBases("ACGT"), dna.StringToBases("ACGT")
    with pytest.raises(IndexError):  # lower-case bases index past the 5x5 matrix in Go
        align.ConstGap(dna.StringToBases("ACgT"), b, align.DefaultScoreMatrix, -430)
    with pytest.raises(IndexError):
        align.AffineGap(a, dna.StringToBases("AC-T"), align.DefaultScoreMatrix, -400, -30)
    with pytest.raises(ValueError):  # the Go loop never terminates on an empty sequence
        align.AffineGap(np.zeros(0, np.uint8), b, align.DefaultScoreMatrix, -400, -30)
    with pytest.raises(gpu_lib.GnxError):
        align.AffineGap(a, b, [[2 ** 40] * 5] * 5, -400, -30)
    # and the library still works afterwards
    s, r = align.AffineGap(a, b, align.DefaultScoreMatrix, -400, -30)
    assert align.PrintCigar(r) == "4M"


def test_workspace_chunking(gpu_lib, monkeypatch):
    """A small workspace limit forces several fill launches of the general path; results must not change.  (GNX_CLONG=0: without the
    switch a batch whose stored matrices do not fit goes to the snapshot path, tests/test_const_long.py)"""
    monkeypatch.setenv("GNX_CLONG", "0")
    alphas, betas = common.random_pairs(77, 600, 20, 200, 20, 300)
    exp = oracle.align_batch(0, MX["Default"], -400, -30, alphas, betas, threads=8)
    gpu_lib.check(gpu_lib.lib().gnx_init(0, 4 << 20))
    try:
        got = _gpu_batch(gpu_lib, 0, MX["Default"], -400, -30, alphas, betas)
        assert gpu_lib.get_timing()["n_launches"] > 1
    finally:
        gpu_lib.check(gpu_lib.lib().gnx_init(0, 8 << 30))
    common.assert_same(got, exp)


def test_c5_scaled_long_read_checkerboards(gpu_lib):
    """Config C5 scaled to what the oracle finishes in seconds: an ONT-like read inside a longer window, several
    checkerboards in both directions (quirk-exact tile walk), several 160-row strips."""
    rng = np.random.default_rng(5)
    win = rng.integers(0, 4, size=20000).astype(np.uint8)
    read = common.mutate(rng, win[6000:9000], sub=0.04, indel=0.06, geo=0.5)
    for mode, go, ge in ((gpu_lib.GNX_CONST_GAP, -430, 0), (gpu_lib.GNX_AFFINE_GAP, -600, -150)):
        p = gpu_lib.make_params(mode, align.HumanChimpTwoScoreMatrix, go, ge, 1000, 1000)
        got = gpu_lib.align_batch(p, [read, read[:1500]], [win, win[5000:12000]])
        exp = oracle.align_batch(mode, MX["HumanChimpTwo"], go, ge, [read, read[:1500]], [win, win[5000:12000]], 1000, 1000, threads=2)
        common.assert_same(got, exp)


def test_full_size_properties(gpu_lib):
    """Size-independent checks at BASELINE sizes where the oracle is too slow to run in a test: every CIGAR must
    consume exactly n and m, and re-scoring the CIGAR with the affine model must reproduce the returned score."""
    reads, chunk = common.c2_workload(9, 4096)
    n = reads.shape[0]
    a_start = np.arange(n, dtype=np.int64) * 150
    a_len = np.full(n, 150, dtype=np.int64)
    b_start = np.zeros(n, dtype=np.int64)
    b_len = np.full(n, chunk.shape[0], dtype=np.int64)
    sc = np.asarray(align.HumanChimpTwoScoreMatrix, dtype=np.int64)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150)
    scores, ops, off = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len)
    again = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len)
    common.assert_same((scores, ops, off), again, "idempotence")
    for k in range(0, n, 37):
        route = [(int(r), int(o)) for r, o in zip(ops["run_length"][off[k]:off[k + 1]], ops["op"][off[k]:off[k + 1]])]
        i = j = 0
        total = 0
        prev = -1
        for run, op in route:
            assert run > 0 and op != prev
            prev = op
            if op == 0:
                total += int(sc[reads[k, i:i + run], chunk[j:j + run]].sum())
                i += run; j += run
            else:
                total += -600 + -150 * run
                if op == 1:
                    j += run
                else:
                    i += run
        assert (i, j) == (150, chunk.shape[0])
        assert total == int(scores[k])


def test_fast_path_chunking_and_fallback(gpu_lib):
    """Short-alpha batches take the checkpoint/re-fill fast path; with a small workspace it runs in several
    sub-batches, with GNX_FASTPATH=0 the general path must give the same bits."""
    reads, chunk = common.c2_workload(21, 600, read_len=150, chunk_len=3000)
    n = reads.shape[0]
    a_start = np.arange(n, dtype=np.int64) * 150
    a_len = np.full(n, 150, dtype=np.int64)
    b_start = np.zeros(n, dtype=np.int64)
    b_len = np.full(n, chunk.shape[0], dtype=np.int64)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150)
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len, threads=8)
    got = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len)
    if os.environ.get("GNX_FASTPATH", "1") != "0":
        common.expect_route(gpu_lib.get_timing(), 1)
    common.assert_same(got, exp, "fast path")
    gpu_lib.check(gpu_lib.lib().gnx_init(0, 12 << 20))
    try:
        got = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len)
        tm = gpu_lib.get_timing()
        if os.environ.get("GNX_FASTPATH", "1") != "0" and "GNX_FP_MAXIT" not in os.environ:  # (MAXIT=0 needs more tile workspace than 12 MB)
            common.expect_route(tm, 1)
            assert common.route_switched() or tm["dominant_launches"] > 1
    finally:
        gpu_lib.check(gpu_lib.lib().gnx_init(0, 8 << 30))
    common.assert_same(got, exp, "fast path, chunked")
    # mixed read lengths stay on the fast path (rows are right-aligned per pair in fp_sweep_kernel)
    reads2 = [reads[k, :150 - (k % 3)] for k in range(64)]
    got = gpu_lib.align_batch(p, reads2, [chunk] * 64)
    if os.environ.get("GNX_FASTPATH", "1") != "0":
        common.expect_route(gpu_lib.get_timing(), 1)
    exp = oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, reads2, [chunk] * 64, threads=8)
    common.assert_same(got, exp, "mixed lengths")
    # a matrix whose 4*(s - 2e) does not fit the int16 profile of the sweep kernel -> general path, same answers
    big = [[v * 40 for v in row] for row in align.HumanChimpTwoScoreMatrix]
    pb = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, big, -600, -150)
    got = gpu_lib.align_batch(pb, reads2, [chunk] * 64)
    common.expect_route(gpu_lib.get_timing(), 0)
    exp = oracle.align_batch(0, np.asarray(big, dtype=np.int64), -600, -150, reads2, [chunk] * 64, threads=8)
    common.assert_same(got, exp, "big scores -> general path")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [31, 32, 33])
def test_fast_path_sweep_geometry(gpu_lib, seed, monkeypatch):
    """fp_sweep_kernel: every alpha length 1..160 (padding slots, 19 and 20 rows per lane, fewer rows than planes),
    ragged beta lengths around the checkpoint spacing, several penalty sets incl. gapOpen = 0; forced onto the fast path."""
    monkeypatch.setenv("GNX_FASTPATH", "2")
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=4000, dtype=np.uint8)
    ref[rng.integers(0, 4000, size=8)] = 4
    n_hi = [152, 160, 40][seed - 31]
    alphas, betas = [], []
    for k in range(200):
        n = int(rng.integers(1, n_hi + 1)) if k >= 12 else [1, 2, 3, 4, 5, n_hi, n_hi - 1, 19, 20, 21, 8, n_hi - 8][k]
        n = max(1, min(n, n_hi))
        m = int(rng.choice([1, 7, 8, 9, 15, 16, 17, 127, 128, 129, 255, 256, 257, 300, 1000, 1500, 2049]))
        off = int(rng.integers(0, 4000 - m + 1))
        beta = ref[off:off + m].copy()
        pos = int(rng.integers(0, max(1, m - n + 1)))
        alpha = common.mutate(rng, beta[pos:pos + n], 0.05, 0.02) if m >= n and rng.random() < 0.8 else rng.integers(0, 4, size=n, dtype=np.uint8)
        alpha = alpha[:n_hi] if alpha.shape[0] > 0 else np.array([1], dtype=np.uint8)
        alphas.append(alpha); betas.append(beta)
    for name, go, ge in [("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("HumanChimpTwo", 0, -150), ("Default", -3, 0)]:
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX[name], go, ge)
        got = gpu_lib.align_batch(p, alphas, betas)
        # cheap gaps give CIGARs with more than 64 runs, which send the batch back through the general path -- also checked
        if go == -600:
            common.expect_route(gpu_lib.get_timing(), 1)
        exp = oracle.align_batch(0, MX[name], go, ge, alphas, betas, threads=8)
        common.assert_same(got, exp, "sweep geometry %s %d %d" % (name, go, ge))
    # small checkerboards on the fast path: quirks Q1/Q2 through the staged walk
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150, 7, 7)
    got = gpu_lib.align_batch(p, alphas, betas)
    exp = oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, alphas, betas, ci=7, cj=7, threads=8)
    common.assert_same(got, exp, "sweep geometry, 7x7 checkerboards")


@pytest.mark.gpu
def test_fast_path_long_cigars_redo_only_those_pairs(gpu_lib):
    """A few pairs whose CIGAR has more runs than the fast path stages (64) are aligned again on the general path --
    the rest of the batch stays on the fast path; scores, offsets and runs must still be those of the oracle."""
    rng = np.random.default_rng(77)
    chunk = rng.integers(0, 4, size=1500).astype(np.uint8)
    alphas, betas = [], []
    for k in range(96):
        off = int(rng.integers(0, 1200))
        if k % 11 == 3:
            a = chunk[off:off + 280:2][:140].copy()      # every other base: with cheap gaps the alignment alternates M and I
        else:
            a = common.mutate(rng, chunk[off:off + 150], 0.02, 0.005)[:150]
        alphas.append(a); betas.append(chunk)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["Default"], -30, -10)
    got = gpu_lib.align_batch(p, alphas, betas)
    if os.environ.get("GNX_FASTPATH", "1") != "0":
        common.expect_route(gpu_lib.get_timing(), 1)
    exp = oracle.align_batch(0, MX["Default"], -30, -10, alphas, betas, threads=8)
    long_ones = sum(1 for k in range(96) if int(exp[2][k + 1] - exp[2][k]) > 64)
    assert 1 <= long_ones <= 24
    common.assert_same(got, exp, "long CIGARs redone on the general path")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [41, 42, 43])
def test_fast_path_local_transposed(gpu_lib, seed, monkeypatch):
    """AffineGapLocal(target = long alpha, query = short beta) runs the fast path on the transposed problem
    (fp_sweep_kernel<.., XP>: swapped gap tags, free row 0, free last-row step, fake padding rows): every query length
    1..160, ragged target lengths around the checkpoint spacing, queries placed at the target's ends and in the middle,
    unrelated queries, several penalty sets; forced onto the fast path for short targets too."""
    monkeypatch.setenv("GNX_FASTPATH", "2")
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=4000, dtype=np.uint8)
    ref[rng.integers(0, 4000, size=8)] = 4
    q_hi = [152, 160, 40][seed - 41]
    targets, queries = [], []
    for k in range(200):
        n = int(rng.integers(1, q_hi + 1)) if k >= 12 else [1, 2, 3, 4, 5, q_hi, q_hi - 1, 19, 20, 21, 8, q_hi - 8][k]
        n = max(1, min(n, q_hi))
        m = int(rng.choice([1, 7, 8, 9, 15, 16, 17, 127, 128, 129, 255, 256, 257, 300, 1000, 1500, 2049]))
        off = int(rng.integers(0, 4000 - m + 1))
        target = ref[off:off + m].copy()
        where = k % 4
        pos = 0 if where == 0 else (max(0, m - n) if where == 1 else int(rng.integers(0, max(1, m - n + 1))))
        query = common.mutate(rng, target[pos:pos + n], 0.05, 0.02) if m >= n and rng.random() < 0.8 else rng.integers(0, 4, size=n, dtype=np.uint8)
        query = query[:q_hi] if query.shape[0] > 0 else np.array([1], dtype=np.uint8)
        targets.append(target); queries.append(query)
    for name, go, ge in [("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("HumanChimpTwo", 0, -150), ("MouseRat", -3, -1)]:
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, MX[name], go, ge)
        got = gpu_lib.align_batch(p, targets, queries)
        if go == -600:
            common.expect_route(gpu_lib.get_timing(), 1)
        exp = oracle.align_batch(3, MX[name], go, ge, targets, queries, threads=8)
        common.assert_same(got, exp, "local transposed %s %d %d" % (name, go, ge))
    # gapExtend = 0 is not eligible (the free last-row step must beat the extension strictly): general path, same answers
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, MX["Default"], -3, 0)
    got = gpu_lib.align_batch(p, targets[:64], queries[:64])
    common.expect_route(gpu_lib.get_timing(), 0)
    common.assert_same(got, oracle.align_batch(3, MX["Default"], -3, 0, targets[:64], queries[:64], threads=8), "local, gapExtend 0")


@pytest.mark.gpu
def test_fast_path_local_c2_series(gpu_lib):
    """Second series of C2: AffineGapLocal(target = chunk, query = read), default dispatch."""
    reads, chunk = common.c2_workload(23, 500, read_len=150, chunk_len=3000)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, align.HumanChimpTwoScoreMatrix, -600, -150)
    queries = [reads[k, :150 - (k % 5)] for k in range(reads.shape[0])]
    got = gpu_lib.align_batch(p, [chunk] * len(queries), queries)
    if os.environ.get("GNX_FASTPATH", "1") != "0":
        common.expect_route(gpu_lib.get_timing(), 1)
    exp = oracle.align_batch(3, MX["HumanChimpTwo"], -600, -150, [chunk] * len(queries), queries, threads=8)
    common.assert_same(got, exp, "C2 local series")
    # cheap gaps: some CIGARs overflow the staging area and are redone on the general path in the original orientation
    rng = np.random.default_rng(78)
    queries = [(chunk[o:o + 280:2][:140].copy() if k % 11 == 3 else common.mutate(rng, chunk[o:o + 150], 0.02, 0.005)[:150])
               for k, o in enumerate(rng.integers(0, 2500, size=96))]
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, MX["Default"], -30, -10)
    got = gpu_lib.align_batch(p, [chunk] * 96, queries)
    exp = oracle.align_batch(3, MX["Default"], -30, -10, [chunk] * 96, queries, threads=8)
    common.assert_same(got, exp, "local, long CIGARs redone")


@pytest.mark.gpu
def test_windows_at_offsets_beyond_4gb(gpu_lib):
    """Config C3 shape: reads against windows of ONE big resident reference, addressed by 64-bit offsets.  The
    reference here is 4.3 GB (mostly untouched zero pages) with windows just below / above 2^31 and 2^32: offset arithmetic
    in the planner and the kernels must be 64-bit.  Fast path (150 x 3000) and general path (300 x 900)."""
    rng = np.random.default_rng(61)
    total = (1 << 32) + (1 << 16)
    ref = np.zeros(total, dtype=np.uint8)
    spots = [0, (1 << 31) - 5000, (1 << 31) + 17, (1 << 32) - 9000, (1 << 32) + 100]
    for sp in spots:
        ref[sp:sp + 12000] = rng.integers(0, 4, size=12000, dtype=np.uint8)
    for n, m in ((150, 3000), (300, 900)):
        reads, a_start, a_len, b_start, b_len = [], [], [], [], []
        for k in range(40):
            sp = spots[k % len(spots)] + int(rng.integers(0, 12000 - m))
            pos = int(rng.integers(0, m - n))
            r = common.mutate(rng, ref[sp + pos:sp + pos + n], 0.03, 0.01)[:n]
            a_start.append(sum(len(x) for x in reads)); a_len.append(len(r)); reads.append(r)
            b_start.append(sp); b_len.append(m)
        a_buf = np.concatenate(reads)
        a_start, a_len, b_start, b_len = (np.asarray(x, dtype=np.int64) for x in (a_start, a_len, b_start, b_len))
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
        got = gpu_lib.align_batch_windows(p, a_buf, a_start, a_len, ref, b_start, b_len)
        exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, a_buf, a_start, a_len, ref, b_start, b_len, threads=8)
        common.assert_same(got, exp, "windows beyond 4 GB, %d x %d" % (n, m))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,name", MODES)
def test_pipelined_strips(gpu_lib, mode, name):
    """Small launches of multi-strip pairs with beta >= 1024 run their strips as pipelined workgroups (per-strip row buffer,
    progress words, non-temporal direction stores): ragged lengths, one-strip pairs in the same launch, checkerboards, and a launch
    big enough to fall back to one wave per 4 pairs must all equal the oracle."""
    lo = 0 if mode >= 2 else 1
    alphas, betas = common.random_pairs(5000 + mode, 40, lo, 900, 1024, 1500, related=0.8)
    alphas[3] = alphas[3][:100]            # a one-strip pair inside a pipelined launch
    alphas[7] = alphas[7][:161]            # two strips, the second one a single row
    got = _gpu_batch(gpu_lib, mode, MX["HumanChimpTwo"], -600, -150, alphas, betas)
    exp = oracle.align_batch(mode, MX["HumanChimpTwo"], -600, -150, alphas, betas, threads=8)
    common.assert_same(got, exp, name + " pipelined")
    if mode < 2:
        got = _gpu_batch(gpu_lib, mode, MX["HumanChimpTwo"], -600, -150, alphas, betas, 300, 300)
        exp = oracle.align_batch(mode, MX["HumanChimpTwo"], -600, -150, alphas, betas, 300, 300, threads=8)
        common.assert_same(got, exp, name + " pipelined, checkersize 300")
    # one long pair alone (the cigarToBed shape), and the same launch repeated (progress words are reset per launch)
    a1, b1 = common.random_pairs(5100 + mode, 1, 2500, 2500, 3000, 3000, related=1.0)
    for _ in range(2):
        got = _gpu_batch(gpu_lib, mode, MX["Default"], -400, -30, a1, b1)
        exp = oracle.align_batch(mode, MX["Default"], -400, -30, a1, b1, threads=1)
        common.assert_same(got, exp, name + " one long pair")
    # reads placed inside much longer windows: long leading / trailing horizontal runs, which the wave-cooperative traceback of
    # small launches takes 64 direction words at a time (and the mirrored shape: long vertical runs)
    rng = np.random.default_rng(5200 + mode)
    alphas, betas = [], []
    for n, m in [(170, 5000), (700, 6000), (1500, 4100), (161, 9000), (320, 1040)]:
        w = rng.integers(0, 4, size=m).astype(np.uint8)
        off = int(rng.integers(0, m - n))
        alphas.append(common.mutate(rng, w[off:off + n], 0.04, 0.03)[:n]); betas.append(w)
    if mode == 3:  # AffineGapLocal(target, query): keep the query short enough to stay multi-strip on the target side
        alphas, betas = betas, alphas
    for go, ge in [(-600, -150), (0, -30)]:
        got = _gpu_batch(gpu_lib, mode, MX["HumanChimpTwo"], go, ge, alphas, betas)
        exp = oracle.align_batch(mode, MX["HumanChimpTwo"], go, ge, alphas, betas, threads=5)
        common.assert_same(got, exp, name + " long gaps %d %d" % (go, ge))
    got = _gpu_batch(gpu_lib, mode, MX["HumanChimpTwo"], -600, -150, betas, alphas)
    exp = oracle.align_batch(mode, MX["HumanChimpTwo"], -600, -150, betas, alphas, threads=5)
    common.assert_same(got, exp, name + " long gaps, mirrored")
    # the cooperative walk is a single pass with staged runs by default; the count + write form is the fallback for huge launches
    os.environ["GNX_TB_TWO_PASS"] = "1"
    try:
        got = _gpu_batch(gpu_lib, mode, MX["HumanChimpTwo"], -600, -150, betas, alphas)
    finally:
        del os.environ["GNX_TB_TWO_PASS"]
    common.assert_same(got, exp, name + " long gaps, mirrored, two-pass walk")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [51, 52, 53])
def test_fast_path_two_row_blocks(gpu_lib, seed, monkeypatch):
    """reads of 161 .. 320 bases on the fast path: two row blocks (fp_sweep_kernel<20, false, 1 / 2>: the top block hands its bottom row
    to the bottom block), windows and straggler tiles re-filled as two strips; every length, ragged windows around the checkpoint
    spacing, several penalty sets, small checkerboards (quirks Q1 / Q2 across the strip border), forced window / tile rounds"""
    monkeypatch.setenv("GNX_FASTPATH", "2")
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=6000, dtype=np.uint8)
    ref[rng.integers(0, 6000, size=10)] = 4
    alphas, betas = [], []
    n_top = {51: 320, 52: 224, 53: 288}[seed]  # the top block runs with 20 / 8 / 16 slots per lane
    for k in range(180):
        n = int(rng.integers(161, n_top + 1)) if k >= 8 else min([161, 162, 320, 319, 200, 250, 165, 300][k], n_top)
        m = int(rng.choice([127, 128, 129, 255, 256, 257, 330, 1000, 1500, 2049, 3000]))
        off = int(rng.integers(0, 6000 - m + 1))
        beta = ref[off:off + m].copy()
        if m >= n and rng.random() < 0.8:
            pos = int(rng.integers(0, m - n + 1))
            alpha = common.mutate(rng, beta[pos:pos + n + 20], 0.04, 0.015)
        else:
            alpha = rng.integers(0, 4, size=n, dtype=np.uint8)
        alpha = alpha[:n]
        if alpha.shape[0] < 161:  # (mutate may have shortened it)
            alpha = np.concatenate([alpha, rng.integers(0, 4, size=161 - alpha.shape[0], dtype=np.uint8)])
        alphas.append(alpha); betas.append(beta)
    for name, go, ge in [("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("HumanChimpTwo", 0, -150)]:
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX[name], go, ge)
        got = gpu_lib.align_batch(p, alphas, betas)
        if go == -600:
            common.expect_route(gpu_lib.get_timing(), 1)
        exp = oracle.align_batch(0, MX[name], go, ge, alphas, betas, threads=8)
        common.assert_same(got, exp, "two row blocks %s %d %d" % (name, go, ge))
    exp7 = oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, alphas, betas, ci=7, cj=7, threads=8)
    p7 = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150, 7, 7)
    for maxit in (None, "0", "3"):
        if maxit is not None:
            monkeypatch.setenv("GNX_FP_MAXIT", maxit)
        common.assert_same(gpu_lib.align_batch(p7, alphas, betas), exp7, "two row blocks, 7x7 checkerboards, maxit %s" % maxit)
    monkeypatch.delenv("GNX_FP_MAXIT", raising=False)
    # a C2-shaped batch of 250-base reads (the shape this exists for) without the forcing switch
    monkeypatch.delenv("GNX_FASTPATH", raising=False)
    reads, chunk = common.c2_workload(seed, 600, read_len=250, chunk_len=5000)
    a_start = np.arange(600, dtype=np.int64) * 250
    a_len = np.full(600, 250, dtype=np.int64)
    b_start = np.zeros(600, dtype=np.int64)
    b_len = np.full(600, 5000, dtype=np.int64)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    got = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len)
    common.expect_route(gpu_lib.get_timing(), 1)
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len, threads=8)
    common.assert_same(got, exp, "250-base reads")


@pytest.mark.gpu
@pytest.mark.parametrize("alphabet", [2, 4])
def test_fast_path_event_tracking(gpu_lib, monkeypatch, alphabet):
    """Round 4: the headline sweep keeps no I-planes in its steady part -- per plane row only the last step at which the gap could have
    been opened (h' + o >= I'), the walk strides to that cell and decides there.  The cases that stress it: TIES (two-letter and
    homopolymer sequences: equal-scoring alternatives everywhere, opens that tie with extensions), windows of very different lengths in
    one wave of 8 pairs (the tagged tail starts at the shortest window's end: the long windows' last thousands of columns run tagged),
    reads that end exactly at the window's end (trailing gap on rows n-1 .. n-3 and beyond), reads placed at the window's start,
    penalties with gapOpen = 0 (every step an event) and tiny extensions."""
    monkeypatch.setenv("GNX_FASTPATH", "2")
    rng = np.random.default_rng(500 + alphabet)
    ref = rng.integers(0, alphabet, size=6000, dtype=np.uint8)
    ref[2000:2300] = 1                      # a homopolymer stretch
    ref[3000:3400] = np.tile([0, 1], 200)   # a dinucleotide repeat
    alphas, betas = [], []
    for k in range(320):
        n = int(rng.integers(1, 153))
        m = int(rng.choice([16, 40, 130, 300, 900, 2500, 5000])) if k % 8 else 5000  # every wave holds one long window
        off = int(rng.integers(0, 6000 - m + 1))
        beta = ref[off:off + m].copy()
        kind = k % 5
        if kind == 0 and m > n:      # read = the window's last bases: the path ends in the corner
            alpha = beta[m - n:].copy()
        elif kind == 1 and m > n:    # ... its first bases: one long trailing gap on row n
            alpha = beta[:n].copy()
        elif kind == 2 and m > n + 4:  # the last d bases match the window's end, the rest sits further left: trailing gap on row n - d
            d = int(rng.integers(1, 7)); pos = int(rng.integers(0, m - n - d + 1))
            alpha = np.concatenate([beta[pos:pos + n - d], beta[m - d:]]) if n > d else beta[m - n:].copy()
        else:
            pos = int(rng.integers(0, max(1, m - n + 1)))
            alpha = common.mutate(rng, beta[pos:pos + n], 0.04, 0.03, alphabet=alphabet)[:152] if m >= n else rng.integers(0, alphabet, size=n, dtype=np.uint8)
        if alpha.shape[0] == 0:
            alpha = np.array([1], dtype=np.uint8)
        alphas.append(alpha.astype(np.uint8)); betas.append(beta)
    for name, go, ge in [("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("Default", -100, -100), ("HumanChimpTwo", 0, -150), ("HoxD55", -1, -1)]:
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX[name], go, ge)
        got = gpu_lib.align_batch(p, alphas, betas)
        exp = oracle.align_batch(0, MX[name], go, ge, alphas, betas, threads=8)
        common.assert_same(got, exp, "event tracking, alphabet %d, %s %d %d" % (alphabet, name, go, ge))
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150, 64, 64)  # checkerboard edges inside the windows (quirks)
    common.assert_same(gpu_lib.align_batch(p, alphas, betas), oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, alphas, betas, 64, 64, threads=8), "event tracking, 64 x 64 checkerboards")


@pytest.mark.gpu
def test_fast_path_mixed_read_lengths(gpu_lib, monkeypatch):
    """a batch that mixes reads of <= 160 and of 161 .. 320 bases: two uniform sub-batches, each on its fast path, merged back into
    input order (run_device, the block under 'mixed with shorter reads')"""
    rng = np.random.default_rng(77)
    reads, chunk = common.c2_workload(77, 900, read_len=320, chunk_len=4000)
    lens = rng.choice([36, 100, 150, 160, 161, 200, 250, 320], size=900).astype(np.int64)
    a_start = np.arange(900, dtype=np.int64) * 320
    b_len = rng.choice([800, 1500, 4000], size=900).astype(np.int64)
    b_start = rng.integers(0, 4000 - b_len + 1).astype(np.int64)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    got = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, lens, chunk, b_start, b_len)
    common.expect_route(gpu_lib.get_timing(), 1)
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, reads.reshape(-1), a_start, lens, chunk, b_start, b_len, threads=8)
    common.assert_same(got, exp, "mixed read lengths")
    # every read in one group: still the uniform paths
    for sel in (lens <= 160, lens > 160):
        got1 = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start[sel], lens[sel], chunk, b_start[sel], b_len[sel])
        exp1 = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, reads.reshape(-1), a_start[sel], lens[sel], chunk, b_start[sel], b_len[sel], threads=8)
        common.assert_same(got1, exp1, "uniform sub-batch")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["global", "lowmem", "local"])
def test_fast_path_big_batch_short_windows(gpu_lib, mode):
    """batches of >= 8192 pairs take the fast path at ANY window length (run_device: min_cols 32; the general path's host-built plans
    cost more than its kernels there): 9000 pairs with windows of 32 .. 400 columns and reads of 1 .. 200 bases (one and two row
    blocks, reads longer than their window, windows with N), natural routing, against the oracle; windows below 32 columns in the
    same batch go the general way and come back in input order"""
    rng = np.random.default_rng(123)
    n_pairs, L = 9000, 3000
    ref = rng.integers(0, 4, size=L, dtype=np.uint8)
    ref[rng.integers(0, L, size=12)] = 4
    lens = rng.integers(1, 201, size=n_pairs).astype(np.int64)
    b_len = rng.choice([32, 33, 47, 64, 100, 150, 216, 256, 400], size=n_pairs).astype(np.int64)
    b_len[::500] = rng.integers(1, 32, size=len(b_len[::500]))
    b_start = rng.integers(0, L - b_len + 1).astype(np.int64)
    a_start = np.zeros(n_pairs, dtype=np.int64)
    a_start[1:] = np.cumsum(lens)[:-1]
    reads = np.empty(int(lens.sum()), dtype=np.uint8)
    for k in range(n_pairs):
        n, m = int(lens[k]), int(b_len[k])
        src = ref[b_start[k]:b_start[k] + m]
        s0 = int(rng.integers(0, max(m - n, 0) + 1))
        seg = src[s0:s0 + n].copy()
        if len(seg) < n:
            seg = np.concatenate([seg, rng.integers(0, 4, size=n - len(seg), dtype=np.uint8)])
        flip = rng.random(n) < 0.06
        seg[flip] = rng.integers(0, 4, size=int(flip.sum()), dtype=np.uint8)
        reads[a_start[k]:a_start[k] + n] = seg
    if mode == "local":  # AffineGapLocal(target = window, query = read): the transposed fast path
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, MX["HumanChimpTwo"], -600, -150)
        got = gpu_lib.align_batch_windows(p, ref, b_start, b_len, reads, a_start, lens)
        exp = oracle.align_batch_windows(gpu_lib.GNX_AFFINE_GAP_LOCAL, MX["HumanChimpTwo"], -600, -150, ref, b_start, b_len, reads, a_start, lens, threads=8)
    else:
        ck = (7, 7) if mode == "lowmem" else (10000, 10000)
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150, *ck)
        got = gpu_lib.align_batch_windows(p, reads, a_start, lens, ref, b_start, b_len)
        exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, reads, a_start, lens, ref, b_start, b_len, *ck, threads=8)
    common.expect_route(gpu_lib.get_timing(), 1)
    common.assert_same(got, exp, "big batch of short windows, " + mode)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_lo,n_hi", [(61, 321, 480), (62, 481, 640), (63, 1441, 1600), (64, 700, 800), (65, 3041, 3200)])
def test_fast_path_many_row_blocks(gpu_lib, seed, n_lo, n_hi, monkeypatch):
    """reads of 321 .. 3200 bases on the fast path: a top block, middle blocks (fp_sweep_kernel<20, false, 3>: take the row above from the
    row buffer and hand their own bottom row down in place) and the bottom block; windows and straggler tiles re-filled as S strips.
    Every number of slots per lane of the top block, ragged windows, several penalty sets, small checkerboards, forced rounds."""
    monkeypatch.setenv("GNX_FASTPATH", "2")
    rng = np.random.default_rng(seed)
    L = 7000
    ref = rng.integers(0, 4, size=L, dtype=np.uint8)
    ref[rng.integers(0, L, size=10)] = 4
    S = (n_hi + 159) // 160
    alphas, betas = [], []
    for k in range(96):
        if (n_hi - 1) // 160 != (n_lo - 1) // 160:  # (seed 64: all in one class anyway)
            raise AssertionError("bad test parameters")
        n = int(rng.integers(n_lo, n_hi + 1)) if k >= 4 else [n_lo, n_hi, n_lo + 1, n_hi - 1][k]
        m = int(rng.choice([n // 2, n, n + 129, 2 * n + 1, 3 * n, 5000]))
        m = min(m, L)
        off = int(rng.integers(0, L - m + 1))
        beta = ref[off:off + m].copy()
        if m >= n and rng.random() < 0.8:
            pos = int(rng.integers(0, m - n + 1))
            alpha = common.mutate(rng, beta[pos:pos + n + 40], 0.04, 0.015)
        else:
            alpha = rng.integers(0, 4, size=n, dtype=np.uint8)
        alpha = alpha[:n]
        if alpha.shape[0] < n_lo:
            alpha = np.concatenate([alpha, rng.integers(0, 4, size=n_lo - alpha.shape[0], dtype=np.uint8)])
        assert (alpha.shape[0] + 159) // 160 == S
        alphas.append(alpha); betas.append(beta)
    for name, go, ge in [("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("HumanChimpTwo", 0, -150)]:
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX[name], go, ge)
        got = gpu_lib.align_batch(p, alphas, betas)
        if go == -600:
            common.expect_route(gpu_lib.get_timing(), 1)
        exp = oracle.align_batch(0, MX[name], go, ge, alphas, betas, threads=8)
        common.assert_same(got, exp, "%d row blocks %s %d %d" % (S, name, go, ge))
    exp7 = oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, alphas, betas, ci=7, cj=7, threads=8)
    p7 = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150, 7, 7)
    for maxit in (None, "0", "3"):
        if maxit is not None:
            monkeypatch.setenv("GNX_FP_MAXIT", maxit)
        common.assert_same(gpu_lib.align_batch(p7, alphas, betas), exp7, "%d row blocks, 7x7 checkerboards, maxit %s" % (S, maxit))
    monkeypatch.setenv("GNX_NO_PIPE", "1")  # one launch per row block instead of one launch whose levels follow each other
    common.assert_same(gpu_lib.align_batch(p7, alphas, betas), exp7, "%d row blocks, a launch per level" % S)
    monkeypatch.delenv("GNX_NO_PIPE")
    monkeypatch.delenv("GNX_FP_MAXIT")
    monkeypatch.setenv("GNX_WALK_LANE", "1")  # what batches of more than 32 768 such reads get: one lane per pair, one window per request
    common.assert_same(gpu_lib.align_batch(p7, alphas, betas), exp7, "%d row blocks, lanes" % S)
    monkeypatch.delenv("GNX_WALK_LANE")
    monkeypatch.setenv("GNX_FP_SPEC", "2")   # one speculative window instead of three
    common.assert_same(gpu_lib.align_batch(p7, alphas, betas), exp7, "%d row blocks, two windows per request" % S)


@pytest.mark.gpu
def test_fast_path_row_blocks_natural_routing(gpu_lib):
    """6400 reads of 330 .. 480 bases against 1500-base windows without any switch: three row blocks by the routing rule of run_device
    (pairs x row blocks >= 8192, windows >= 768 columns); and the same batch mixed with shorter reads and with pairs that are not for the fast path."""
    rng = np.random.default_rng(91)
    P, L = 6400, 1500
    reads, chunk = common.c2_workload(91, P, read_len=480, chunk_len=L)
    lens = rng.integers(330, 481, size=P).astype(np.int64)
    a_start = np.arange(P, dtype=np.int64) * 480
    b_start = np.zeros(P, dtype=np.int64)
    b_len = np.full(P, L, dtype=np.int64)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    got = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, lens, chunk, b_start, b_len)
    common.expect_route(gpu_lib.get_timing(), 1)
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, reads.reshape(-1), a_start, lens, chunk, b_start, b_len, threads=16)
    common.assert_same(got, exp, "three row blocks, natural routing")
    # mixed: 1 .. 3 row blocks, and windows too short for the fast path (general path for those)
    lens2 = rng.choice([100, 150, 200, 320, 400, 480], size=P).astype(np.int64)
    b_len2 = rng.choice([600, 1000, 1500], size=P).astype(np.int64)
    got = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, lens2, chunk, b_start, b_len2)
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, reads.reshape(-1), a_start, lens2, chunk, b_start, b_len2, threads=16)
    common.assert_same(got, exp, "mixed numbers of row blocks")


@pytest.mark.gpu
def test_fast_path_long_reads(gpu_lib, monkeypatch):
    """AffineGap(20 kb ONT-like read, 100 kb window) on the fast path: 125 row blocks in one launch, 125 walk rounds (SURVEY's C5 shape with
    the affine recurrence; two pairs are too few for the routing rule, so the path is forced)"""
    import bench
    monkeypatch.setenv("GNX_FASTPATH", "2")
    reads, wins = bench.make_long_workload(5, 2, n=20000, m=100000)
    alphas = [reads[0], reads[1][:19900]]
    betas = [wins[0], wins[1][:99000]]
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    got = gpu_lib.align_batch(p, alphas, betas)
    common.expect_route(gpu_lib.get_timing(), 1)
    exp = oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, alphas, betas, threads=2)
    common.assert_same(got, exp, "20 kb x 100 kb on the fast path")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,q_lo,q_hi", [(71, 161, 320), (72, 321, 480), (73, 700, 800)])
def test_fast_path_local_transposed_row_blocks(gpu_lib, seed, q_lo, q_hi, monkeypatch):
    """AffineGapLocal with a query of more than 160 bases: the transposed fast path with several row blocks (free row 0 above the top
    block only, free last-row step in the bottom block only, fake padding rows in the top block); queries at the target's ends and in
    the middle, unrelated queries, targets shorter than the query, several penalty sets, mixed with short queries."""
    monkeypatch.setenv("GNX_FASTPATH", "2")
    rng = np.random.default_rng(seed)
    L = 5000
    ref = rng.integers(0, 4, size=L, dtype=np.uint8)
    ref[rng.integers(0, L, size=8)] = 4
    targets, queries = [], []
    for k in range(120):
        n = int(rng.integers(q_lo, q_hi + 1)) if k >= 4 else [q_lo, q_hi, q_lo + 1, q_hi - 1][k]
        m = int(rng.choice([n // 3, n - 1, n, n + 127, n + 129, 2 * n + 1, 2049, 4000]))
        m = max(1, min(m, L))
        off = int(rng.integers(0, L - m + 1))
        target = ref[off:off + m].copy()
        where = k % 4
        pos = 0 if where == 0 else (max(0, m - n) if where == 1 else int(rng.integers(0, max(1, m - n + 1))))
        query = common.mutate(rng, target[pos:pos + n + 30], 0.05, 0.02) if m >= n and rng.random() < 0.8 else rng.integers(0, 4, size=n, dtype=np.uint8)
        query = query[:n]
        if query.shape[0] < q_lo:
            query = np.concatenate([query, rng.integers(0, 4, size=q_lo - query.shape[0], dtype=np.uint8)])
        targets.append(target); queries.append(query)
    for name, go, ge in [("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("HumanChimpTwo", 0, -150), ("MouseRat", -3, -1)]:
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, MX[name], go, ge)
        got = gpu_lib.align_batch(p, targets, queries)
        if go == -600:
            common.expect_route(gpu_lib.get_timing(), 1)
        exp = oracle.align_batch(3, MX[name], go, ge, targets, queries, threads=8)
        common.assert_same(got, exp, "local transposed, row blocks %s %d %d" % (name, go, ge))
    # mixed with short queries (sub-batches per number of row blocks), forced straggler rounds, a launch per level
    queries2 = [q[:int(rng.integers(1, 161))] if k % 3 == 0 else q for k, q in enumerate(queries)]
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, MX["HumanChimpTwo"], -600, -150)
    exp = oracle.align_batch(3, MX["HumanChimpTwo"], -600, -150, targets, queries2, threads=8)
    common.assert_same(gpu_lib.align_batch(p, targets, queries2), exp, "local transposed, mixed query lengths")
    monkeypatch.setenv("GNX_FP_MAXIT", "0")
    common.assert_same(gpu_lib.align_batch(p, targets, queries2), exp, "local transposed, tiles for everyone")
    monkeypatch.delenv("GNX_FP_MAXIT")
    monkeypatch.setenv("GNX_NO_PIPE", "1")
    common.assert_same(gpu_lib.align_batch(p, targets, queries2), exp, "local transposed, a launch per level")


@pytest.mark.gpu
def test_stress_regression_q1_at_block_top(gpu_lib, monkeypatch):
    """a batch tools/stress.py found (round 2): 64 x 64 checkerboards, reads of 1 .. 5 row blocks against 568-base windows, tiles for
    everyone.  A vertical step out of a row block's first row in the first column of a window / tile, on a checkerboard edge (quirk Q1),
    needs the argmax tag of a cell of the block above: it is the key that block handed down (fp_walk_kernel, `i == wrow`)."""
    d = np.load(os.path.join(common.DATA, "stress_r2_q1_block_top.npz"))  # plain arrays (concatenated sequences + offsets): no pickle
    mode, go, ge, cs = int(d["mode"]), int(d["go"]), int(d["ge"]), int(d["cs"])
    mx = d["mx"]
    ao, bo = d["alpha_off"], d["beta_off"]
    alphas = [np.asarray(d["alpha_cat"][ao[k]:ao[k + 1]], dtype=np.uint8) for k in range(ao.shape[0] - 1)]
    betas = [np.asarray(d["beta_cat"][bo[k]:bo[k + 1]], dtype=np.uint8) for k in range(bo.shape[0] - 1)]
    exp = oracle.align_batch(mode, mx, go, ge, alphas, betas, cs, cs, threads=8)
    p = gpu_lib.make_params(mode, mx, go, ge, cs, cs)
    monkeypatch.setenv("GNX_FASTPATH", "2")
    for maxit in ("0", "1", None):
        if maxit is None:
            monkeypatch.delenv("GNX_FP_MAXIT")
        else:
            monkeypatch.setenv("GNX_FP_MAXIT", maxit)
        common.assert_same(gpu_lib.align_batch(p, alphas, betas), exp, "stress batch, maxit %s" % maxit)


@pytest.mark.gpu
def test_small_batch_routing(gpu_lib, monkeypatch):
    """Round 4: a batch of fewer than 3072 one-block reads goes to the general path (its time is one wave's chain of steps either way, and
    the general path has its directions when the sweep ends); GNX_FP_SMALL=1 (what the suites run with) and big batches keep the fast path.
    Same bits on both."""
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150)
    reads, chunk = common.c2_workload(31, 96, read_len=150, chunk_len=2500)
    n = reads.shape[0]
    a_start, a_len = np.arange(n, dtype=np.int64) * 150, np.full(n, 150, dtype=np.int64)
    b_start, b_len = np.zeros(n, dtype=np.int64), np.full(n, chunk.shape[0], dtype=np.int64)
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len, threads=8)
    for small, route in (("1", 1), ("0", 3)):  # 3: the stored-matrix path in its latency geometry (96 reads = 192 strips of 128 rows: a small launch)
        monkeypatch.setenv("GNX_FP_SMALL", small)
        got = gpu_lib.align_batch_windows(p, reads.reshape(-1), a_start, a_len, chunk, b_start, b_len)
        if not common.OUTER_ROUTE_SWITCH:
            assert gpu_lib.get_timing()["fast_path"] == route
        common.assert_same(got, exp, "GNX_FP_SMALL=%s" % small)
    # one pair per call (a loop of align.AffineGap calls): the latency geometry by default, the general path's 16 x 10 mapping with GNX_LAT=0
    monkeypatch.delenv("GNX_FP_SMALL")
    got = gpu_lib.align_batch_windows(p, reads[0], a_start[:1], a_len[:1], chunk, b_start[:1], b_len[:1])
    if not common.OUTER_ROUTE_SWITCH:
        assert gpu_lib.get_timing()["fast_path"] == 3
    common.assert_same(got, (exp[0][:1], exp[1][:int(exp[2][1])], exp[2][:2]), "one pair")
    monkeypatch.setenv("GNX_LAT", "0")
    got = gpu_lib.align_batch_windows(p, reads[0], a_start[:1], a_len[:1], chunk, b_start[:1], b_len[:1])
    if not common.OUTER_ROUTE_SWITCH:
        assert gpu_lib.get_timing()["fast_path"] == 0
    common.assert_same(got, (exp[0][:1], exp[1][:int(exp[2][1])], exp[2][:2]), "one pair")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [330, 1000, 1120])
def test_row_block_shortcut(gpu_lib, n, monkeypatch):
    """Round 4 (fp_walk.hip.h: block_diag): reads of several row blocks whose alignment crosses whole blocks on a plain diagonal -- no indel,
    one indel in the bottom / a middle / the top block, two indels, an indel right on a block boundary, N bases, a read that starts at
    column 0 of its window -- take those blocks without a window.  Against the oracle with the default and with small / odd checkerboards
    (quirk Q1 at and inside the skipped blocks), other penalties, lanes instead of waves, forced tile rounds."""
    monkeypatch.setenv("GNX_FASTPATH", "2")
    rng = np.random.default_rng(1000 + n)
    L = n + 400
    ref = rng.integers(0, 4, size=L, dtype=np.uint8)
    S = (n + 159) // 160
    top_rows = n - 160 * (S - 1)
    alphas, betas = [], []
    cuts = [None, 5, 80, 159, 160, 161, 160 + 77, n - top_rows - 1, n - top_rows, n - top_rows + 1, n - 3, (40, n - 200), (200, 360)]
    for k in range(104):
        off = 0 if k % 13 == 12 else int(rng.integers(1, 300))
        src = ref[off:off + n + 40].copy()
        cut = cuts[k % len(cuts)]
        rows = []  # positions counted from the read's END (row n - x), so that the indel lands in the block the case names
        for x in ([] if cut is None else (cut if isinstance(cut, tuple) else (cut,))):
            rows.append(max(2, min(n - 2, n - int(x))))
        alpha = src[:n + 20].copy()
        for r in sorted(rows, reverse=True):
            ln = int(rng.integers(1, 5))
            if k % 2:
                alpha = np.concatenate([alpha[:r], alpha[r + ln:]])          # deletion from the read
            else:
                alpha = np.concatenate([alpha[:r], rng.integers(0, 4, size=ln, dtype=np.uint8), alpha[r:]])
        alpha = alpha[:n].copy()
        sub = rng.random(n) < (0.0 if k % 5 == 0 else 0.01)
        alpha[sub] = rng.integers(0, 4, size=int(sub.sum()))
        if k % 17 == 3:
            alpha[int(rng.integers(0, n))] = 4
        alphas.append(alpha); betas.append(ref.copy())
    for name, go, ge, ci, cj in [("HumanChimpTwo", -600, -150, 10000, 10000), ("HumanChimpTwo", -600, -150, 7, 7), ("HumanChimpTwo", -600, -150, 160, 160),
                                 ("HumanChimpTwo", -600, -150, 53, 53), ("Default", -400, -30, 10000, 10000), ("HumanChimpTwo", 0, -150, 31, 31)]:
        # (square checkerboards only: with ci != cj the reference's own index expression runs out of range on inputs like these -- the oracle refuses them)
        exp = oracle.align_batch(0, MX[name], go, ge, alphas, betas, ci=ci, cj=cj, threads=8)
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX[name], go, ge, ci, cj)
        common.assert_same(gpu_lib.align_batch(p, alphas, betas), exp, "%d rows %s %d %d checker %d x %d" % (n, name, go, ge, ci, cj))
        if ci == 7:
            for sw, val in (("GNX_WALK_LANE", "1"), ("GNX_FP_MAXIT", "0"), ("GNX_FP_SPEC", "1"), ("GNX_NO_PIPE", "1")):
                monkeypatch.setenv(sw, val)
                common.assert_same(gpu_lib.align_batch(p, alphas, betas), exp, "%d rows, %s=%s" % (n, sw, val))
                monkeypatch.delenv(sw)
