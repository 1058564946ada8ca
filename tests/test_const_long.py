"""Constant-gap alignments without a stored direction matrix (csrc/const_long.hip.h: score-only sweep with snapshots, fused
re-fill + walk) against the CPU oracle -- bit-exact -- and config C5 at FULL size: ConstGap(20 kb read, 100 kb window,
HumanChimpTwo, -430) with the reference's 10 000 x 10 000 checkerboards (align/constGap.go:13-68), quirk Q2 included."""
import os

import numpy as np
import pytest

import common
import oracle
from gonomics_amd import align

pytestmark = pytest.mark.gpu
MX = common.matrices()


@pytest.fixture(autouse=True)
def _sixteen_lane_kernels(monkeypatch):
    """this suite is the 16-lane snapshot kernels' (cl_sweep_wg_kernel / cl_sweep_kernel / cl_walk_*): launches of up to 64 long pairs would go to the 64-lane
    kernels + walk farm (tests/test_long_range.py has their suites) unless GNX_W64 says otherwise -- tools/switch_matrix.sh runs this file under GNX_W64=2 too"""
    if "GNX_W64" not in os.environ:
        monkeypatch.setenv("GNX_W64", "0")


def _ragged(seed, count, nmax, mmax):
    rng = np.random.default_rng(seed)
    alphas, betas = [], []
    for k in range(count):
        n = int(rng.integers(1, nmax + 1))
        m = int(rng.integers(1, mmax + 1))
        if k % 3 == 0:  # a read inside a longer window
            b = rng.integers(0, 5, size=m).astype(np.uint8)
            lo = int(rng.integers(0, max(m - n, 0) + 1))
            a = common.mutate(rng, b[lo:lo + n], sub=0.06, indel=0.08, geo=0.5, alphabet=5)[:n]
        elif k % 3 == 1:  # related, similar length
            a = rng.integers(0, 4, size=n).astype(np.uint8)
            b = common.mutate(rng, a, sub=0.05, indel=0.05, geo=0.4)[:mmax]
        else:
            a = rng.integers(0, 5, size=n).astype(np.uint8)
            b = rng.integers(0, 5, size=m).astype(np.uint8)
        alphas.append(a)
        betas.append(b)
    return alphas, betas


@pytest.mark.parametrize("cs", [2, 7, 16, 300, 10000])
@pytest.mark.parametrize("mode", [1, 4])
def test_const_long_fuzz(gpu_lib, monkeypatch, mode, cs):
    """every shape through the snapshot path (GNX_CLONG=2 forces it for single-strip pairs too): ragged batches, several strips,
    several snapshot intervals, checkerboards of every size (quirk Q2), ConstGap and ConstGap_highMem"""
    monkeypatch.setenv("GNX_CLONG", "2")
    for seed, nmax, mmax, count in ((11, 40, 60, 96), (12, 700, 1500, 48), (13, 400, 2600, 24)):
        alphas, betas = _ragged(seed + 100 * cs, count, nmax, mmax)
        for name, g in (("Default", -430), ("HumanChimpTwo", -430), ("HoxD55", -100)):
            p = gpu_lib.make_params(mode, MX[name], g, 0, cs, cs)
            got = gpu_lib.align_batch(p, alphas, betas)
            common.expect_route(gpu_lib.get_timing(), 2)
            exp = oracle.align_batch(mode, MX[name], g, 0, alphas, betas, cs, cs, threads=8)
            common.assert_same(got, exp, "seed %d %s" % (seed, name))


@pytest.mark.parametrize("spec", ["0", "3", "4"])
@pytest.mark.parametrize("cs", [7, 10000])
def test_const_long_walk_speculation(gpu_lib, monkeypatch, spec, cs):
    """the walk with 0 / 3 / 4 tiles re-filled per round (cl_walk_kernel / cl_walk_spec_kernel: the wave's other lane groups re-fill the tiles
    a diagonal path enters next): near-diagonal ONT-like pairs (every guess right), ragged random pairs (most guesses wrong), both
    snapshot spacings, small checkerboards (quirk Q2) -- a wrong guess may cost a round, never a bit"""
    if os.environ.get("GNX_CLONG") == "0":
        pytest.skip("the snapshot path is switched off from outside")
    monkeypatch.setenv("GNX_CLONG", "2")
    monkeypatch.setenv("GNX_CL_WALK_SPEC", spec)
    rng = np.random.default_rng(77)
    alphas, betas = _ragged(31, 20, 1900, 5000)
    for _ in range(6):  # reads of 1500 .. 4000 bases inside windows of 6 .. 9 kb, 10 % errors
        m = int(rng.integers(6000, 9000)); n = int(rng.integers(1500, 4000))
        win = rng.integers(0, 4, size=m).astype(np.uint8)
        off = int(rng.integers(0, m - n - n // 8))
        alphas.append(common.mutate(rng, win[off:off + n + n // 8], sub=0.04, indel=0.06, geo=0.6)[:n]); betas.append(win)
    exp = oracle.align_batch(1, MX["HumanChimpTwo"], -430, 0, alphas, betas, cs, cs, threads=8)
    p = gpu_lib.make_params(gpu_lib.GNX_CONST_GAP, MX["HumanChimpTwo"], -430, 0, cs, cs)
    for ckc in ("224", "448"):
        monkeypatch.setenv("GNX_CL_CKC", ckc)
        got = gpu_lib.align_batch(p, alphas, betas)
        common.expect_route(gpu_lib.get_timing(), 2)
        common.assert_same(got, exp, "spec %s ckc %s" % (spec, ckc))


@pytest.mark.parametrize("ckc", ["", "224", "448"])
@pytest.mark.parametrize("nopipe", [False, True])
def test_const_long_strips_and_chunks(gpu_lib, monkeypatch, nopipe, ckc):
    """pipelined strips vs one wave per group of 4, both snapshot spacings (the sweep and the walk must agree on it: GNX_CL_CKC),
    and a workspace small enough to split the batch into several launches"""
    if os.environ.get("GNX_CLONG") == "0":
        pytest.skip("the snapshot path is switched off from outside: nothing to compare here")
    if nopipe:
        monkeypatch.setenv("GNX_NO_PIPE", "1")
    if ckc:
        monkeypatch.setenv("GNX_CL_CKC", ckc)
    alphas, betas = _ragged(21, 23, 2500, 6000)
    exp = oracle.align_batch(1, MX["HumanChimpTwo"], -430, 0, alphas, betas, 1000, 1000, threads=8)
    p = gpu_lib.make_params(gpu_lib.GNX_CONST_GAP, MX["HumanChimpTwo"], -430, 0, 1000, 1000)
    got = gpu_lib.align_batch(p, alphas, betas)
    common.expect_route(gpu_lib.get_timing(), 2)
    common.assert_same(got, exp)
    gpu_lib.check(gpu_lib.lib().gnx_init(0, 1 << 20))
    try:
        got = gpu_lib.align_batch(p, alphas, betas)
        assert gpu_lib.get_timing()["n_launches"] > 1
    finally:
        gpu_lib.check(gpu_lib.lib().gnx_init(0, 8 << 30))
    common.assert_same(got, exp)


def ont_pair(rng, n=20000, m=100000):
    """config C5 generator: a 20 kb ONT-like read (4 % substitutions, 3 % insertions, 3 % deletions) of a slice of a
    100 kb window"""
    win = rng.integers(0, 4, size=m).astype(np.uint8)
    off = int(rng.integers(0, m - n))
    read = common.mutate(rng, win[off:off + n + n // 8], sub=0.04, indel=0.06, geo=0.6)[:n]
    return read, win


def rescore_const(a, b, ops, scores, g):
    """(rows consumed, columns consumed, score) of a run-length CIGAR under the constant-gap model"""
    sc = np.asarray(scores, dtype=np.int64)
    run = ops["run_length"].astype(np.int64)
    op = ops["op"]
    di = np.where(op != 1, run, 0)
    dj = np.where(op != 2, run, 0)
    i0 = np.concatenate([[0], np.cumsum(di)[:-1]])
    j0 = np.concatenate([[0], np.cumsum(dj)[:-1]])
    total = int(g) * int(run[op != 0].sum())
    mm = op == 0
    if mm.any():
        # expand the M runs: positions (i0 + k, j0 + k)
        ln = run[mm]
        idx = np.repeat(np.arange(ln.shape[0]), ln)
        k = np.arange(ln.sum()) - np.repeat(np.cumsum(ln) - ln, ln)
        ai = a[i0[mm][idx] + k]
        bj = b[j0[mm][idx] + k]
        total += int(sc[ai, bj].sum())
    return int(di.sum()), int(dj.sum()), total


def test_c5_full_size(gpu_lib):
    """C5 as specified: ConstGap, 10 000 x 10 000 checkerboards, 20 kb x 100 kb.  Three pairs against the oracle (7 s of CPU each),
    one of them built so that quirk Q2 fires; 256 more through size-independent properties."""
    rng = np.random.default_rng(55)
    sc = MX["HumanChimpTwo"]
    g = -430
    n, m = 20000, 100000
    reads, wins = [], []
    for _ in range(2):
        r, w = ont_pair(rng)
        reads.append(r); wins.append(w)
    # Q2 by construction: the window is N except for an exact copy of the read starting at column 10 001, so the walk runs down
    # the diagonal to (0, 10 000) -- a corner of a checkerboard that is not the origin -- and Step 4 appends nothing
    # (constGap.go:59-63): the reference's CIGAR is 20000M 70000I, 10 000 columns short
    r = rng.integers(0, 4, size=n).astype(np.uint8)
    w = np.full(m, 4, dtype=np.uint8)
    w[10000:10000 + n] = r
    reads.append(r); wins.append(w)
    exp = oracle.align_batch(oracle.MODE_CONST, sc, g, 0, reads, wins, 10000, 10000, threads=3)
    exp_hi = oracle.align_batch(oracle.MODE_CONST_HIGHMEM, sc, g, 0, reads[2:], wins[2:], threads=1)
    o2 = exp[1][int(exp[2][2]):int(exp[2][3])]
    assert [(int(x["run_length"]), int(x["op"])) for x in o2] == [(20000, 0), (70000, 1)]  # the oracle shows the quirk ...
    assert int(exp_hi[2][1]) == 3  # ... and ConstGap_highMem does not (10000I 20000M 70000I)
    # ---- the GPU batch: the 3 oracle pairs + 256 generated ones, one call ----
    for _ in range(256):
        r, w = ont_pair(rng)
        reads.append(r); wins.append(w)
    gpu_lib.check(gpu_lib.lib().gnx_init(0, 40 << 30))
    try:
        p = gpu_lib.make_params(gpu_lib.GNX_CONST_GAP, sc, g, 0, 10000, 10000)
        score, ops, off = gpu_lib.align_batch(p, reads, wins)
        tm = gpu_lib.get_timing()
        ph = gpu_lib.make_params(gpu_lib.GNX_CONST_GAP_HIGHMEM, sc, g, 0)
        got_hi = gpu_lib.align_batch(ph, reads[2:3], wins[2:3])
    finally:
        gpu_lib.check(gpu_lib.lib().gnx_init(0, 8 << 30))
    common.expect_route(tm, 2)
    assert common.OUTER_ROUTE_SWITCH or tm["n_launches"] == 1  # 259 pairs in ONE launch (a stored direction matrix would need 130 GB)
    k = int(exp[2][-1])
    assert np.array_equal(score[:3], exp[0]) and np.array_equal(off[:4], exp[2])
    assert np.array_equal(ops["run_length"][:k], exp[1]["run_length"]) and np.array_equal(ops["op"][:k], exp[1]["op"])
    common.assert_same(got_hi, exp_hi)
    # properties: the CIGAR consumes all of alpha and all of beta, and re-scoring it gives the returned score -- except where quirk
    # Q2 dropped a leading gap: then the walk's first M step sits on a column that is a multiple of 10 000
    q2 = 0
    for x in range(len(reads)):
        o = ops[int(off[x]):int(off[x + 1])]
        ri, rj, total = rescore_const(reads[x], wins[x], o, sc, g)
        assert ri == n
        if rj == m:
            assert total == int(score[x]), "pair %d" % x
        else:
            q2 += 1
            assert (m - rj) % 10000 == 0 and int(o[0]["op"]) == 0, "pair %d" % x
            # the dropped leading gap accounts for the difference
            ri2, rj2, total2 = rescore_const(reads[x], wins[x][m - rj:], o, sc, g)
            assert total2 + g * (m - rj) == int(score[x])
    assert q2 >= 1


@pytest.mark.parametrize("cs", [2, 7, 16, 300, 10000])
@pytest.mark.parametrize("mode", [0, 2])
def test_affine_long_fuzz(gpu_lib, monkeypatch, mode, cs):
    """the same scheme for the three-state recurrence (csrc/affine_long.hip.h): AffineGap with checkerboards of every size (quirks Q1 / Q2,
    also across strip borders and in the last column) and AffineGap_highMem, ragged batches, several strips and snapshot intervals"""
    monkeypatch.setenv("GNX_CLONG", "2")
    for seed, nmax, mmax, count in ((31, 40, 60, 96), (32, 700, 700, 48), (33, 400, 1300, 24)):
        alphas, betas = _ragged(seed + 100 * cs, count, nmax, mmax)
        for name, go, ge in (("Default", -400, -30), ("HumanChimpTwo", -600, -150), ("HoxD55", 0, -40)):
            p = gpu_lib.make_params(mode, MX[name], go, ge, cs, cs)
            got = gpu_lib.align_batch(p, alphas, betas)
            common.expect_route(gpu_lib.get_timing(), 2)
            exp = oracle.align_batch(mode, MX[name], go, ge, alphas, betas, cs, cs, threads=8)
            common.assert_same(got, exp, "seed %d %s" % (seed, name))


def test_affine_long_big_pairs(gpu_lib, monkeypatch):
    """reads longer than the fast path's 320 rows against a 10 kb window and one 10 kb x 10 kb pair (cmd/cigarToBed's shape), pipelined
    strips -- forced (few pairs are faster over the stored matrix); and the natural route: a workspace the stored matrices do not fit"""
    rng = np.random.default_rng(77)
    win = rng.integers(0, 4, size=10000).astype(np.uint8)
    alphas, betas = [], []
    for n in (400, 900, 1700, 2500):
        o = int(rng.integers(0, 10000 - n - 50))
        alphas.append(common.mutate(rng, win[o:o + n + 40], sub=0.02, indel=0.01, geo=0.5)[:n]); betas.append(win)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    exp = oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, alphas, betas, threads=4)
    a = rng.integers(0, 4, size=9700).astype(np.uint8)
    b = common.mutate(rng, a, 0.01, 0.003)[:10000]
    exp1 = oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, [a], [b])
    monkeypatch.setenv("GNX_CLONG", "2")
    got = gpu_lib.align_batch(p, alphas, betas)
    common.expect_route(gpu_lib.get_timing(), 2)
    common.assert_same(got, exp)
    got = gpu_lib.align_batch(p, [a], [b])
    common.expect_route(gpu_lib.get_timing(), 2)
    common.assert_same(got, exp1)
    monkeypatch.delenv("GNX_CLONG")
    got = gpu_lib.align_batch(p, [a], [b])
    common.expect_route(gpu_lib.get_timing(), 0)  # one pair: the stored matrix (73 MB) and the wave-cooperative walk
    common.assert_same(got, exp1)
    if "GNX_NO_HFORM" in os.environ or os.environ.get("GNX_CLONG") == "0":
        return  # (the snapshot path is switched off from outside: a 32 MB workspace is then, rightly, GNX_ENOMEM)
    gpu_lib.check(gpu_lib.lib().gnx_init(0, 32 << 20))
    try:
        got = gpu_lib.align_batch(p, [a], [b])
        common.expect_route(gpu_lib.get_timing(), 2)  # ... unless it does not fit the workspace
    finally:
        gpu_lib.check(gpu_lib.lib().gnx_init(0, 8 << 30))
    common.assert_same(got, exp1)
