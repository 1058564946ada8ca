"""Alignments beyond the static int32 range of the kernels' keys.  The reference is int64 end to end
(align/align.go:8; align/affineGap.go:151-207, align/constGap.go:129-176) and its low-memory checkerboard exists for sequences far
longer than 4 * score fits int32 for; here such pairs run on int32 keys relative to a base every strip moves along (REBASE,
csrc/const_long.hip.h).  Checked: the REBASE kernels against the oracle on every shape of the snapshot suites (forced with GNX_REBASE=1),
pairs that really leave the static range (scores x 40 at 10 kb x 10 kb: keys up to 2.6e9) against the oracle, pairs the old bound
refused (n + m >= 178 955) against the oracle, and 1 Mb x 1 Mb / 300 kb x 2 Mb pairs through what the CIGAR must satisfy: it consumes
both sequences and re-scores, in int64, to the score returned."""
import os

import numpy as np
import pytest

import common
import oracle
from test_const_long import _ragged, rescore_const

pytestmark = pytest.mark.gpu
MX = common.matrices()


def rescore_affine(a, b, ops, scores, go, ge):
    """(rows consumed, columns consumed, score) of a run-length CIGAR under the affine model (a gap run of length L costs go + L * ge)"""
    sc = np.asarray(scores, dtype=np.int64)
    run = ops["run_length"].astype(np.int64)
    op = ops["op"]
    di = np.where(op != 1, run, 0)
    dj = np.where(op != 2, run, 0)
    i0 = np.concatenate([[0], np.cumsum(di)[:-1]])
    j0 = np.concatenate([[0], np.cumsum(dj)[:-1]])
    gaps = op != 0
    total = int(go) * int(gaps.sum()) + int(ge) * int(run[gaps].sum())
    mm = op == 0
    if mm.any():
        ln = run[mm]
        idx = np.repeat(np.arange(ln.shape[0]), ln)
        k = np.arange(ln.sum()) - np.repeat(np.cumsum(ln) - ln, ln)
        total += int(sc[a[i0[mm][idx] + k], b[j0[mm][idx] + k]].sum())
    return int(di.sum()), int(dj.sum()), total


def _related(rng, n, m_extra=0, sub=0.03, indel=0.01):
    a = rng.integers(0, 4, size=n).astype(np.uint8)
    b = common.mutate(rng, a, sub=sub, indel=indel, geo=0.4)
    if m_extra:
        b = np.concatenate([b, rng.integers(0, 4, size=m_extra).astype(np.uint8)])
    return a, b


@pytest.mark.parametrize("cs", [3, 16, 10000])
@pytest.mark.parametrize("mode", [0, 1, 2, 4])  # AffineGap, ConstGap, AffineGap_highMem, ConstGap_highMem
def test_rebase_forced(gpu_lib, monkeypatch, mode, cs):
    """GNX_CLONG=2 + GNX_REBASE=1: every pair through the snapshot path on moving bases (a rebase every 128 / 224 / 448 steps), ragged
    batches of one to several strips, small checkerboards (quirks Q1 / Q2) -- bit-exact against the oracle"""
    monkeypatch.setenv("GNX_CLONG", "2")
    monkeypatch.setenv("GNX_REBASE", "1")
    monkeypatch.setenv("GNX_W64", "0")  # (the 16-lane REBASE kernels; the 64-lane ones: test_w64_*)
    affine = mode in (0, 2)
    for seed, nmax, mmax, count in ((21, 60, 400, 64), (22, 700, 1500, 40), (23, 400, 2600, 24)):
        alphas, betas = _ragged(seed + 100 * cs, count, nmax, mmax)
        for name, go, ge in (("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("HoxD55", 0, -70)) if affine else (("HumanChimpTwo", -430, 0), ("HoxD55", -100, 0)):
            p = gpu_lib.make_params(mode, MX[name], go, ge, cs, cs)
            for ckc in ("224", "448") if not affine else ("",):
                if ckc:
                    monkeypatch.setenv("GNX_CL_CKC", ckc)
                got = gpu_lib.align_batch(p, alphas, betas)
                common.expect_route(gpu_lib.get_timing(), 2)
                exp = oracle.align_batch(mode, MX[name], go, ge, alphas, betas, cs, cs, threads=8)
                common.assert_same(got, exp, "seed %d %s ckc %s" % (seed, name, ckc))


@pytest.mark.parametrize("piped", ["1", "0"])
def test_rebase_forced_long_strips(gpu_lib, monkeypatch, piped):
    """the same with pairs of many strips and many blocks (piped: the strips of a pair as separate workgroups, bases handed over through memory)"""
    monkeypatch.setenv("GNX_CLONG", "2")
    monkeypatch.setenv("GNX_REBASE", "1")
    monkeypatch.setenv("GNX_W64", "0")  # (the 16-lane REBASE kernels; the 64-lane ones: test_w64_*)
    if piped == "0":
        monkeypatch.setenv("GNX_NO_PIPE", "1")
    rng = np.random.default_rng(5)
    alphas, betas = [], []
    for n, extra in ((3000, 0), (2500, 1500), (1700, 4000), (5000, 100)):
        a, b = _related(rng, n, extra, sub=0.05, indel=0.03)
        alphas.append(a); betas.append(b)
    alphas.append(rng.integers(0, 4, size=2000).astype(np.uint8)); betas.append(rng.integers(0, 4, size=3000).astype(np.uint8))  # unrelated
    for mode, name, go, ge in ((0, "HumanChimpTwo", -600, -150), (1, "HumanChimpTwo", -430, 0)):
        for cs in (1000, 10000):
            p = gpu_lib.make_params(mode, MX[name], go, ge, cs, cs)
            got = gpu_lib.align_batch(p, alphas, betas)
            exp = oracle.align_batch(mode, MX[name], go, ge, alphas, betas, cs, cs, threads=8)
            common.assert_same(got, exp, "mode %d cs %d" % (mode, cs))


@pytest.mark.parametrize("mode", [0, 1])
def test_keys_beyond_int32(gpu_lib, mode):
    """pairs that REALLY leave the static range, no switch: scores x 40 at 10 kb x 10 kb -- 4 * score reaches 2.6e9 along the diagonal
    (the old bound refused them with GNX_ERANGE); the int32 profile does not fit int16 either.  Against the oracle (int64)."""
    rng = np.random.default_rng(8 + mode)
    sc = [[40 * int(v) for v in row] for row in MX["HumanChimpTwo"]]
    alphas, betas = [], []
    for n in (10000, 9000, 2000):
        a, b = _related(rng, n, 0)
        alphas.append(a); betas.append(b)
    go, ge = (-600 * 40, -150 * 40) if mode == 0 else (-430 * 40, 0)
    p = gpu_lib.make_params(mode, sc, go, ge, 10000, 10000)
    got = gpu_lib.align_batch(p, alphas, betas)
    common.expect_route(gpu_lib.get_timing(), 2)
    exp = oracle.align_batch(mode, sc, go, ge, alphas, betas, 10000, 10000, threads=3)
    common.assert_same(got, exp)
    # what the kernels hold is 4 * (score - gapExtend * (i + j)) (ConstGap: - gapPen * (i + j)): at (n, m) beyond the 2^29 the absolute keys end at
    assert 4 * (int(got[0][0]) - (ge if mode == 0 else go) * (alphas[0].shape[0] + betas[0].shape[0])) > (1 << 29)


def test_range_boundary(gpu_lib):
    """the largest pairs the static bound admits and the smallest it does not, both exact: AffineGap(HumanChimpTwo,
    -600, -150) admits n + m + 2 < 2^27 / 600 = 223 696 (the bound is on (n + m + 2) * the largest penalty)"""
    rng = np.random.default_rng(99)
    sc, go, ge = MX["HumanChimpTwo"], -600, -150
    lim = (1 << 27) // 600  # (n + m + 2) * 600 < 2^27
    for total, route in ((lim - 3, None), (lim + 40, 2)):
        n = 300
        m = total - n
        win = rng.integers(0, 4, size=m).astype(np.uint8)
        off = int(rng.integers(0, m - 400))
        a = common.mutate(rng, win[off:off + 340], sub=0.02, indel=0.01, geo=0.5)[:n]
        p = gpu_lib.make_params(0, sc, go, ge, 10000, 10000)
        got = gpu_lib.align_batch(p, [a], [win])
        if route is not None:
            common.expect_route(gpu_lib.get_timing(), route)
        exp = oracle.align_batch(0, sc, go, ge, [a], [win], 10000, 10000, threads=1)
        common.assert_same(got, exp, "n + m = %d" % total)


@pytest.mark.parametrize("cs", [2, 7, 10000])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_wide_forced(gpu_lib, monkeypatch, mode, cs):
    """GNX_WIDE=2: every batch through the int64 kernel (csrc/lat_wide.hip.h: literal three-candidate recurrences on int64 keys, the direction
    words of the latency geometry) -- every mode incl. AffineGapLocal, gapOpen > 0 and = 0, ragged batches, small checkerboards"""
    if mode in (2, 3, 4) and cs != 10000:
        pytest.skip("checkerboards are a parameter of the low-memory modes")
    monkeypatch.setenv("GNX_WIDE", "2")
    affine = mode in (0, 2, 3)
    for seed, nmax, mmax, count in ((31, 40, 60, 64), (32, 300, 700, 32), (33, 700, 1500, 12), (34, 1300, 200, 8)):
        alphas, betas = _ragged(seed + 100 * cs + mode, count, nmax, mmax)
        for name, go, ge in (("HumanChimpTwo", -600, -150), ("Default", 25, -30), ("HoxD55", 0, -70)) if affine else (("HumanChimpTwo", -430, 0), ("HoxD55", -100, 0), ("Default", 3, 0)):
            p = gpu_lib.make_params(mode, MX[name], go, ge, cs, cs)
            got = gpu_lib.align_batch(p, alphas, betas)
            assert gpu_lib.get_timing()["fast_path"] == 4
            exp = oracle.align_batch(mode, MX[name], go, ge, alphas, betas, cs, cs, threads=8)
            common.assert_same(got, exp, "seed %d %s" % (seed, name))


def test_beyond_int32_in_every_mode(gpu_lib):
    """no switch: what the old bound refused with GNX_ERANGE and moving bases cannot take either -- scores x 1000 at 3 kb x 3 kb (VERDICT r4 item 1 b:
    a strip's band spans more than int32), AffineGapLocal of a 230 kb target, gapOpen > 0 at 400 kb -- runs on the int64 kernel and equals the oracle"""
    rng = np.random.default_rng(3)
    big = [[1000 * int(v) for v in row] for row in MX["HumanChimpTwo"]]
    a3, b3 = _related(rng, 3000, 0)
    a3b, b3b = _related(rng, 2500, 700)
    for mode, go, ge in ((0, -600000, -150000), (1, -430000, 0), (2, -600000, -150000)):
        p = gpu_lib.make_params(mode, big, go, ge, 10000, 10000)
        got = gpu_lib.align_batch(p, [a3, a3b], [b3, b3b])
        assert gpu_lib.get_timing()["fast_path"] == 4
        common.assert_same(got, oracle.align_batch(mode, big, go, ge, [a3, a3b], [b3, b3b], 10000, 10000, threads=2), "scores x 1000, mode %d" % mode)
    q = rng.integers(0, 4, size=200).astype(np.uint8)
    t = rng.integers(0, 4, size=230000).astype(np.uint8)
    t[120000:120200] = q
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_LOCAL, MX["HumanChimpTwo"], -600, -150)
    got = gpu_lib.align_batch(p, [t], [q])
    assert gpu_lib.get_timing()["fast_path"] == 4
    common.assert_same(got, oracle.align_batch(3, MX["HumanChimpTwo"], -600, -150, [t], [q], threads=1), "AffineGapLocal, 230 kb target")
    a, b = _related(rng, 300, 0)
    win = np.concatenate([rng.integers(0, 4, size=250000).astype(np.uint8), b, rng.integers(0, 4, size=150000).astype(np.uint8)])
    p = gpu_lib.make_params(0, MX["HumanChimpTwo"], 50, -150, 10000, 10000)
    got = gpu_lib.align_batch(p, [a], [win])
    assert gpu_lib.get_timing()["fast_path"] == 4
    common.assert_same(got, oracle.align_batch(0, MX["HumanChimpTwo"], 50, -150, [a], [win], 10000, 10000, threads=1), "gapOpen > 0")


@pytest.mark.parametrize("strips", ["2", "3", "7"])
@pytest.mark.parametrize("mode", [0, 1, 2, 4])
def test_row_panels_forced(gpu_lib, monkeypatch, mode, strips):
    """GNX_MEGA_STRIPS=k: the row-panel path (run_device_mega: forward sweep panel by panel, the bottom row of a panel handed to the next one with
    its bases; backward re-sweep of a panel up to the column the walk has reached, the walk resumed from what it carried out of the panel below)
    with panels of k strips -- pairs of 1 .. 14 panels, checkerboards small and large (quirks Q1 / Q2 across panel borders), against the oracle"""
    monkeypatch.setenv("GNX_MEGA_STRIPS", strips)
    affine = mode in (0, 2)
    rng = np.random.default_rng(40 + mode)
    alphas, betas = [], []
    for n, extra, sub, indel in ((2100, 0, 0.05, 0.03), (1500, 900, 0.05, 0.03), (700, 2500, 0.08, 0.05), (3300, 40, 0.02, 0.004), (150, 1200, 0.05, 0.02), (330, 700, 0.2, 0.1)):
        a, b = _related(rng, n, extra, sub=sub, indel=indel)
        alphas.append(a); betas.append(b)
    alphas.append(rng.integers(0, 4, size=1000).astype(np.uint8)); betas.append(rng.integers(0, 4, size=800).astype(np.uint8))  # unrelated
    w = rng.integers(0, 4, size=3000).astype(np.uint8)
    alphas.append(common.mutate(rng, w[1200:2300], sub=0.04, indel=0.03, geo=0.5)); betas.append(w)  # a read inside a window: long leading / trailing gaps
    for cs in ((10000, 1000, 7) if mode in (0, 1) else (10000,)):
        for name, go, ge in ((("HumanChimpTwo", -600, -150), ("HoxD55", 0, -70)) if affine else (("HumanChimpTwo", -430, 0),)):
            p = gpu_lib.make_params(mode, MX[name], go, ge, cs, cs)
            got = gpu_lib.align_batch(p, alphas, betas)
            assert gpu_lib.get_timing()["fast_path"] == 5
            exp = oracle.align_batch(mode, MX[name], go, ge, alphas, betas, cs, cs, threads=8)
            common.assert_same(got, exp, "mode %d cs %d %s" % (mode, cs, name))


@pytest.mark.parametrize("cs", [3, 16, 10000])
@pytest.mark.parametrize("walk", ["farm", "farm1", "farm3", "farm32", "farm:seq", "farm3:seq", "farm:ck128", "farm3:ck256", "farm:r6", "farm:r8", "farm:r16", "farm3:r6:seq", "farm1:r16:ck128", "farm32:r8:ck256", "two_waves", "one_wave"])  # the walk farm (farm64.hip.h: 24 (the default) / 1 / 3 / 32 tiles per round, overlapped rounds or {fill, walk} launches), al64_walk2_kernel / cl64_walk2_kernel (the left neighbour tile re-filled by a second wave) or the one-wave kernels
@pytest.mark.parametrize("mode", [0, 1, 2, 4])  # AffineGap, ConstGap, AffineGap_highMem, ConstGap_highMem
def test_w64_forced(gpu_lib, monkeypatch, mode, cs, walk):
    """GNX_CLONG=2 + GNX_W64=2: every pair through the snapshot path with the whole wave on one pair (affine_long64.hip.h / const_long64.hip.h:
    64 lanes x 10 rows, strips of 640 rows, wave_shr moves, moving bases) -- ragged batches of one to eight strips, small checkerboards, against the oracle"""
    monkeypatch.setenv("GNX_CLONG", "2")
    monkeypatch.setenv("GNX_W64", "2")
    affine = mode in (0, 2)
    monkeypatch.delenv("GNX_W64_FARM", raising=False)
    monkeypatch.delenv("GNX_W64_FARM_PIPE", raising=False)
    monkeypatch.delenv("GNX_W64_CK", raising=False)
    monkeypatch.setenv("GNX_W64_R", "10")  # rows per lane of the affine sweep + farm (round 6: 6 / 8 / 10 / 16, strips of 384 .. 1 024 rows; w64_pick_rows chooses by the strip count)
    if walk.endswith(":seq"):
        monkeypatch.setenv("GNX_W64_FARM_PIPE", "0")
        walk = walk[:-4]
    monkeypatch.setenv("GNX_W64_RC", "4" if ":r" in walk else "10")  # ... of the constant-gap sweep + farm (4 / 10: strips of 256 / 640 rows)
    if ":r" in walk:
        monkeypatch.setenv("GNX_W64_R", walk.split(":r")[1].split(":")[0])
        walk = walk.replace(":r" + walk.split(":r")[1].split(":")[0], "")
    if ":ck" in walk:  # snapshot spacing of the affine sweep under the farm (default 512 steps)
        monkeypatch.setenv("GNX_W64_CK", walk.split(":ck")[1])
        walk = walk.split(":ck")[0]
    if walk.startswith("farm") and walk != "farm":
        monkeypatch.setenv("GNX_W64_FARM", walk[4:])
    if walk in ("two_waves", "one_wave"):
        monkeypatch.setenv("GNX_W64_FARM", "0")
    if walk == "one_wave":
        monkeypatch.setenv("GNX_W64_SPEC", "0")
    for seed, nmax, mmax, count in ((31, 60, 400, 48), (32, 2000, 1500, 30), (33, 5000, 2600, 12), (34, 700, 9000, 10)):
        alphas, betas = _ragged(seed + 100 * cs, count, nmax, mmax)
        for name, go, ge in (("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("HoxD55", 0, -70)) if affine else (("HumanChimpTwo", -430, 0), ("HoxD55", -100, 0)):
            p = gpu_lib.make_params(mode, MX[name], go, ge, cs, cs)
            got = gpu_lib.align_batch(p, alphas, betas)
            assert gpu_lib.get_timing()["fast_path"] == 6
            exp = oracle.align_batch(mode, MX[name], go, ge, alphas, betas, cs, cs, threads=8)
            common.assert_same(got, exp, "seed %d %s" % (seed, name))


@pytest.mark.parametrize("mode", [0, 1])
def test_w64_is_the_route_of_few_pairs(gpu_lib, monkeypatch, mode):
    """the shipped rule: launches of up to three pairs of the snapshot path, and of up to 64 pairs of two or more 640-row strips each, run in the
    64-lane geometry with the walk farm (route 6); a batch of more than three with a shorter pair in it fills the lane groups of al_sweep_kernel's
    waves (route 2); related pairs of many strips and columns, a scaled matrix that leaves int16 (P16 off)"""
    monkeypatch.setenv("GNX_CLONG", "2")
    monkeypatch.delenv("GNX_W64", raising=False)
    monkeypatch.delenv("GNX_W64_FARM", raising=False)
    rng = np.random.default_rng(77)
    pairs = [_related(rng, n, extra, sub=0.05, indel=0.03) for n, extra in ((6000, 0), (2500, 5000), (1300, 100), (4000, 300), (700, 900))]
    for npairs, route in ((1, 6), (3, 6), (4, 6), (5, 2)):
        alphas, betas = [x[0] for x in pairs[:npairs]], [x[1] for x in pairs[:npairs]]
        x25 = [[25 * int(v) for v in row] for row in MX["HumanChimpTwo"]]
        for sc, go, ge in ((MX["HumanChimpTwo"], -600, -150), (x25, -15000, -3750)) if mode == 0 else ((MX["HumanChimpTwo"], -430, 0), (x25, -10750, 0)):
            for cs in (1000, 10000):
                p = gpu_lib.make_params(mode, sc, go, ge, cs, cs)
                got = gpu_lib.align_batch(p, alphas, betas)
                if not common.OUTER_ROUTE_SWITCH:
                    assert gpu_lib.get_timing()["fast_path"] == route, (npairs, go, cs)
                exp = oracle.align_batch(mode, sc, go, ge, alphas, betas, cs, cs, threads=4)
                common.assert_same(got, exp, "%d pairs cs %d" % (npairs, cs))


@pytest.mark.parametrize("strips", ["2", "3", "3:farm2", "2:farm_seq", "2:farm_ck128", "2:farm_r6", "3:farm_r8", "2:farm_r16", "2:two_waves", "2:one_wave"])
@pytest.mark.parametrize("mode", [0, 1, 2, 4])
def test_w64_row_panels(gpu_lib, monkeypatch, mode, strips):
    """row panels (run_device_mega) of k strips of 640 rows in the 64-lane geometry: the stand-in strip, MegaState across panel borders;
    the walk as a farm (default; 2 tiles per round) or on one workgroup of two waves / one wave"""
    monkeypatch.delenv("GNX_W64_FARM", raising=False)
    monkeypatch.delenv("GNX_W64_FARM_PIPE", raising=False)
    monkeypatch.delenv("GNX_W64_CK", raising=False)
    monkeypatch.setenv("GNX_W64_R", strips.split(":farm_r")[1] if ":farm_r" in strips else "10")  # rows per lane of the affine sweep + farm
    monkeypatch.setenv("GNX_W64_RC", "4" if ":farm_r" in strips else "10")  # ... of the constant-gap one
    monkeypatch.delenv("GNX_W64_CK_MEGA", raising=False)  # (default: 2 048 steps between the snapshots of a row panel's strips)
    if strips.endswith(":farm_ck128"):
        monkeypatch.setenv("GNX_W64_CK", "128")
        monkeypatch.setenv("GNX_W64_CK_MEGA", "128")
    if strips.endswith(":farm_r8"):
        monkeypatch.setenv("GNX_W64_CK_MEGA", "1024")
    if strips.endswith(":farm_seq"):
        monkeypatch.setenv("GNX_W64_FARM_PIPE", "0")
    if strips.endswith(":farm2"):
        monkeypatch.setenv("GNX_W64_FARM", "2")
    if strips.endswith(":one_wave") or strips.endswith(":two_waves"):
        monkeypatch.setenv("GNX_W64_FARM", "0")
    if strips.endswith(":one_wave"):
        monkeypatch.setenv("GNX_W64_SPEC", "0")
    monkeypatch.setenv("GNX_MEGA_STRIPS", strips.split(":")[0])
    monkeypatch.setenv("GNX_W64", "2")
    rng = np.random.default_rng(60 + mode)
    alphas, betas = [], []
    for n, extra, sub, indel in ((6500, 0, 0.05, 0.03), (2700, 1900, 0.05, 0.03), (1400, 3500, 0.08, 0.05), (9000, 40, 0.02, 0.004), (150, 1200, 0.05, 0.02), (700, 700, 0.2, 0.1)):
        a, b = _related(rng, n, extra, sub=sub, indel=indel)
        alphas.append(a); betas.append(b)
    w = rng.integers(0, 4, size=6000).astype(np.uint8)
    alphas.append(common.mutate(rng, w[1200:4300], sub=0.04, indel=0.03, geo=0.5)); betas.append(w)  # a read inside a window: long leading / trailing gaps
    affine = mode in (0, 2)
    for cs in ((10000, 1000, 7) if mode in (0, 1) else (10000,)):
        for name, go, ge in ((("HumanChimpTwo", -600, -150), ("HoxD55", 0, -70)) if affine else (("HumanChimpTwo", -430, 0),)):
            p = gpu_lib.make_params(mode, MX[name], go, ge, cs, cs)
            got = gpu_lib.align_batch(p, alphas, betas)
            assert gpu_lib.get_timing()["fast_path"] == 5
            exp = oracle.align_batch(mode, MX[name], go, ge, alphas, betas, cs, cs, threads=8)
            common.assert_same(got, exp, "mode %d cs %d %s" % (mode, cs, name))


def _long_pairs():
    import importlib.util
    spec = importlib.util.spec_from_file_location("long_pairs", os.path.join(common.HERE, "..", "tools", "long_pairs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("case", ["const_150k", "affine_340k", "affine_q1_300k", "affine_1M"])
def test_long_pairs_equal_the_oracle(gpu_lib, case):
    """the callers' own parameters at lengths where the keys leave int32 (ConstGap 150 000 x 180 009: 4 (score - g (n + m)) = 5.7e8;
    AffineGap 340 000 x 339 906: 5.3e8), a 300 kb pair built so that quirk Q1 fires at EVERY 10 000-row checkerboard edge (beta lacks the bases of
    alpha around each row 10 000 k: the walk crosses the edge inside a D run, align/affineGap.go:305), and the megabase regime itself, AffineGap
    1 Mb x 1 Mb (1e12 cells): score, number of runs and sha256 of the CIGAR equal what the CPU oracle produced in 185 s / 895 s / ~700 s / ~2 h
    of one core (tests/golden/long_pairs.json, written by `python tools/long_pairs.py oracle`; the pairs are seeded, tools/long_pairs.py gen)"""
    import json
    lp = _long_pairs()
    with open(lp.FIXTURE) as fh:
        fx = json.load(fh)[case]
    affine, a, b = lp.gen(case)
    assert (a.shape[0], b.shape[0]) == (fx["n"], fx["m"])
    sc, go, ge = lp.params(affine)
    p = gpu_lib.make_params(0 if affine else 1, sc, go, ge, 10000, 10000)
    gpu_lib.check(gpu_lib.lib().gnx_init(0, 0))
    gpu_lib.debug_counter(3, reset=True)
    score, ops, off = gpu_lib.align_batch(p, [a], [b])
    common.expect_route(gpu_lib.get_timing(), 2)
    assert lp.digest(score[0], ops) == {k: fx[k] for k in ("score", "runs", "sha256")}
    if affine and gpu_lib.get_timing()["fast_path"] in (5, 6):  # what the walk reports about quirk Q1 bounds the CIGAR's re-score (the check of the pairs no oracle finishes)
        changed = gpu_lib.debug_counter(3, reset=False)
        ni, nj, total = rescore_affine(a, b, ops, sc, go, ge)
        assert (ni, nj) == (a.shape[0], b.shape[0]) and 0 <= int(score[0]) - total <= -go * changed
        if case == "affine_q1_300k":
            assert changed >= 3 and total < int(score[0])  # the pair does what it was built for


def _megabase_check(gpu_lib, lp, case, route):
    affine, a, b = lp.gen(case)
    sc, go, ge = lp.params(affine)
    gpu_lib.check(gpu_lib.lib().gnx_init(0, 0))
    # highMem semantics: the CIGAR is an optimal path -- it consumes both sequences and re-scores to the score exactly
    ph = gpu_lib.make_params(2 if affine else 4, sc, go, ge)
    score_h, ops_h, _ = gpu_lib.align_batch(ph, [a], [b])
    ni, nj, total = rescore_affine(a, b, ops_h, sc, go, ge) if affine else rescore_const(a, b, ops_h, sc, go)
    common.expect_route(gpu_lib.get_timing(), route)
    assert (ni, nj) == (a.shape[0], b.shape[0]) and total == int(score_h[0])
    # the callers' function (10 000 x 10 000 checkerboards): the same score; its CIGAR may carry the reference's quirk Q1 -- where the walk leaves a
    # checkerboard upwards it restarts in the argmax state X of the entry cell (align/affineGap.go:305); when that is not the traced gap state the part of
    # the gap below the edge is paid as a new gap: a deficit of gapOpen - (X - D) in [0, gapOpen] per such restart.  The walk counts them
    # (gnx_debug_counter(3)): the CIGAR re-scores to within gapOpen x that count below the score -- exactly to it when there was none (ConstGap has no state: always)
    p = gpu_lib.make_params(0 if affine else 1, sc, go, ge, 10000, 10000)
    gpu_lib.debug_counter(3, reset=True)
    score, ops, off = gpu_lib.align_batch(p, [a], [b])
    changed = gpu_lib.debug_counter(3, reset=False) if affine else 0
    ni, nj, total = rescore_affine(a, b, ops, sc, go, ge) if affine else rescore_const(a, b, ops, sc, go)
    assert int(score[0]) == int(score_h[0])
    assert (ni, nj) == (a.shape[0], b.shape[0]) and 0 <= int(score[0]) - total <= (-go if affine else 0) * changed, (int(score[0]) - total, changed)


@pytest.mark.parametrize("case", ["const_300k_2M"])
def test_megabase_pairs(gpu_lib, case):
    """one 300 kb x 2 Mb ConstGap pair (6e11 cells; 450 000 runs), the callers' parameters and 10 000 x 10 000 checkerboards (cmd/globalAlignment/globalAlignment.go:84):
    no oracle finishes it -- the CIGAR consumes both sequences and re-scores in int64 to the returned score.  (AffineGap 1 Mb x 1 Mb has an oracle digest since
    round 6: test_long_pairs_equal_the_oracle.)"""
    _megabase_check(gpu_lib, _long_pairs(), case, 2)


def test_row_panels_at_their_natural_size(gpu_lib, monkeypatch):
    """AffineGap 2 Mb x 2 Mb (4e12 cells) with a snapshot every 128 steps: 350 GB of bottom rows + snapshots -- more than the device has, so run_device_mega
    cuts the pair into row panels of the size IT chooses (no GNX_MEGA_STRIPS; ~2 s of kernels): forward panels, backward panels re-swept with snapshots, the walk
    farm crossing panel borders with MegaState (.MISSING_LARGE_BLOBS:1-3: the reference ships a 5 Mb x 5 Mb fixture for cmd/cigarToBed)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    fr, tot = ctypes.c_size_t(), ctypes.c_size_t()
    gpu_lib.check(gpu_lib.lib().gnx_init(0, 0))
    assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
    free = fr.value + 0  # (plus what the library's context already holds: it re-uses it)
    if free < 150 * 2 ** 30:
        pytest.skip("needs a device with 150 GB free")
    monkeypatch.setenv("GNX_W64_CK", "128")       # (the one-launch path: 350 GB -> row panels)
    monkeypatch.setenv("GNX_W64_CK_MEGA", "128")  # (and the panels keep that spacing: several backward panels instead of one)
    _megabase_check(gpu_lib, _long_pairs(), "affine_2M", 5)
