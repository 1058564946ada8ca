"""The latency geometry of the stored-matrix path (csrc/lat_fill.hip.h: one pair per wave, 64 lanes x 2 rows, the strips of a pair as
piped workgroups that hand rows over through sentinel-marked memory) against the CPU oracle, bit-exact: every mode it serves
(align/affineGap.go:59-344, align/constGap.go:13-311, the highMem twins, AffineGapLocal), ragged batches, one to many strips, small
checkerboards (quirks Q1 / Q2), the golden pairs of cmd/cigarToBed and cmd/globalAlignmentAnchor, and the routing: a single
align.AffineGap / ConstGap call takes it by default."""
import os

import numpy as np
import pytest

import common
import oracle
from test_const_long import _ragged
from gonomics_amd import align

pytestmark = pytest.mark.gpu
MX = common.matrices()


@pytest.mark.parametrize("cs", [2, 7, 300, 10000])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_lat_fuzz(gpu_lib, monkeypatch, mode, cs):
    """GNX_LAT=2: every batch through lat_fill_kernel + traceback_kernel<.., 64, 2>, whatever its size"""
    if mode in (2, 3, 4) and cs != 10000:
        pytest.skip("checkerboards are a parameter of the low-memory modes")
    monkeypatch.setenv("GNX_LAT", "2")
    affine = mode in (0, 2, 3)
    for seed, nmax, mmax, count in ((11, 40, 60, 64), (12, 300, 700, 32), (13, 700, 1500, 16), (14, 1300, 200, 12)):
        alphas, betas = _ragged(seed + 100 * cs + mode, count, nmax, mmax)
        for name, go, ge in (("HumanChimpTwo", -600, -150), ("Default", -400, -30), ("HoxD55", 0, -70)) if affine else (("HumanChimpTwo", -430, 0), ("HoxD55", -100, 0)):
            p = gpu_lib.make_params(mode, MX[name], go, ge, cs, cs)
            got = gpu_lib.align_batch(p, alphas, betas)
            assert gpu_lib.get_timing()["fast_path"] == 3
            exp = oracle.align_batch(mode, MX[name], go, ge, alphas, betas, cs, cs, threads=8)
            common.assert_same(got, exp, "seed %d %s" % (seed, name))


def test_lat_single_pairs_by_default(gpu_lib):
    """no switch: one pair per call -- the shapes of the named commands -- takes the latency geometry; 1 kb x 1 kb, 150 x 10 kb, 3 kb x 5 kb (24
    strips), the 9 673 x 10 000 pair of cmd/cigarToBed's test, both functions"""
    rng = np.random.default_rng(17)
    shapes = [(1000, 1000), (150, 10000), (3000, 5000), (1, 1), (129, 64), (128, 2000)]
    pairs = []
    for n, m in shapes:
        a = rng.integers(0, 4, size=n).astype(np.uint8)
        b = common.mutate(rng, a, sub=0.05, indel=0.03, geo=0.4)
        b = np.concatenate([b, rng.integers(0, 4, size=max(m - len(b), 0)).astype(np.uint8)])[:max(m, 1)]
        pairs.append((a, b))
    for mode, go, ge in ((0, -600, -150), (1, -430, 0)):
        p = gpu_lib.make_params(mode, MX["HumanChimpTwo"], go, ge, 10000, 10000)
        for a, b in pairs:
            got = gpu_lib.align_batch(p, [a], [b])
            tm = gpu_lib.get_timing()
            if not common.OUTER_ROUTE_SWITCH and "GNX_LAT" not in os.environ and os.environ.get("GNX_FP_SMALL") != "1":
                assert tm["fast_path"] == 3, (a.shape, b.shape, tm["fast_path"])
            exp = oracle.align_batch(mode, MX["HumanChimpTwo"], go, ge, [a], [b], 10000, 10000, threads=1)
            common.assert_same(got, exp, "%d x %d mode %d" % (a.shape[0], b.shape[0], mode))


def test_lat_golden_anchor_pairs(gpu_lib, monkeypatch):
    """the reference-held vectors of cmd/globalAlignmentAnchor (score + full CIGAR, incl. 660 x 1265 with 114 runs) through the latency geometry"""
    monkeypatch.setenv("GNX_LAT", "2")
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, align.HumanChimpTwoScoreMatrix, -600, -150)
    for idx in (1, 2):
        for a, b, score, cig in common.anchor_cases(idx):
            s, ops, off = gpu_lib.align_batch(p, [a], [b])
            assert gpu_lib.get_timing()["fast_path"] == 3
            assert int(s[0]) == score
            assert common.fmt_v([(int(r), int(o)) for r, o in zip(ops["run_length"], ops["op"])]) == cig


def test_lat_many_strips_and_stolen_items(gpu_lib, monkeypatch):
    """a tall pair (40 strips) and a batch of them; then the same with every other workgroup of a chain asleep under a 50 us grace period, so
    that strips are run by their successors (claim_items): same bits"""
    monkeypatch.setenv("GNX_LAT", "2")
    rng = np.random.default_rng(23)
    alphas, betas = [], []
    for n, m in ((5000, 900), (2600, 2600), (640, 3000), (1281, 77)):
        a = rng.integers(0, 4, size=n).astype(np.uint8)
        b = common.mutate(rng, a, sub=0.06, indel=0.04, geo=0.5)[:m]
        alphas.append(a); betas.append(b)
    for mode, go, ge, cs in ((0, -600, -150, 10000), (0, -400, -30, 500), (1, -430, 0, 10000), (3, -600, -150, 10000)):
        p = gpu_lib.make_params(mode, MX["HumanChimpTwo"], go, ge, cs, cs)
        exp = oracle.align_batch(mode, MX["HumanChimpTwo"], go, ge, alphas, betas, cs, cs, threads=4)
        common.assert_same(gpu_lib.align_batch(p, alphas, betas), exp, "mode %d" % mode)
        monkeypatch.setenv("GNX_TICKET_DELAY", "600")   # ~2 ms before the sleepers claim ...
        monkeypatch.setenv("GNX_CLAIM_GRACE_US", "50")  # ... a successor takes an unclaimed predecessor after 50 us
        gpu_lib.debug_counter(0, reset=True)
        common.assert_same(gpu_lib.align_batch(p, alphas, betas), exp, "mode %d, sleeping workgroups" % mode)
        assert gpu_lib.debug_counter(0) > 0  # items really were run by a successor
        monkeypatch.delenv("GNX_TICKET_DELAY"); monkeypatch.delenv("GNX_CLAIM_GRACE_US")
