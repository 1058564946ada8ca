"""Forward progress of the pipelined launches must not depend on the order in which workgroups start (VERDICT r2 item 3;
MI355X_MICROARCH "Workgroup dispatch": HIP promises nothing about dispatch order).  In every piped kernel a workgroup CLAIMS its item
and every unclaimed predecessor of it, and runs those first (csrc/gnx_common.hip.h: claim_items): nobody waits for work that has not been
taken.  Two stress legs, each over all piped kernels, results against the oracle:
  * GNX_TICKET_DELAY: the workgroups of the lower half of the grid sleep before they claim, so the upper half finds its predecessors
    unclaimed and runs whole chains itself (what a dispatcher that starts workgroups in another order would cause);
  * gnx_debug_occupy: 128 workgroups that each hold a whole CU's LDS spin on another stream while the piped launch runs, so half
    the CUs cannot take workgroups of the launch at all.
  * steal: the delay alone never makes anybody TAKE an item -- the default grace period (20 ms) outlasts it.  This leg
    sets GNX_CLAIM_GRACE_US=50 for the launch together with a delay of ~2 ms, so the upper half of the grid really claims and runs its
    predecessors' items (the s_lo .. s_own loops of the strip kernels, the level loop of fp_sweep_levels_kernel), and asserts through
    gnx_debug_counter(0) that at least one item was run by a workgroup other than its own.
The bug trap (5 s spin timeout -> error flag 16 -> sequential re-run) must stay silent: the legs assert that the piped launch was
the one that produced the result by checking that the call took far less than the timeout."""
import os
import time

import numpy as np
import pytest

import common
import oracle

pytestmark = pytest.mark.gpu
MX = common.matrices()


def _pairs(seed, n_pairs, n_lo, n_hi, m_lo, m_hi):
    rng = np.random.default_rng(seed)
    alphas, betas = [], []
    for _ in range(n_pairs):
        m = int(rng.integers(m_lo, m_hi + 1))
        b = rng.integers(0, 4, size=m).astype(np.uint8)
        n = int(rng.integers(n_lo, n_hi + 1))
        o = int(rng.integers(0, max(1, m - n)))
        a = common.mutate(rng, b[o:o + n], sub=0.05, indel=0.02, geo=0.4, alphabet=4)
        if len(a) == 0:
            a = b[:1].copy()
        alphas.append(a); betas.append(b)
    return alphas, betas


# (name, environment that routes the batch to the kernel, mode, matrix, penalties, pairs)
LEGS = [
    ("fill_affine_kernel<MULTI> piped strips", {"GNX_FASTPATH": "0", "GNX_LAT": "0"}, 0, "HumanChimpTwo", (-600, -150), (101, 6, 900, 1400, 2200, 2600)),
    ("fill_const_kernel<MULTI> piped strips", {"GNX_CLONG": "0", "GNX_LAT": "0"}, 1, "Default", (-430, 0), (102, 6, 900, 1400, 2200, 2600)),
    ("cl_sweep_wg_kernel piped items of 5 strips", {"GNX_CLONG": "2", "GNX_W64": "0"}, 1, "HumanChimpTwo", (-430, 0), (103, 6, 1700, 2500, 2200, 2600)),
    ("cl_sweep_kernel piped strips", {"GNX_CLONG": "2", "GNX_CL_WG": "0", "GNX_W64": "0"}, 1, "HumanChimpTwo", (-430, 0), (103, 6, 900, 1400, 2200, 2600)),
    ("al_sweep_kernel piped strips", {"GNX_CLONG": "2", "GNX_FASTPATH": "0", "GNX_W64": "0"}, 0, "HumanChimpTwo", (-600, -150), (104, 6, 900, 1400, 2200, 2600)),
    ("fp_sweep_levels_kernel (row blocks)", {"GNX_FASTPATH": "2"}, 0, "HumanChimpTwo", (-600, -150), (105, 96, 500, 800, 1500, 1800)),
    # round 5: the latency geometry's strips (one wave per strip, rows handed over through sentinel-marked memory) and its int64 form
    ("lat_fill_kernel<affine> piped strips", {"GNX_LAT": "2"}, 0, "HumanChimpTwo", (-600, -150), (106, 6, 900, 1400, 2200, 2600)),
    ("lat_fill_kernel<const> piped strips", {"GNX_LAT": "2"}, 1, "Default", (-430, 0), (107, 6, 900, 1400, 2200, 2600)),
    ("lat_wide_kernel piped strips", {"GNX_WIDE": "2"}, 0, "HumanChimpTwo", (-600, -150), (108, 6, 900, 1400, 2200, 2600)),
    # the snapshot path with the whole wave on one pair (strips of 640 rows)
    ("al64_sweep_kernel piped strips", {"GNX_CLONG": "2", "GNX_FASTPATH": "0", "GNX_W64": "2"}, 0, "HumanChimpTwo", (-600, -150), (109, 5, 2000, 3800, 3000, 4200)),
    ("cl64_sweep_kernel piped strips", {"GNX_CLONG": "2", "GNX_W64": "2"}, 1, "HumanChimpTwo", (-430, 0), (110, 5, 2000, 3800, 3000, 4200)),
]


@pytest.mark.parametrize("leg", LEGS, ids=[x[0].split()[0].replace("<", "_").replace(">", "") for x in LEGS])
@pytest.mark.parametrize("stress", ["delay", "occupy", "both", "steal"])
def test_piped_launches_do_not_depend_on_dispatch_order(gpu_lib, monkeypatch, leg, stress):
    name, env, mode, mx, (go, ge), spec = leg
    alphas, betas = _pairs(*spec)
    exp = oracle.align_batch(mode, MX[mx], go, ge, alphas, betas, 10000, 10000, threads=8)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP if mode == 0 else gpu_lib.GNX_CONST_GAP, MX[mx], go, ge)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("GNX_DEBUG_ENTRY", "1")
    plain = gpu_lib.align_batch(p, alphas, betas)
    common.assert_same(plain, exp, name)
    gpu_lib.debug_counter(0, reset=True)
    if stress == "steal":
        monkeypatch.setenv("GNX_TICKET_DELAY", "600")     # ~ 600 x 127 x 64 cycles = 2 ms before the lower half claims ...
        monkeypatch.setenv("GNX_CLAIM_GRACE_US", "50")    # ... and a successor takes an unclaimed predecessor after 50 us
    if stress in ("delay", "both"):
        monkeypatch.setenv("GNX_TICKET_DELAY", "40")   # ~ 40 x 127 x 64 cycles = 0.15 ms before the lower half claims
    if stress in ("occupy", "both"):
        gpu_lib.check(gpu_lib.lib().gnx_debug_occupy(128, 300))  # half the CUs gone for 0.3 s
    t0 = time.time()
    got = gpu_lib.align_batch(p, alphas, betas)
    dt = time.time() - t0
    common.assert_same(got, exp, name + " / " + stress)
    assert dt < 3.0, "%s: %.1f s -- the spin timeout fired, the pipeline did not make progress on its own" % (name, dt)
    stolen = gpu_lib.debug_counter(0, reset=True)
    if stress == "steal" and not common.OUTER_ROUTE_SWITCH and "GNX_NO_PIPE" not in os.environ:
        assert stolen > 0, "%s: nobody ran a predecessor's item -- the n_stolen > 0 paths were not exercised" % name
    time.sleep(0.35)  # let the occupying workgroups end before the next test
