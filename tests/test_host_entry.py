"""The host-buffer side of the C ABI (csrc/gnx_host.hip.h): pipelined sub-batches, the resident reference, and the multi-context
(one per GPU) flow -- sharding by DP cells, reference broadcast, ordered gather -- run here with two contexts on ONE device, and
with a 1-rank RCCL communicator so that every RCCL call of the flow is exercised on a 1-GPU box."""
import os

import numpy as np
import pytest

import common
import oracle
from gonomics_amd import align

pytestmark = pytest.mark.gpu
MX = common.matrices()


def _c2_batch(seed, n_pairs, chunk_len=4000):
    reads, chunk = common.c2_workload(seed, n_pairs, read_len=150, chunk_len=chunk_len)
    a_start = np.arange(n_pairs, dtype=np.int64) * 150
    a_len = np.full(n_pairs, 150, dtype=np.int64)
    b_start = np.zeros(n_pairs, dtype=np.int64)
    b_len = np.full(n_pairs, chunk_len, dtype=np.int64)
    return reads.reshape(-1), a_start, a_len, chunk, b_start, b_len


def test_pipelined_sub_batches(gpu_lib, monkeypatch):
    """a batch cut into sub-batches (stager thread, pinned staging, second stream) gives what one launch gives; ragged lengths and
    disjoint per-pair windows travel with their sub-batches"""
    a, a_start, a_len, chunk, b_start, b_len = _c2_batch(71, 1000)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, a, a_start, a_len, chunk, b_start, b_len, threads=8)
    for sub in ("1000000", "256", "96"):
        monkeypatch.setenv("GNX_HOST_SUB", sub)
        common.assert_same(gpu_lib.align_batch_windows(p, a, a_start, a_len, chunk, b_start, b_len), exp, "sub " + sub)
    # a CIGAR buffer that starts too small: ConstGap of the same pairs has ~100 x the runs (the retry path of every sub-batch)
    pc = gpu_lib.make_params(gpu_lib.GNX_CONST_GAP, MX["HumanChimpTwo"], -430)
    expc = oracle.align_batch_windows(1, MX["HumanChimpTwo"], -430, 0, a, a_start, a_len, chunk, b_start, b_len, threads=8)
    common.assert_same(gpu_lib.align_batch_windows(pc, a, a_start, a_len, chunk, b_start, b_len), expc, "const")
    alphas, betas = common.random_pairs(72, 300, 1, 300, 1, 700)
    monkeypatch.setenv("GNX_HOST_SUB", "64")
    for mode, go, ge in ((0, -400, -30), (1, -430, 0), (3, -400, -30)):
        pm = gpu_lib.make_params(mode, MX["Default"], go, ge)
        common.assert_same(gpu_lib.align_batch(pm, alphas, betas), oracle.align_batch(mode, MX["Default"], go, ge, alphas, betas, threads=8), "mode %d" % mode)


def test_sub_batches_with_reads_of_several_row_blocks(gpu_lib, monkeypatch):
    """the host entry point with reads of 100 .. 700 bases (1 .. 5 row blocks of the fast path, grouped per sub-batch) against windows
    of 800 .. 2500 bases: sub-batches, the CIGAR buffer retry, global and local mode"""
    rng = np.random.default_rng(74)
    L = 2500
    chunk = rng.integers(0, 4, size=L).astype(np.uint8)
    n_pairs = 700
    lens = rng.choice([100, 150, 200, 250, 320, 400, 700], size=n_pairs).astype(np.int64)
    reads = []
    for k in range(n_pairs):
        n = int(lens[k]); o = int(rng.integers(0, L - n - 60))
        r = common.mutate(rng, chunk[o:o + n + 50], sub=0.02, indel=0.006, geo=0.5)[:n]
        if len(r) < n:
            r = np.concatenate([r, rng.integers(0, 4, size=n - len(r)).astype(np.uint8)])
        reads.append(r)
    a = np.concatenate(reads)
    a_start = np.zeros(n_pairs, dtype=np.int64); a_start[1:] = np.cumsum(lens)[:-1]
    b_len = rng.choice([800, 1500, 2500], size=n_pairs).astype(np.int64)
    b_len = np.maximum(b_len, lens + 100)
    b_start = np.zeros(n_pairs, dtype=np.int64)
    monkeypatch.setenv("GNX_FASTPATH", "2")  # (the batches are small: no routing rule)
    for mode, go, ge in ((0, -600, -150), (3, -600, -150)):
        p = gpu_lib.make_params(mode, MX["HumanChimpTwo"], go, ge)
        if mode == 3:  # AffineGapLocal(target = window, query = read)
            exp = oracle.align_batch_windows(mode, MX["HumanChimpTwo"], go, ge, chunk, b_start, b_len, a, a_start, lens, threads=8)
        else:
            exp = oracle.align_batch_windows(mode, MX["HumanChimpTwo"], go, ge, a, a_start, lens, chunk, b_start, b_len, threads=8)
        for sub in ("1000000", "200", "64"):
            monkeypatch.setenv("GNX_HOST_SUB", sub)
            if mode == 3:
                got = gpu_lib.align_batch_windows(p, chunk, b_start, b_len, a, a_start, lens)
            else:
                got = gpu_lib.align_batch_windows(p, a, a_start, lens, chunk, b_start, b_len)
            common.assert_same(got, exp, "mode %d sub %s" % (mode, sub))


def test_resident_reference(gpu_lib, monkeypatch):
    """gnx_set_reference + gnx_align_batch_by_offset == the same windows passed as host buffers"""
    rng = np.random.default_rng(73)
    ref = rng.integers(0, 4, size=300000).astype(np.uint8)
    ref[rng.random(ref.shape[0]) < 0.001] = 4
    n = 600
    starts = rng.integers(0, ref.shape[0] - 3000, size=n).astype(np.int64)
    lens = np.full(n, 3000, dtype=np.int64)
    reads = []
    for k in range(n):
        o = int(starts[k]) + int(rng.integers(0, 2800))
        reads.append(common.mutate(rng, ref[o:o + 190], sub=0.01, indel=0.004, geo=0.5)[:150])
    a_off = np.zeros(n + 1, dtype=np.int64)
    a_off[1:] = np.cumsum([len(r) for r in reads])
    a_cat = np.concatenate(reads)
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    gpu_lib.set_reference(ref)
    monkeypatch.setenv("GNX_HOST_SUB", "128")
    got = gpu_lib.align_batch_by_offset(p, a_cat, a_off, starts, lens)
    exp = oracle.align_batch(0, MX["HumanChimpTwo"], -600, -150, reads, [ref[s:s + 3000] for s in starts], threads=8)
    common.assert_same(got, exp)
    with pytest.raises(gpu_lib.GnxError):  # a window that leaves the reference
        gpu_lib.align_batch_by_offset(p, a_cat, a_off, starts + ref.shape[0], lens)


def test_synthetic_reference_matches_its_host_statement(gpu_lib):
    """the device-generated reference of config C3 is the pure function of the position that tests / bench.py evaluate on the host"""
    L = gpu_lib.lib()
    seed, length = 3, 50003000
    gpu_lib.check(L.gnx_set_reference_synthetic(length, seed))
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP_HIGHMEM, MX["Default"], -400, -30)
    # read the reference back through alignments: a window aligned against itself is all M with score = sum of diagonal scores
    starts = np.asarray([0, 31, 49999000, 50000500, 50002000], dtype=np.int64)
    lens = np.full(starts.shape[0], 900, dtype=np.int64)
    wins = [gpu_lib.synthetic_reference_bases(int(s), 900, seed) for s in starts]
    assert (wins[2][:1000] == 4).sum() == 0 and (wins[3] == 4).sum() == 500  # the N run sits at [5e7, 5e7 + 1000)
    a_off = np.arange(starts.shape[0] + 1, dtype=np.int64) * 900
    score, ops, off = gpu_lib.align_batch_by_offset(p, np.concatenate(wins), a_off, starts, lens)
    sc = np.asarray(MX["Default"], dtype=np.int64)
    for k, w in enumerate(wins):
        assert int(score[k]) == int(sc[w, w].sum()) and int(off[k + 1] - off[k]) == 1 and int(ops[int(off[k])]["run_length"]) == 900


def test_contexts_created_after_the_reference(gpu_lib, monkeypatch):
    """gnx_set_reference, THEN gnx_init_devices -- the new contexts have no copy of the reference; the first call that
    shards over them brings them up to date from context 0 (it used to hand the workers a null pointer).  Also: a reference that
    is replaced reaches every context, and GNX_HOST_SUB values that are not multiples of 8 do not break the sub-batch loop."""
    L = gpu_lib.lib()
    a, a_start, a_len, chunk, b_start, b_len = _c2_batch(81, 1000, chunk_len=2500)
    a_off = np.concatenate([a_start, [a_start[-1] + 150]])
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, a, a_start, a_len, chunk, b_start, b_len, threads=8)
    chunk2 = chunk[::-1].copy()
    exp2 = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, a, a_start, a_len, chunk2, b_start, b_len, threads=8)
    try:
        gpu_lib.check(L.gnx_shutdown() or 0)
        gpu_lib.check(L.gnx_init(0, 8 << 30))
        gpu_lib.set_reference(chunk)                       # one context exists
        monkeypatch.setenv("GNX_RCCL", "0")
        monkeypatch.setenv("GNX_HOST_SUB", "12")           # sub = 12 used to give K = 84 sub-batches of 16 > n
        assert gpu_lib.init_devices([0, 0, 0], 4 << 30) == 3  # three contexts, two of them new
        common.assert_same(gpu_lib.align_batch_by_offset(p, a, a_off, b_start, b_len), exp, "contexts created after the reference")
        assert gpu_lib.get_timing()["n_contexts"] == 3
        monkeypatch.setenv("GNX_HOST_SUB", "100")
        gpu_lib.set_reference(chunk2)                      # replaced: every context must see the new one
        common.assert_same(gpu_lib.align_batch_by_offset(p, a, a_off, b_start, b_len), exp2, "replaced reference")
        assert gpu_lib.init_devices([0, 0], 4 << 30) == 2  # fewer contexts: still right
        common.assert_same(gpu_lib.align_batch_by_offset(p, a, a_off, b_start, b_len), exp2, "fewer contexts")
    finally:
        monkeypatch.delenv("GNX_RCCL", raising=False)
        monkeypatch.delenv("GNX_HOST_SUB", raising=False)
        L.gnx_shutdown()
        gpu_lib.check(L.gnx_init(0, 8 << 30))


@pytest.mark.parametrize("rccl", ["0", "1", "2"])
def test_two_contexts_on_one_device(gpu_lib, monkeypatch, rccl):
    """the N > 1 flow of the C ABI on a 1-GPU box: contexts (0, 0) -> two worker threads, blocks of equal DP cells, the shared chunk /
    the resident reference copied to the second context, results gathered in input order; must equal the unsharded result.
    rccl == "1" afterwards runs the single-context flow with a 1-rank RCCL communicator (dlopen, ncclCommInitAll, broadcast) and an
    injected RCCL failure BEFORE the group; rccl == "2" the same with the failure INSIDE the group, after an operation has been
    enqueued (VERDICT r3 weak 3 / ADVICE r3): the group must be closed again, the call must still succeed over peer copies
    (transport 3), and after gnx_shutdown a new communicator must come up and work (an open group would swallow ncclCommInitAll)."""
    L = gpu_lib.lib()
    a, a_start, a_len, chunk, b_start, b_len = _c2_batch(74, 1500, chunk_len=3000)
    # ragged: every third read shorter, so that equal DP cells != equal pair counts
    a_len[::3] = 60
    p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
    one = gpu_lib.align_batch_windows(p, a, a_start, a_len, chunk, b_start, b_len)
    tm1 = gpu_lib.get_timing()
    exp = oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, a, a_start, a_len, chunk, b_start, b_len, threads=8)
    common.assert_same(one, exp)
    alphas, betas = common.random_pairs(75, 240, 1, 500, 1, 900)
    pc = gpu_lib.make_params(gpu_lib.GNX_CONST_GAP, MX["Default"], -430, 0, 7, 7)
    expc = oracle.align_batch(1, MX["Default"], -430, 0, alphas, betas, 7, 7, threads=8)
    monkeypatch.setenv("GNX_HOST_SUB", "200")
    try:
        gpu_lib.check(L.gnx_shutdown() or 0)
        monkeypatch.setenv("GNX_RCCL", "0")
        assert gpu_lib.init_devices([0, 0], 8 << 30) == 2
        two = gpu_lib.align_batch_windows(p, a, a_start, a_len, chunk, b_start, b_len)
        tm2 = gpu_lib.get_timing()
        common.assert_same(two, one, "two contexts, shared chunk")
        assert tm2["cells"] == tm1["cells"]
        assert tm2["n_contexts"] == 2 and tm2["transport"] == 2 and tm1["transport"] == 0  # peer copies; one context: no exchange
        common.assert_same(gpu_lib.align_batch(pc, alphas, betas), expc, "two contexts, disjoint windows")
        gpu_lib.set_reference(chunk)
        a_off = np.concatenate([a_start, [a_start[-1] + 150]])
        # by_offset takes concatenated reads: use the full-length ones
        full = np.full(a_len.shape[0], 150, dtype=np.int64)
        ref_two = gpu_lib.align_batch_by_offset(p, a, a_off, b_start, b_len)
        common.assert_same(ref_two, oracle.align_batch_windows(0, MX["HumanChimpTwo"], -600, -150, a, a_start, full, chunk, b_start, b_len, threads=8),
                           "two contexts, resident reference")
        if rccl in ("1", "2"):
            gpu_lib.check(L.gnx_shutdown() or 0)
            monkeypatch.setenv("GNX_RCCL", "1")
            assert gpu_lib.init_devices([0], 8 << 30) == 1
            gpu_lib.set_reference(chunk)  # ncclBroadcast on the 1-rank communicator
            common.assert_same(gpu_lib.align_batch_by_offset(p, a, a_off, b_start, b_len), ref_two, "1-rank RCCL")
            one_r = gpu_lib.align_batch_windows(p, a, a_start, a_len, chunk, b_start, b_len)  # shared chunk: broadcast inside the call
            assert gpu_lib.get_timing()["transport"] == 1
            common.assert_same(one_r, one, "1-rank RCCL, shared chunk")
            # a RCCL call that fails must not fail the alignment: peer copies from then on, and the timing says so
            monkeypatch.setenv("GNX_RCCL_INJECT_FAIL", rccl)
            again = gpu_lib.align_batch_windows(p, a, a_start, a_len, chunk, b_start, b_len)
            monkeypatch.delenv("GNX_RCCL_INJECT_FAIL")
            assert gpu_lib.get_timing()["transport"] == 3
            common.assert_same(again, one, "after an injected RCCL failure")
            common.assert_same(gpu_lib.align_batch_windows(p, a, a_start, a_len, chunk, b_start, b_len), one, "RCCL stays off")
            assert gpu_lib.get_timing()["transport"] == 0  # one context and no usable communicator: nothing is exchanged any more
            # RCCL stays off for the process until gnx_shutdown, also across gnx_init_devices ...
            assert gpu_lib.init_devices([0], 8 << 30) == 1
            common.assert_same(gpu_lib.align_batch_windows(p, a, a_start, a_len, chunk, b_start, b_len), one, "RCCL off across gnx_init_devices")
            assert gpu_lib.get_timing()["transport"] == 0
            # ... and comes back after it: the failed exchange left no group open, the aborted communicator is gone
            gpu_lib.check(L.gnx_shutdown() or 0)
            assert gpu_lib.init_devices([0], 8 << 30) == 1
            common.assert_same(gpu_lib.align_batch_windows(p, a, a_start, a_len, chunk, b_start, b_len), one, "RCCL back after gnx_shutdown")
            assert gpu_lib.get_timing()["transport"] == 1
    finally:
        monkeypatch.delenv("GNX_RCCL", raising=False)
        L.gnx_shutdown()
        gpu_lib.check(L.gnx_init(0, 8 << 30))


c3_reads = common.c3_reads


def rescore_affine_batch(reads, starts, ref_seed, window, score, ops, off, scores, go, ge):
    """size-independent properties of a whole batch, vectorised: per pair (rows consumed, columns consumed, re-scored CIGAR)"""
    from gonomics_amd import _lib
    n = reads.shape[0]
    sc = np.asarray(scores, dtype=np.int64)
    cnt = np.diff(off)
    pair = np.repeat(np.arange(n), cnt)
    run = ops["run_length"].astype(np.int64)
    op = ops["op"]
    di = np.where(op != 1, run, 0)
    dj = np.where(op != 2, run, 0)
    ci = np.cumsum(di) - di
    cj = np.cumsum(dj) - dj
    first = off[:-1]
    i0 = ci - np.repeat(ci[first], cnt)
    j0 = cj - np.repeat(cj[first], cnt)
    rows = np.bincount(pair, weights=di, minlength=n).astype(np.int64)
    cols = np.bincount(pair, weights=dj, minlength=n).astype(np.int64)
    total = np.bincount(pair, weights=np.where(op != 0, go + ge * run, 0), minlength=n).astype(np.int64)
    mm = np.nonzero(op == 0)[0]
    for lo in range(0, mm.shape[0], 200000):  # expand the M runs in slices
        sel = mm[lo:lo + 200000]
        ln = run[sel]
        idx = np.repeat(np.arange(sel.shape[0]), ln)
        k = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
        pp = pair[sel][idx]
        a = reads[pp, i0[sel][idx] + k]
        b = _lib.synthetic_reference_positions(starts[pp] + j0[sel][idx] + k, ref_seed)
        total += np.bincount(pp, weights=sc[a, b], minlength=n).astype(np.int64)
    return rows, cols, total


def test_c3_one_million_reads_against_a_resident_4gb_reference(gpu_lib):
    """config C3 through the C ABI a cgo shim binds: 4.4e9-base reference resident on the device (generated there), ONE call with
    1 048 576 reads at distinct uniform window offsets (8 pipelined sub-batches); every pair property-checked, 10 000 against the oracle"""
    L = gpu_lib.lib()
    ref_len, ref_seed, window, n = 4400000000, 33, 10000, 1 << 20
    sc = MX["HumanChimpTwo"]
    reads, starts = c3_reads(34, n, ref_len, ref_seed, window)
    gpu_lib.check(L.gnx_init(0, 60 << 30))
    try:
        gpu_lib.check(L.gnx_set_reference_synthetic(ref_len, ref_seed))
        bases, dev_bytes, nexc = gpu_lib.reference_info()
        assert bases == ref_len and dev_bytes <= 1.2e9 and nexc >= 87 * 15  # 4.4e9 bases resident in 1.1 GB; the N runs are on the exception list
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, sc, -600, -150)
        a_off = np.arange(n + 1, dtype=np.int64) * 150
        score, ops, off = gpu_lib.align_batch_by_offset(p, reads.reshape(-1), a_off, starts, np.full(n, window, dtype=np.int64))
        tm = gpu_lib.get_timing()
    finally:
        L.gnx_shutdown()
        gpu_lib.check(L.gnx_init(0, 8 << 30))
    common.expect_route(tm, 1)
    assert tm["cells"] == n * 150 * window
    assert starts.max() > (1 << 32)  # windows beyond 4 GB offsets
    rows, cols, total = rescore_affine_batch(reads, starts, ref_seed, window, score, ops, off, sc, -600, -150)
    assert np.array_equal(rows, np.full(n, 150)) and np.array_equal(cols, np.full(n, window))
    assert np.array_equal(total, score)
    k = 10000
    sel = np.linspace(0, n - 1, k).astype(np.int64)  # spread over every sub-batch
    wins = [gpu_lib.synthetic_reference_bases(int(starts[x]), window, ref_seed) for x in sel]
    exp = oracle.align_batch(0, sc, -600, -150, [reads[x] for x in sel], wins, threads=os.cpu_count() or 8)
    assert np.array_equal(score[sel], exp[0])
    got_cnt = (off[sel + 1] - off[sel])
    assert np.array_equal(got_cnt, np.diff(exp[2]))
    flat = np.concatenate([np.arange(off[x], off[x + 1]) for x in sel])
    assert np.array_equal(ops["run_length"][flat], exp[1]["run_length"]) and np.array_equal(ops["op"][flat], exp[1]["op"])


def test_c3_ten_million_reads(gpu_lib):
    """config C3 at its STATED size (BASELINE.json configs[2]): 10 x 1 Mi = 10 485 760 reads of 150 bases against
    windows at uniform offsets of a resident 3e9-base reference, ten gnx_align_batch_by_offset calls ("batches of 1 M"),
    every batch with its own reads and windows.  Every pair: the CIGAR consumes read and window; 10 000 pairs spread over all calls
    and all sub-batches: score + CIGAR against the oracle; two batches re-scored in full (CIGAR score == reported score)."""
    import torch
    L = gpu_lib.lib()
    ref_len, ref_seed, window, n, calls = 3000000000, 3, 10000, 1 << 20, 10
    sc = MX["HumanChimpTwo"]
    gpu_lib.check(L.gnx_init(0, 60 << 30))
    try:
        gpu_lib.check(L.gnx_set_reference_synthetic(ref_len, ref_seed))
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, sc, -600, -150)
        a_off = np.arange(n + 1, dtype=np.int64) * 150
        wl = np.full(n, window, dtype=np.int64)
        cells = 0
        for c in range(calls):
            reads, starts = common.c3_reads_torch(700 + c, n, ref_len, ref_seed, window, 150, torch.device("cuda", 0))
            score, ops, off = gpu_lib.align_batch_by_offset(p, reads.reshape(-1), a_off, starts, wl)
            tm = gpu_lib.get_timing()
            common.expect_route(tm, 1)
            cells += tm["cells"]
            seg = np.repeat(np.arange(n), np.diff(off))
            rl = ops["run_length"]
            assert np.all(np.bincount(seg, weights=np.where(ops["op"] != 1, rl, 0), minlength=n) == 150), c
            assert np.all(np.bincount(seg, weights=np.where(ops["op"] != 2, rl, 0), minlength=n) == window), c
            if c in (0, calls - 1):
                rows, cols, total = rescore_affine_batch(reads, starts, ref_seed, window, score, ops, off, sc, -600, -150)
                assert np.array_equal(total, score), c
            sel = np.linspace(0, n - 1, 1000).astype(np.int64)  # spread over every sub-batch of the call
            wins = [gpu_lib.synthetic_reference_bases(int(starts[x]), window, ref_seed) for x in sel]
            exp = oracle.align_batch(0, sc, -600, -150, [reads[x] for x in sel], wins, threads=os.cpu_count() or 8)
            assert np.array_equal(score[sel], exp[0]), c
            assert np.array_equal(off[sel + 1] - off[sel], np.diff(exp[2])), c
            flat = np.concatenate([np.arange(off[x], off[x + 1]) for x in sel])
            assert np.array_equal(ops["run_length"][flat], exp[1]["run_length"]) and np.array_equal(ops["op"][flat], exp[1]["op"]), c
        assert cells == calls * n * 150 * window
    finally:
        L.gnx_shutdown()
        gpu_lib.check(L.gnx_init(0, 8 << 30))


def test_reference_release_on_every_context(gpu_lib, monkeypatch):
    """gnx_set_reference(len = 0) gives the reference back on EVERY context (not only context 0), and afterwards there is
    no resident reference: gnx_reference_info and gnx_align_batch_by_offset say so."""
    import torch
    L = gpu_lib.lib()
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 4, size=64 << 20).astype(np.uint8)
    monkeypatch.setenv("GNX_RCCL", "0")
    try:
        gpu_lib.check(L.gnx_shutdown() or 0)
        assert gpu_lib.init_devices([0, 0], 8 << 30) == 2
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        gpu_lib.set_reference(ref)  # 16 MB packed, on both contexts
        a, a_start, a_len, chunk, b_start, b_len = _c2_batch(76, 64, chunk_len=3000)
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
        a_off = np.concatenate([a_start, [a_start[-1] + 150]])
        gpu_lib.align_batch_by_offset(p, a, a_off, b_start + 1000, b_len)
        f1 = torch.cuda.mem_get_info(0)[0]
        gpu_lib.check(L.gnx_set_reference(None, 0))
        assert torch.cuda.mem_get_info(0)[0] - f1 >= 2 * (15 << 20), (free0, f1, torch.cuda.mem_get_info(0)[0])  # BOTH copies of the 16 MB came back
        with pytest.raises(gpu_lib.GnxError):
            gpu_lib.reference_info()
        with pytest.raises(gpu_lib.GnxError):
            gpu_lib.align_batch_by_offset(p, a, a_off, b_start, b_len)
    finally:
        L.gnx_shutdown()
        gpu_lib.check(L.gnx_init(0, 8 << 30))


def _ref_with_exceptions(seed, n):
    """a reference with N runs, scattered N, and two stretches of bytes the Go code would panic on (lower-case-like 5 .. 9, dna.Gap 10)"""
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, size=n).astype(np.uint8)
    ref[rng.random(n) < 0.0005] = 4
    for s0 in (3000, 50000, 50063, 50064, 90001):
        ref[s0:s0 + int(rng.integers(1, 700))] = 4
    ref[120000:120040] = rng.integers(5, 11, size=40).astype(np.uint8)
    ref[n - 3] = 7
    return ref


@pytest.mark.parametrize("route", ["default", "general", "snapshot"])
def test_packed_reference_equals_bytes(gpu_lib, monkeypatch, route):
    """The resident reference is kept 2 bits per base + an exception list and the kernels read the packed words (BetaSrc): every mode
    and route must give what the byte windows give -- against the oracle, and against GNX_REF_UNPACK=1 (windows expanded to bytes
    first).  Windows start at every alignment modulo 64, cross N runs, end at the last base."""
    L = gpu_lib.lib()
    n_ref = 200000
    ref = _ref_with_exceptions(91, n_ref)
    rng = np.random.default_rng(92)
    env = {"default": {}, "general": {"GNX_FASTPATH": "0", "GNX_CLONG": "0"}, "snapshot": {"GNX_CLONG": "2", "GNX_FASTPATH": "0"}}[route]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cases = [  # (mode, oracle mode, matrix, go, ge, read lengths, window lengths)
        (gpu_lib.GNX_AFFINE_GAP, 0, "HumanChimpTwo", -600, -150, (100, 160), (900, 2500)),
        (gpu_lib.GNX_AFFINE_GAP, 0, "Default", -400, -30, (200, 700), (800, 1500)),       # several row blocks / strips
        (gpu_lib.GNX_CONST_GAP, 1, "HumanChimpTwo", -430, 0, (100, 500), (300, 1200)),
        (gpu_lib.GNX_AFFINE_GAP_HIGHMEM, 2, "Default", -400, -30, (30, 200), (30, 400)),
        (gpu_lib.GNX_AFFINE_GAP_LOCAL, 3, "HumanChimpTwo", -600, -150, (100, 160), (900, 2500)),  # target = alpha = the read here: by_offset's beta is the window
    ]
    try:
        gpu_lib.set_reference(ref)
        bases, dev_bytes, nexc = gpu_lib.reference_info()
        assert bases == n_ref and nexc > 10 and dev_bytes < 0.27 * n_ref + 4096
        for mode, omode, mx, go, ge, (n_lo, n_hi), (m_lo, m_hi) in cases:
            n_pairs = 300
            wl = rng.integers(m_lo, m_hi + 1, size=n_pairs).astype(np.int64)
            ws = rng.integers(0, 119000 - m_hi, size=n_pairs).astype(np.int64)   # before the stretch of bad bytes
            ws[:64] = 40000 + np.arange(64)                                          # every alignment modulo 64; crosses the N runs at 50 000
            ws[64] = n_ref - 3 - wl[64]                                               # ends right before the bad byte at n - 3 ...
            ws[64] = max(ws[64], 130000)
            reads = []
            for k in range(n_pairs):
                w = ref[ws[k]:ws[k] + wl[k]]
                ln = int(rng.integers(n_lo, n_hi + 1))
                o = int(rng.integers(0, max(1, wl[k] - ln)))
                r = common.mutate(rng, np.minimum(w[o:o + ln], 4), sub=0.03, indel=0.01, geo=0.4, alphabet=4)
                reads.append(r if len(r) else np.zeros(1, np.uint8))
            a_cat = np.concatenate(reads)
            a_off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
            p = gpu_lib.make_params(mode, MX[mx], go, ge)
            got = gpu_lib.align_batch_by_offset(p, a_cat, a_off, ws, wl)
            exp = oracle.align_batch(omode, MX[mx], go, ge, reads, [ref[ws[k]:ws[k] + wl[k]] for k in range(n_pairs)], 10000, 10000, threads=8)
            common.assert_same(got, exp, "packed reference, mode %d, %s" % (mode, route))
            monkeypatch.setenv("GNX_REF_UNPACK", "1")
            common.assert_same(gpu_lib.align_batch_by_offset(p, a_cat, a_off, ws, wl), exp, "windows unpacked to bytes, mode %d" % mode)
            monkeypatch.delenv("GNX_REF_UNPACK")
        # a window that touches a byte >= 5 makes GNX_EBASE (the Go code indexes its 5 x 5 matrix with it); the others do not care
        p = gpu_lib.make_params(gpu_lib.GNX_AFFINE_GAP, MX["HumanChimpTwo"], -600, -150)
        rd = np.minimum(ref[119000:119150], 3)
        a_off = np.asarray([0, 150], dtype=np.int64)
        gpu_lib.align_batch_by_offset(p, rd, a_off, np.asarray([119000]), np.asarray([1000]))  # ends at 120 000: clean
        for bad_start, bad_len in ((119100, 1000), (n_ref - 1000, 1000)):
            with pytest.raises(gpu_lib.GnxError) as ei:
                gpu_lib.align_batch_by_offset(p, rd, a_off, np.asarray([bad_start]), np.asarray([bad_len]))
            assert ei.value.code == gpu_lib.GNX_EBASE
    finally:
        for k in env:
            monkeypatch.delenv(k, raising=False)
        monkeypatch.delenv("GNX_REF_UNPACK", raising=False)
        L.gnx_shutdown()
        gpu_lib.check(L.gnx_init(0, 8 << 30))
