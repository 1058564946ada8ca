"""N2 ("next" row): the seed-extension DPs of the graph aligner, genomeGraph.LeftDynamicAln / RightDynamicAln
(/root/reference/genomeGraph/search.go:234-321).  The reference's own tests of this path only log, so parity is UNPINNED:
the oracle is a literal restatement, checked here against an independent pure-Python statement and known answers; the GPU
path must equal the oracle bit for bit."""
import numpy as np
import pytest

import common
import oracle
from gonomics_amd import cigar, genomeGraph

MX = common.matrices()


def py_gsw(side, alpha, beta, sc, gap, route_in=None, curr_max=0):
    """Independent statement: dict-based matrices, ops as letters."""
    n, m = len(alpha), len(beta)
    M, T = {}, {}

    def tmt(a, b, c):
        if a >= b and a >= c:
            return a, "M"
        if b >= c:
            return b, "I"
        return c, "D"

    route = [[r, "MID"[o]] for r, o in (route_in or [])]
    idx = 0

    def step(op):
        nonlocal idx
        if not route:
            route.append([1, op])
        elif route[idx][1] == op:
            route[idx][0] += 1
        else:
            route.append([1, op])
            idx += 1

    if side == 0:
        for i in range(n + 1):
            M[i, 0] = 0
        for j in range(m + 1):
            M[0, j] = 0
        for i in range(1, n + 1):
            for j in range(1, m + 1):
                v, t = tmt(M[i - 1, j - 1] + int(sc[alpha[i - 1]][beta[j - 1]]), M[i, j - 1] + gap, M[i - 1, j] + gap)
                M[i, j], T[i, j] = max(v, 0), t
        i, j = n, m
        while M[i, j] > 0:
            step(T[i, j])
            i, j = (i - 1, j - 1) if T[i, j] == "M" else ((i, j - 1) if T[i, j] == "I" else (i - 1, j))
        return M[n, m], [(r, "MID".index(o)) for r, o in route], i, j
    best, bi, bj = curr_max, 0, 0
    for i in range(n + 1):
        for j in range(m + 1):
            if i == 0 and j == 0:
                M[0, 0] = 0
            elif i == 0:
                M[i, j], T[i, j] = M[i, j - 1] + gap, "I"
            elif j == 0:
                M[i, j], T[i, j] = M[i - 1, j] + gap, "D"
            else:
                M[i, j], T[i, j] = tmt(M[i - 1, j - 1] + int(sc[alpha[i - 1]][beta[j - 1]]), M[i, j - 1] + gap, M[i - 1, j] + gap)
            if M[i, j] > best:
                best, bi, bj = M[i, j], i, j
    i, j = bi, bj
    while i > 0 or j > 0:
        step(T[i, j])
        i, j = (i - 1, j - 1) if T[i, j] == "M" else ((i, j - 1) if T[i, j] == "I" else (i - 1, j))
    return M[bi, bj], [(r, "MID".index(o)) for r, o in route], bi, bj


def _cases(seed, count, nmax, mmax):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        n, m = int(rng.integers(0, nmax + 1)), int(rng.integers(0, mmax + 1))
        beta = rng.integers(0, 5 if k % 7 == 0 else 4, size=m).astype(np.uint8)
        if m and rng.random() < 0.75:  # related: the target is the read with errors and flanks, like a seed extension
            alpha = common.mutate(rng, beta, 0.06, 0.04)
            alpha = np.concatenate([rng.integers(0, 4, size=int(rng.integers(0, 6))).astype(np.uint8), alpha,
                                    rng.integers(0, 4, size=int(rng.integers(0, 6))).astype(np.uint8)])[:max(n, 1) if n else 0]
        else:
            alpha = rng.integers(0, 4, size=n).astype(np.uint8)
        out.append((alpha, beta))
    return out


def test_gsw_oracle_vs_independent_statement():
    for name, gap in [("HumanChimpTwo", -600), ("Default", -430), ("HumanChimpTwo", -1), ("Default", 0)]:
        for side in (0, 1):
            for alpha, beta in _cases(5 + side, 60, 40, 40):
                assert oracle.gsw_extend(side, MX[name], gap, alpha, beta) == py_gsw(side, alpha, beta, MX[name], gap)


def test_gsw_known_answers():
    sc = MX["HumanChimpTwo"]
    a = np.array([0, 1, 2, 3, 0, 1, 2, 3, 2, 2], dtype=np.uint8)
    perfect = sum(int(sc[x][x]) for x in a)
    assert oracle.gsw_extend(0, sc, -600, a, a) == (perfect, [(10, 0)], 0, 0)   # LeftDynamicAln: all M back to the origin
    assert oracle.gsw_extend(1, sc, -600, a, a) == (perfect, [(10, 0)], 10, 10)  # RightDynamicAln: maximum at (n, m)
    # nothing positive: Left stops at once at (n, m); Right keeps (0, 0)
    b = np.array([3, 3, 3], dtype=np.uint8)
    c = np.array([0, 0, 0], dtype=np.uint8)
    assert oracle.gsw_extend(0, sc, -600, b, c) == (0, [], 3, 3)
    assert oracle.gsw_extend(1, sc, -600, b, c) == (0, [], 0, 0)
    # empty sequences
    e = np.zeros(0, dtype=np.uint8)
    assert oracle.gsw_extend(0, sc, -600, e, a) == (0, [], 0, 10)
    assert oracle.gsw_extend(1, sc, -600, a, e) == (0, [], 0, 0)


def test_go_slice_model():
    """The traversals of the graph aligner share ONE route slice between sibling branches (search.go:185-195), so the restatement and
    the mirrors model Go slices: header (array, offset, len, cap) + Go 1.25's growth rule.  Known facts of the Go runtime (16-byte
    elements): appending one element at a time to nil gives the capacities 1, 2, 4 ... 512, 848; appending 5 elements to a full slice
    of 4 gives 9.  And the aliasing case itself: a second DP started from the route of the first increments the first's cells through
    the shared array until its append outgrows the capacity."""
    import pyref_gsw as ref
    s, caps = ref.GoSlice(), []
    for _ in range(600):
        s = ref.go_append(s, [1, 0]); caps.append(s.cap)
    assert sorted(set(caps)) == [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 848]
    full = ref.go_append(ref.go_append(ref.GoSlice(), [1, 0], [1, 1]), [1, 0], [1, 1])
    assert (full.n, full.cap) == (4, 4) and ref.go_append(full, *[[1, 0]] * 5).cap == 9
    assert genomeGraph._go_next_cap(9, 4) == 9 and genomeGraph._go_next_cap(513, 512) == 848 and genomeGraph._go_next_cap(1, 0) == 1
    a = ref.dp_merge(ref.GoSlice(), [(3, 0), (1, 1)])
    assert [tuple(c) for c in a.cells()] == [(3, 0), (1, 1)] and a.cap == 2
    b = ref.dp_merge(a, [(2, 0), (1, 2)])  # M M D on top of a's route, routeIdx restarting at 0
    assert [tuple(c) for c in b.cells()] == [(5, 0), (1, 1), (1, 2)] and b.cap == 4 and b.arr is not a.arr
    assert [tuple(c) for c in a.cells()] == [(5, 0), (1, 1)]  # the first route, seen through its own header: changed by the second DP
    ga = genomeGraph._merge_route_go(genomeGraph.GoSlice(), [(3, cigar.Match), (1, cigar.Insertion)])
    gb = genomeGraph._merge_route_go(ga, [(2, cigar.Match), (1, cigar.Deletion)])
    assert [(c.RunLength, c.Op) for c in ga] == [(5, cigar.Match), (1, cigar.Insertion)]
    assert [(c.RunLength, c.Op) for c in gb] == [(5, cigar.Match), (1, cigar.Insertion), (1, cigar.Deletion)] and gb.cap == 4
    ga.reverse()  # cigar.ReverseCigar in place: gb has its own array by now
    assert [(c.RunLength, c.Op) for c in ga] == [(1, cigar.Insertion), (5, cigar.Match)] and gb[0].RunLength == 5


def test_gsw_route_carry_over_host_logic():
    """resetDynamicScore is a no-op in the reference: a route passed in is kept and merged with routeIdx restarting at 0.
    genomeGraph._merge_route (product host code) must equal the oracle's literal loop."""
    rng = np.random.default_rng(11)
    sc = MX["HumanChimpTwo"]
    for alpha, beta in _cases(12, 80, 30, 30):
        rin = [(int(rng.integers(1, 5)), int(rng.integers(0, 3))) for _ in range(int(rng.integers(0, 4)))]
        for side in (0, 1):
            s0, plain, i0, j0 = oracle.gsw_extend(side, sc, -600, alpha, beta)
            s1, merged, i1, j1 = oracle.gsw_extend(side, sc, -600, alpha, beta, route_in=rin)
            assert (s0, i0, j0) == (s1, i1, j1)
            assert merged == py_gsw(side, alpha, beta, sc, -600, route_in=rin)[1]
            got = genomeGraph._merge_route([cigar.Cigar(r, cigar.from_col(o)) for r, o in rin], [(r, cigar.from_col(o)) for r, o in plain])
            assert [(c.RunLength, c.Op) for c in got] == [(r, cigar.from_col(o)) for r, o in merged]


def _check_gpu(side, name, gap, pairs):
    sc = MX[name]
    got = genomeGraph.DynamicAlnBatch("left" if side == 0 else "right", [a for a, _ in pairs], [b for _, b in pairs], sc, gap)
    for (alpha, beta), (score, route, i, j) in zip(pairs, got):
        es, er, ei, ej = oracle.gsw_extend(side, sc, gap, alpha, beta)
        assert (score, i, j) == (es, ei, ej), (side, name, gap, len(alpha), len(beta))
        assert [(c.RunLength, c.Op) for c in route] == [(r, cigar.from_col(o)) for r, o in er]


@pytest.mark.gpu
@pytest.mark.parametrize("side", [0, 1])
def test_gsw_gpu_parity(gpu_lib, side):
    for name, gap in [("HumanChimpTwo", -600), ("Default", -430), ("HumanChimpTwo", -1), ("Default", 0)]:
        _check_gpu(side, name, gap, _cases(21 + side, 300, 60, 60))
    # the shape of a real extension (read part <= 150, target = read + perfectScore/600) and multi-strip targets (> 160 rows)
    _check_gpu(side, "HumanChimpTwo", -600, _cases(31 + side, 200, 175, 150))
    _check_gpu(side, "HumanChimpTwo", -600, _cases(41 + side, 40, 700, 400))
    _check_gpu(side, "HumanChimpTwo", -600, [(np.zeros(0, np.uint8), np.zeros(0, np.uint8)), (np.zeros(0, np.uint8), np.array([1, 2], np.uint8)),
                                              (np.array([1, 2], np.uint8), np.zeros(0, np.uint8))])


@pytest.mark.gpu
def test_gsw_gpu_carry_over_and_single(gpu_lib):
    sc = MX["HumanChimpTwo"]
    a = np.array([0, 1, 2, 3, 0, 1, 2, 3, 2, 2], dtype=np.uint8)
    perfect = sum(int(sc[x][x]) for x in a)
    assert genomeGraph.LeftDynamicAln(a, a, sc, -600) == (perfect, [cigar.Cigar(10, cigar.Match)], 0, 0)
    assert genomeGraph.RightDynamicAln(a, a, sc, -600) == (perfect, [cigar.Cigar(10, cigar.Match)], 10, 10)
    rin = [cigar.Cigar(2, cigar.Insertion), cigar.Cigar(1, cigar.Match)]
    for side, fn in ((0, genomeGraph.LeftDynamicAln), (1, genomeGraph.RightDynamicAln)):
        for alpha, beta in _cases(51, 30, 40, 40):
            es, er, ei, ej = oracle.gsw_extend(side, sc, -600, alpha, beta, route_in=[(2, 1), (1, 0)])
            score, route, i, j = fn(alpha, beta, sc, -600, route=rin)
            assert (score, i, j) == (es, ei, ej)
            assert [(c.RunLength, c.Op) for c in route] == [(r, cigar.from_col(o)) for r, o in er]


@pytest.mark.gpu
def test_gsw_gpu_errors(gpu_lib):
    sc = MX["HumanChimpTwo"]
    a = np.array([0, 1, 7], dtype=np.uint8)
    with pytest.raises(gpu_lib.GnxError):
        genomeGraph.LeftDynamicAln(a, a, sc, -600)          # base >= 5: the Go code panics
    b = np.zeros(5000, dtype=np.uint8)
    with pytest.raises(gpu_lib.GnxError):
        genomeGraph.RightDynamicAln(b, b, sc, -600)         # beyond the 4095-base packed maximum (reference matrix: 2480)
    with pytest.raises(gpu_lib.GnxError):
        genomeGraph.LeftDynamicAln(a[:2], a[:2], sc, 5)     # positive gap "penalty"
