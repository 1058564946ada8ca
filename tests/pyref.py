"""Second, independent statement of the path's semantics in pure Python (small inputs only).

Unlike oracle/gnx_oracle.c (a literal restatement of the Go control flow, tile by tile), this one follows
SURVEY.md Appendix A: one full DP matrix, then a walk in GLOBAL coordinates that applies the two
checkerboard quirks analytically:
  Q1 (affine only): when the walk leaves a tile through its top edge, the state restarts as the
     argmax of (M,I,D) at the entry cell (affineGap.go:305).
  Q2: when the final move leaves through a tile's exact corner, no leading gap is appended
     (affineGap.go:135-139, constGap.go:59-63).
It is the derivation the GPU traceback kernel implements, so agreement of pyref with the literal oracle
on small checkersizes validates that derivation on the CPU before any GPU run.
"""

VNN = -(2 ** 62)


def tmt(a, b, c):
    if a >= b and a >= c:
        return a, 0
    if b >= c:
        return b, 1
    return c, 2


def _emit(route, op, run):
    if run <= 0 and route:
        pass
    if route and route[-1][1] == op:
        route[-1][0] += run
    else:
        route.append([run, op])


def affine(alpha, beta, sc, gap_open, gap_extend, ci=None, cj=None, free_end=False):
    n, m = len(alpha), len(beta)
    big = 1 << 60
    ci = ci or big
    cj = cj or big
    oe, e = gap_open + gap_extend, gap_extend
    M = [[VNN] * (m + 1) for _ in range(n + 1)]
    I = [[VNN] * (m + 1) for _ in range(n + 1)]
    D = [[VNN] * (m + 1) for _ in range(n + 1)]
    tr = [[[0] * (m + 1) for _ in range(n + 1)] for _ in range(3)]
    M[0][0], I[0][0], D[0][0] = 0, gap_open, (0 if free_end else gap_open)
    for j in range(1, m + 1):
        I[0][j] = e + I[0][j - 1]
    for i in range(1, n + 1):
        D[i][0] = (0 if free_end else e) + D[i - 1][0]
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            s = sc[alpha[i - 1]][beta[j - 1]]
            M[i][j], tr[0][i][j] = tmt(s + M[i - 1][j - 1], s + I[i - 1][j - 1], s + D[i - 1][j - 1])
            I[i][j], tr[1][i][j] = tmt(oe + M[i][j - 1], e + I[i][j - 1], oe + D[i][j - 1])
            if free_end and j == m:
                D[i][j], tr[2][i][j] = tmt(M[i - 1][j], I[i - 1][j], D[i - 1][j])
            else:
                D[i][j], tr[2][i][j] = tmt(oe + M[i - 1][j], oe + I[i - 1][j], e + D[i - 1][j])
    score, k = tmt(M[n][m], I[n][m], D[n][m])
    route = []
    i, j = n, m
    up_exit = left_exit = False
    walked = False
    while i > 0 and j > 0:
        walked = True
        _emit(route, k, 1)
        nk = tr[k][i][j]
        up_exit = left_exit = False
        if k != 1:
            up_exit = (i - 1) % ci == 0
            i -= 1
        if k != 2:
            left_exit = (j - 1) % cj == 0
            j -= 1
        k = nk
        if up_exit and i > 0 and j > 0:
            k = tmt(M[i][j], I[i][j], D[i][j])[1]
    if walked:
        if (not up_exit) and left_exit:
            _emit(route, 2, i)
        elif up_exit and (not left_exit):
            _emit(route, 1, j)
    else:
        if i == 0 and j > 0:
            _emit(route, 1, j)
        elif j == 0 and i > 0:
            _emit(route, 2, i)
        else:
            route.append([0, 0])
    route.reverse()
    return score, [(r, o) for r, o in route]


def const(alpha, beta, sc, gap_pen, ci=None, cj=None):
    n, m = len(alpha), len(beta)
    big = 1 << 60
    ci = ci or big
    cj = cj or big
    V = [[0] * (m + 1) for _ in range(n + 1)]
    tr = [[0] * (m + 1) for _ in range(n + 1)]
    for j in range(1, m + 1):
        V[0][j] = V[0][j - 1] + gap_pen
    for i in range(1, n + 1):
        V[i][0] = V[i - 1][0] + gap_pen
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            V[i][j], tr[i][j] = tmt(V[i - 1][j - 1] + sc[alpha[i - 1]][beta[j - 1]], V[i][j - 1] + gap_pen, V[i - 1][j] + gap_pen)
    route = []
    i, j = n, m
    up_exit = left_exit = False
    walked = False
    while i > 0 and j > 0:
        walked = True
        k = tr[i][j]
        _emit(route, k, 1)
        up_exit = left_exit = False
        if k != 1:
            up_exit = (i - 1) % ci == 0
            i -= 1
        if k != 2:
            left_exit = (j - 1) % cj == 0
            j -= 1
    if walked:
        if (not up_exit) and left_exit:
            _emit(route, 2, i)
        elif up_exit and (not left_exit):
            _emit(route, 1, j)
    else:
        if i == 0 and j > 0:
            _emit(route, 1, j)
        elif j == 0 and i > 0:
            _emit(route, 2, i)
        else:
            route.append([0, 0])
    route.reverse()
    return V[n][m], [(r, o) for r, o in route]
