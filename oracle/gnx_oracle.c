/*
 * gnx_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, int64, scalar) of the gonomics `align` pairwise DP hot path:
 *   align/align.go            tripleMaxTrace, Cigar, reverseCigar           (:8-90)
 *   align/affineGap.go        AffineGap, AffineGap_customizeCheckersize,
 *                             highestScore_affineGap, fillTraceback_affineGap,
 *                             writeCigar_affineGap                           (:20-344)
 *   align/constGap.go         ConstGap, ConstGap_customizeCheckersize,
 *                             highestScore, fillTraceback, writeCigar,
 *                             lastCigar                                      (:13-311)
 *   align/affineGap_highMem.go  affineGap_highMem(freeEndGaps), affineTrace  (:57-89,:181-223)
 *   align/constGap_highMem.go   ConstGap_highMem                             (:11-67)
 *
 * The reference is Go and cannot be built in this image (no Go toolchain), so this restatement is
 * pinned by the reference's own golden vectors (tests/golden/, see tests/test_oracle_golden.py):
 * align/affineGap_test.go, align/view_test.go, cmd/globalAlignmentAnchor/testdata/out_alignment.*.tsv,
 * cmd/cigarToBed/testdata/.  The multi-checkerboard (n or m > checkersize) behaviour is a literal
 * restatement of the code; the reference has no test that pins it beyond checkersize 3 on 9 tiny pairs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 *
 * Go semantics preserved: truncating integer division, sign-following %, append() on the route slice,
 * route[0] starting as {0,0} and being overwritten while RunLength==0.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define OR_OK 0
#define OR_EINVAL 1  /* reference would hang / panic on this input (empty seq, base >= 5, bad index) */
#define OR_ENOMEM 2
#define OR_ETRACE 3  /* reference would log.Fatalf("unexpected traceback") */

typedef struct {
    int64_t run; /* align.Cigar.RunLength */
    uint8_t op;  /* align.ColType: 0=ColM 1=ColI 2=ColD */
    uint8_t pad[7];
} or_cigar;

static const int64_t VNN = INT64_MIN / 2; /* align/align.go:8 veryNegNum */

/* Per-thread arena used by the batch driver: the reference allocates its work arrays per call (affineGap.go:20-54,:99);
 * with hundreds of worker threads that turns into page-fault / mmap-lock contention which would understate the CPU
 * baseline, so the workers recycle one zero-filled arena per pair instead (the reference's own engine recycles its
 * buffers the same way, affineGap_highMem.go:29-55).  Outside the batch driver plain calloc/free are used. */
static __thread char *tl_base = NULL;
static __thread size_t tl_cap = 0, tl_used = 0;
static __thread int tl_on = 0;

static void *xcalloc(size_t n, size_t sz) {
    size_t bytes = (n * sz + 63) & ~(size_t)63;
    if (tl_on && tl_used + bytes <= tl_cap) {
        void *p = tl_base + tl_used;
        tl_used += bytes;
        memset(p, 0, n * sz);
        return p;
    }
    return calloc(n ? n : 1, sz ? sz : 1);
}
static int in_arena(const void *p) { return tl_base && (const char *)p >= tl_base && (const char *)p < tl_base + tl_cap; }
static void xfree(void *p) { if (p && !in_arena(p)) free(p); }
static void *xrealloc(void *p, size_t old_bytes, size_t new_bytes) {
    if (p && !in_arena(p)) return realloc(p, new_bytes);
    void *q = xcalloc(1, new_bytes);
    if (q && p) memcpy(q, p, old_bytes);
    return q;
}

/* align/align.go:76-84 */
static inline int64_t tmt(int64_t a, int64_t b, int64_t c, uint8_t *k) {
    if (a >= b && a >= c) { *k = 0; return a; }
    else if (b >= c)      { *k = 1; return b; }
    else                  { *k = 2; return c; }
}

/* growable route == Go slice with append */
typedef struct { or_cigar *v; int64_t len, cap; } route_t;

static int route_init(route_t *r) { /* route := make([]Cigar, 1) */
    r->cap = 16; r->len = 1;
    r->v = (or_cigar *)xcalloc((size_t)r->cap, sizeof(or_cigar));
    return r->v ? OR_OK : OR_ENOMEM;
}
static int route_append(route_t *r, int64_t run, uint8_t op) {
    if (r->len == r->cap) {
        int64_t nc = r->cap * 2;
        or_cigar *nv = (or_cigar *)xrealloc(r->v, (size_t)r->cap * sizeof(or_cigar), (size_t)nc * sizeof(or_cigar));
        if (!nv) return OR_ENOMEM;
        memset(nv + r->cap, 0, (size_t)(nc - r->cap) * sizeof(or_cigar));
        r->v = nv; r->cap = nc;
    }
    memset(&r->v[r->len], 0, sizeof(or_cigar));
    r->v[r->len].run = run; r->v[r->len].op = op; r->len++;
    return OR_OK;
}
/* align/align.go:86-90 */
static void route_reverse(route_t *r) {
    for (int64_t i = 0, j = r->len - 1; i < j; i++, j--) { or_cigar t = r->v[i]; r->v[i] = r->v[j]; r->v[j] = t; }
}

static int bases_ok(const uint8_t *s, int64_t n) {
    for (int64_t i = 0; i < n; i++) if (s[i] >= 5) return 0; /* Go: index out of range on the 5x5 matrix */
    return 1;
}

static inline int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; } /* numbers/minMax.go:18 */

/* align/constGap.go:280-311 lastCigar */
static int last_cigar(int64_t len_alpha, int64_t len_beta, route_t *route, int64_t *routeIdx, uint8_t op_end) {
    int64_t total = 0, last;
    if (op_end == 1) {
        for (int64_t x = 0; x < route->len; x++) if (route->v[x].op == 0 || route->v[x].op == 1) total += route->v[x].run;
        last = len_beta - total;
    } else if (op_end == 2) {
        for (int64_t x = 0; x < route->len; x++) if (route->v[x].op == 0 || route->v[x].op == 2) total += route->v[x].run;
        last = len_alpha - total;
    } else return OR_ETRACE;
    if (route->v[*routeIdx].op == op_end) route->v[*routeIdx].run += last;
    else { int rc = route_append(route, last, op_end); if (rc) return rc; (*routeIdx)++; }
    return OR_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Affine gap, low-memory checkerboard version.  align/affineGap.go:73-144 (driver), :151-207 (step 1),
 * :219-273 (step 2), :287-344 (step 3).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t n, m, ci, cj, ni, nj;  /* ni = n/ci+1 saved rows, nj = m/cj+1 saved columns */
    int64_t *prep_i[3];            /* [k][idx*(m+1)+j]  trace_prep_i */
    int64_t *prep_j[3];            /* [k][idx*(n+1)+i]  trace_prep_j */
} aff_prep;

static void aff_prep_free(aff_prep *p) { for (int k = 0; k < 3; k++) { xfree(p->prep_i[k]); xfree(p->prep_j[k]); } }

/* Step 1.  align/affineGap.go:151-207 (+ initAffineScoring :20-39) */
static int aff_highest_score(const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc,
                             int64_t gapOpen, int64_t gapExtend, int64_t ci, int64_t cj, aff_prep *p, int64_t *score) {
    int64_t *cur[3], *prev[3];
    int rc = OR_OK;
    memset(p, 0, sizeof(*p));
    p->n = n; p->m = m; p->ci = ci; p->cj = cj; p->ni = n / ci + 1; p->nj = m / cj + 1;
    for (int k = 0; k < 3; k++) { cur[k] = prev[k] = NULL; }
    for (int k = 0; k < 3; k++) {
        cur[k] = (int64_t *)xcalloc((size_t)(m + 1), 8); prev[k] = (int64_t *)xcalloc((size_t)(m + 1), 8);
        p->prep_i[k] = (int64_t *)xcalloc((size_t)(p->ni * (m + 1)), 8);
        p->prep_j[k] = (int64_t *)xcalloc((size_t)(p->nj * (n + 1)), 8);
        if (!cur[k] || !prev[k] || !p->prep_i[k] || !p->prep_j[k]) rc = OR_ENOMEM;
    }
    if (rc) goto done;
    int64_t mColumn = n + 1;
    for (int64_t i = 0; i < mColumn; i++) {
        for (int64_t j = 0; j <= m; j++) {
            uint8_t d;
            if (i == 0 && j == 0) {
                cur[0][j] = 0;       p->prep_j[0][(j / cj) * (n + 1) + i] = cur[0][j];
                cur[1][j] = gapOpen; p->prep_j[1][(j / cj) * (n + 1) + i] = cur[1][j];
                cur[2][j] = gapOpen; p->prep_j[2][(j / cj) * (n + 1) + i] = cur[2][j];
            } else if (i == 0) {
                cur[0][j] = VNN;
                cur[1][j] = gapExtend + cur[1][j - 1];
                cur[2][j] = VNN;
                if (j % cj == 0) for (int k = 0; k < 3; k++) p->prep_j[k][(j / cj) * (n + 1) + i] = cur[k][j];
            } else if (j == 0) {
                cur[0][j] = VNN;
                cur[1][j] = VNN;
                cur[2][j] = gapExtend + prev[2][j];
                for (int k = 0; k < 3; k++) p->prep_j[k][(j / cj) * (n + 1) + i] = cur[k][j];
            } else {
                int64_t s = sc[alpha[i - 1] * 5 + beta[j - 1]];
                cur[0][j] = tmt(s + prev[0][j - 1], s + prev[1][j - 1], s + prev[2][j - 1], &d);
                cur[1][j] = tmt(gapOpen + gapExtend + cur[0][j - 1], gapExtend + cur[1][j - 1], gapOpen + gapExtend + cur[2][j - 1], &d);
                cur[2][j] = tmt(gapOpen + gapExtend + prev[0][j], gapOpen + gapExtend + prev[1][j], gapExtend + prev[2][j], &d);
                if (j % cj == 0) for (int k = 0; k < 3; k++) p->prep_j[k][(j / cj) * (n + 1) + i] = cur[k][j];
            }
        }
        if (i % ci == 0 && i < mColumn - 1) {
            for (int k = 0; k < 3; k++) memcpy(p->prep_i[k] + (i / ci) * (m + 1), cur[k], (size_t)(m + 1) * 8);
            for (int k = 0; k < 3; k++) { int64_t *t = prev[k]; prev[k] = cur[k]; cur[k] = t; }
        } else if (i < mColumn - 1) {
            for (int k = 0; k < 3; k++) { int64_t *t = prev[k]; prev[k] = cur[k]; cur[k] = t; }
        }
    }
    { uint8_t d; *score = tmt(cur[0][m], cur[1][m], cur[2][m], &d); }
done:
    for (int k = 0; k < 3; k++) { xfree(cur[k]); xfree(prev[k]); }
    if (rc) aff_prep_free(p);
    return rc;
}

int or_affine_gap_checkersize(const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc,
                              int64_t gapOpen, int64_t gapExtend, int64_t ci, int64_t cj,
                              int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    if (n < 1 || m < 1 || ci < 1 || cj < 1) return OR_EINVAL; /* reference never terminates on empty input */
    if (!bases_ok(alpha, n) || !bases_ok(beta, m)) return OR_EINVAL;
    aff_prep P; int64_t score = 0;
    int rc = aff_highest_score(alpha, n, beta, m, sc, gapOpen, gapExtend, ci, cj, &P, &score);
    if (rc) return rc;
    const int64_t hi = n, hj = m; /* score_highest_i, score_highest_j */
    /* initAffineTrace :43-54 */
    const int64_t tsi = imin(n, ci), tsj = imin(m, cj);
    uint8_t *trace[3] = {0, 0, 0};
    int64_t *cur[3] = {0, 0, 0}, *prev[3] = {0, 0, 0};
    route_t route; route.v = NULL;
    for (int k = 0; k < 3; k++) { trace[k] = (uint8_t *)xcalloc((size_t)(tsi * tsj), 1); if (!trace[k]) rc = OR_ENOMEM; }
    if (!rc) rc = route_init(&route);
    int64_t routeIdx = 0;
    int64_t i_min = -2, j_min = -2;
    uint8_t k_max = 0, k_min = 0;
    int64_t k1 = (hi - 1) / ci, k2 = (hj - 1) / cj;
    while (!rc && k1 >= 0 && k2 >= 0) {
        /* ---- Step 2: fillTraceback_affineGap :219-273 ---- */
        for (int k = 0; k < 3; k++) {
            xfree(cur[k]); xfree(prev[k]);
            cur[k] = (int64_t *)xcalloc((size_t)(m + 1), 8); prev[k] = (int64_t *)xcalloc((size_t)(m + 1), 8);
            if (!cur[k] || !prev[k]) { rc = OR_ENOMEM; break; }
            memcpy(prev[k], P.prep_i[k] + k1 * (m + 1), (size_t)(m + 1) * 8);
        }
        if (rc) break;
        int64_t i_max, j_max;
        if (i_min >= 0) i_max = ci * k1 + 1 + i_min; else i_max = imin(ci * (k1 + 1), hi);
        if (j_min >= 0) j_max = cj * k2 + 1 + j_min; else j_max = imin(cj * (k2 + 1), hj);
        int64_t i_inmax = (i_max - 1) % ci, j_inmax = (j_max - 1) % cj;
        for (int64_t i = ci * k1 + 1; i <= i_max; i++) {
            int64_t ii = (i - 1) % ci;
            int64_t pj = cj * k1 + 1 + ii; /* literal: the reference indexes with checkersize_j here (:252-254) */
            if (pj < 0 || pj > n) { rc = OR_EINVAL; break; } /* Go would panic: index out of range */
            for (int k = 0; k < 3; k++) cur[k][cj * k2] = P.prep_j[k][k2 * (n + 1) + pj];
            for (int64_t j = cj * k2 + 1; j <= j_max; j++) {
                int64_t jj = (j - 1) % cj;
                int64_t s = sc[alpha[i - 1] * 5 + beta[j - 1]];
                cur[0][j] = tmt(s + prev[0][j - 1], s + prev[1][j - 1], s + prev[2][j - 1], &trace[0][ii * tsj + jj]);
                cur[1][j] = tmt(gapOpen + gapExtend + cur[0][j - 1], gapExtend + cur[1][j - 1], gapOpen + gapExtend + cur[2][j - 1], &trace[1][ii * tsj + jj]);
                cur[2][j] = tmt(gapOpen + gapExtend + prev[0][j], gapOpen + gapExtend + prev[1][j], gapExtend + prev[2][j], &trace[2][ii * tsj + jj]);
            }
            if (i <= ci * (k1 + 1) - 1 && i <= hi - 1) for (int k = 0; k < 3; k++) { int64_t *t = prev[k]; prev[k] = cur[k]; cur[k] = t; }
        }
        if (rc) break;
        tmt(cur[0][j_max], cur[1][j_max], cur[2][j_max], &k_max);
        /* ---- Step 3: writeCigar_affineGap :287-344 ---- */
        {
            int64_t ic = (i_min >= 0) ? i_min : i_inmax;
            int64_t jc = (j_min >= 0) ? j_min : j_inmax;
            uint8_t kc = (i_min >= 0 && j_inmax >= 0) ? k_min : k_max;
            int64_t ridx = routeIdx;
            int64_t o_i = 0, o_j = 0; uint8_t o_k = 0; /* Go zero values if the loop body never runs */
            while (ic >= 0 && jc >= 0) {
                if (route.v[ridx].run == 0) { route.v[ridx].run = 1; route.v[ridx].op = kc; }
                else if (route.v[ridx].op == kc) route.v[ridx].run += 1;
                else { rc = route_append(&route, 1, kc); if (rc) break; ridx++; }
                switch (kc) {
                case 0: kc = trace[0][ic * tsj + jc]; ic--; jc--; break;
                case 1: kc = trace[1][ic * tsj + jc]; jc--; break;
                case 2: kc = trace[2][ic * tsj + jc]; ic--; break;
                default: rc = OR_ETRACE;
                }
                if (rc) break;
                o_i = ic; o_j = jc; o_k = kc;
            }
            if (rc) break;
            routeIdx = ridx; i_min = o_i; j_min = o_j; k_min = o_k;
        }
        if (i_min < 0 && j_min < 0) { k1--; k2--; }
        else if (i_min < 0) k1--;
        else if (j_min < 0) k2--;
    }
    /* Step 4 :135-139 */
    if (!rc) {
        if (i_min != -1 && j_min == -1) rc = last_cigar(n, m, &route, &routeIdx, 2);
        else if (i_min == -1 && j_min != -1) rc = last_cigar(n, m, &route, &routeIdx, 1);
    }
    for (int k = 0; k < 3; k++) { xfree(trace[k]); xfree(cur[k]); xfree(prev[k]); }
    aff_prep_free(&P);
    if (rc) { xfree(route.v); return rc; }
    route_reverse(&route);
    *out_score = score; *out_ops = route.v; *out_nops = route.len;
    return OR_OK;
}

/* align/affineGap.go:59-68 */
int or_affine_gap(const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc,
                  int64_t gapOpen, int64_t gapExtend, int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    return or_affine_gap_checkersize(alpha, n, beta, m, sc, gapOpen, gapExtend, 10000, 10000, out_score, out_ops, out_nops);
}

/* ------------------------------------------------------------------------------------------------
 * Constant gap, low-memory checkerboard version.  align/constGap.go:73-124 (driver; ConstGap :13-68 is
 * the same body with 10000x10000), :129-176 (step 1), :185-222 (step 2), :230-275 (step 3).
 * ---------------------------------------------------------------------------------------------- */
int or_const_gap_checkersize(const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc,
                             int64_t gapPen, int64_t ci, int64_t cj,
                             int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    if (n < 1 || m < 1 || ci < 1 || cj < 1) return OR_EINVAL;
    if (!bases_ok(alpha, n) || !bases_ok(beta, m)) return OR_EINVAL;
    int rc = OR_OK;
    const int64_t ni = n / ci + 1, nj = m / cj + 1;
    int64_t *cur = (int64_t *)xcalloc((size_t)(m + 1), 8), *prev = (int64_t *)xcalloc((size_t)(m + 1), 8);
    int64_t *prep_i = (int64_t *)xcalloc((size_t)(ni * (m + 1)), 8), *prep_j = (int64_t *)xcalloc((size_t)(nj * (n + 1)), 8);
    const int64_t tsi = imin(n, ci), tsj = imin(m, cj);
    uint8_t *trace = (uint8_t *)xcalloc((size_t)(tsi * tsj), 1);
    route_t route; route.v = NULL;
    if (!cur || !prev || !prep_i || !prep_j || !trace) rc = OR_ENOMEM;
    if (!rc) rc = route_init(&route);
    int64_t score = 0;
    const int64_t mColumn = n + 1;
    /* Step 1: highestScore :129-176 */
    for (int64_t i = 0; !rc && i < mColumn; i++) {
        for (int64_t j = 0; j <= m; j++) {
            uint8_t d;
            if (i == 0 && j == 0) { cur[j] = 0; prep_j[(j / cj) * (n + 1) + i] = cur[j]; }
            else if (i == 0) { cur[j] = cur[j - 1] + gapPen; if (j % cj == 0) prep_j[(j / cj) * (n + 1) + i] = cur[j]; }
            else if (j == 0) { cur[j] = prev[j] + gapPen; prep_j[(j / cj) * (n + 1) + i] = cur[j]; }
            else {
                cur[j] = tmt(prev[j - 1] + sc[alpha[i - 1] * 5 + beta[j - 1]], cur[j - 1] + gapPen, prev[j] + gapPen, &d);
                if (j % cj == 0) prep_j[(j / cj) * (n + 1) + i] = cur[j];
            }
        }
        if (i % ci == 0 && i < mColumn - 1) { memcpy(prep_i + (i / ci) * (m + 1), cur, (size_t)(m + 1) * 8); int64_t *t = prev; prev = cur; cur = t; }
        else if (i < mColumn - 1) { int64_t *t = prev; prev = cur; cur = t; }
    }
    if (!rc) score = cur[m];
    const int64_t hi = n, hj = m;
    int64_t routeIdx = 0, i_min = -2, j_min = -2;
    int64_t k1 = (hi - 1) / ci, k2 = (hj - 1) / cj;
    while (!rc && k1 >= 0 && k2 >= 0) {
        /* Step 2: fillTraceback :185-222 */
        memset(cur, 0, (size_t)(m + 1) * 8); /* fresh make() in the reference */
        memcpy(prev, prep_i + k1 * (m + 1), (size_t)(m + 1) * 8);
        int64_t i_max, j_max;
        if (i_min >= 0) i_max = ci * k1 + 1 + i_min; else i_max = imin(ci * (k1 + 1), hi);
        if (j_min >= 0) j_max = cj * k2 + 1 + j_min; else j_max = imin(cj * (k2 + 1), hj);
        int64_t i_inmax = (i_max - 1) % ci, j_inmax = (j_max - 1) % cj;
        for (int64_t i = ci * k1 + 1; i <= i_max; i++) {
            int64_t ii = (i - 1) % ci;
            int64_t pj = cj * k1 + 1 + ii; /* literal (:207) */
            if (pj < 0 || pj > n) { rc = OR_EINVAL; break; }
            cur[cj * k2] = prep_j[k2 * (n + 1) + pj];
            for (int64_t j = cj * k2 + 1; j <= j_max; j++) {
                int64_t jj = (j - 1) % cj;
                cur[j] = tmt(prev[j - 1] + sc[alpha[i - 1] * 5 + beta[j - 1]], cur[j - 1] + gapPen, prev[j] + gapPen, &trace[ii * tsj + jj]);
            }
            if (i <= ci * (k1 + 1) - 1 && i <= hi - 1) { int64_t *t = prev; prev = cur; cur = t; }
        }
        if (rc) break;
        /* Step 3: writeCigar :230-275 */
        {
            int64_t ic = (i_min >= 0) ? i_min : i_inmax;
            int64_t jc = (j_min >= 0) ? j_min : j_inmax;
            int64_t ridx = routeIdx, o_i = 0, o_j = 0;
            while (ic >= 0 && jc >= 0) {
                uint8_t t = trace[ic * tsj + jc];
                if (route.v[ridx].run == 0) { route.v[ridx].run = 1; route.v[ridx].op = t; }
                else if (route.v[ridx].op == t) route.v[ridx].run += 1;
                else { rc = route_append(&route, 1, t); if (rc) break; ridx++; }
                switch (t) {
                case 0: ic--; jc--; break;
                case 1: jc--; break;
                case 2: ic--; break;
                default: rc = OR_ETRACE;
                }
                if (rc) break;
                o_i = ic; o_j = jc;
            }
            if (rc) break;
            routeIdx = ridx; i_min = o_i; j_min = o_j;
        }
        if (i_min < 0 && j_min < 0) { k1--; k2--; }
        else if (i_min < 0) k1--;
        else if (j_min < 0) k2--;
    }
    if (!rc) {
        if (i_min != -1 && j_min == -1) rc = last_cigar(n, m, &route, &routeIdx, 2);
        else if (i_min == -1 && j_min != -1) rc = last_cigar(n, m, &route, &routeIdx, 1);
    }
    xfree(cur); xfree(prev); xfree(prep_i); xfree(prep_j); xfree(trace);
    if (rc) { xfree(route.v); return rc; }
    route_reverse(&route);
    *out_score = score; *out_ops = route.v; *out_nops = route.len;
    return OR_OK;
}

/* align/constGap.go:13-68 */
int or_const_gap(const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc, int64_t gapPen,
                 int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    return or_const_gap_checkersize(alpha, n, beta, m, sc, gapPen, 10000, 10000, out_score, out_ops, out_nops);
}

/* ------------------------------------------------------------------------------------------------
 * High-memory twins.  align/affineGap_highMem.go:181-223 (fill) + :57-89 (affineTrace);
 * free_end_gaps != 0 is AffineGapLocal (:105-107).  Empty sequences are legal here.
 * ---------------------------------------------------------------------------------------------- */
int or_affine_gap_highmem(const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc,
                          int64_t gapOpen, int64_t gapExtend, int free_end_gaps,
                          int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    if (n < 0 || m < 0) return OR_EINVAL;
    if (!bases_ok(alpha, n) || !bases_ok(beta, m)) return OR_EINVAL;
    int rc = OR_OK;
    int64_t *cur[3], *prev[3]; uint8_t *trace[3];
    const int64_t W = m + 1;
    for (int k = 0; k < 3; k++) {
        cur[k] = (int64_t *)xcalloc((size_t)W, 8); prev[k] = (int64_t *)xcalloc((size_t)W, 8);
        trace[k] = (uint8_t *)xcalloc((size_t)((n + 1) * W), 1);
        if (!cur[k] || !prev[k] || !trace[k]) rc = OR_ENOMEM;
    }
    route_t route; route.v = NULL;
    if (!rc) rc = route_init(&route);
    const int64_t mColumn = n + 1;
    for (int64_t i = 0; !rc && i < mColumn; i++) {
        for (int64_t j = 0; j < W; j++) {
            if (i == 0 && j == 0) {
                cur[0][j] = 0; cur[1][j] = gapOpen; cur[2][j] = free_end_gaps ? 0 : gapOpen;
            } else if (i == 0) {
                cur[0][j] = VNN; cur[1][j] = gapExtend + cur[1][j - 1]; trace[1][i * W + j] = 1; cur[2][j] = VNN;
            } else if (j == 0) {
                cur[0][j] = VNN; cur[1][j] = VNN;
                cur[2][j] = (free_end_gaps ? 0 : gapExtend) + prev[2][j];
                trace[2][i * W + j] = 2;
            } else {
                int64_t s = sc[alpha[i - 1] * 5 + beta[j - 1]];
                cur[0][j] = tmt(s + prev[0][j - 1], s + prev[1][j - 1], s + prev[2][j - 1], &trace[0][i * W + j]);
                cur[1][j] = tmt(gapOpen + gapExtend + cur[0][j - 1], gapExtend + cur[1][j - 1], gapOpen + gapExtend + cur[2][j - 1], &trace[1][i * W + j]);
                if (free_end_gaps && j == W - 1)
                    cur[2][j] = tmt(0 + 0 + prev[0][j], 0 + 0 + prev[1][j], 0 + prev[2][j], &trace[2][i * W + j]);
                else
                    cur[2][j] = tmt(gapOpen + gapExtend + prev[0][j], gapOpen + gapExtend + prev[1][j], gapExtend + prev[2][j], &trace[2][i * W + j]);
            }
        }
        if (i < mColumn - 1) for (int k = 0; k < 3; k++) { int64_t *t = prev[k]; prev[k] = cur[k]; cur[k] = t; }
    }
    int64_t maxScore = 0;
    if (!rc) { /* affineTrace :57-89 */
        uint8_t k;
        maxScore = tmt(cur[0][m], cur[1][m], cur[2][m], &k);
        int64_t i = n, j = m, ridx = 0;
        while (i > 0 || j > 0) {
            if (route.v[ridx].run == 0) { route.v[ridx].run = 1; route.v[ridx].op = k; }
            else if (route.v[ridx].op == k) route.v[ridx].run += 1;
            else { rc = route_append(&route, 1, k); if (rc) break; ridx++; }
            if (i < 0 || j < 0) { rc = OR_EINVAL; break; } /* Go would panic: negative index */
            switch (k) {
            case 0: k = trace[0][i * W + j]; i--; j--; break;
            case 1: k = trace[1][i * W + j]; j--; break;
            case 2: k = trace[2][i * W + j]; i--; break;
            default: rc = OR_ETRACE;
            }
            if (rc) break;
        }
    }
    for (int k = 0; k < 3; k++) { xfree(cur[k]); xfree(prev[k]); xfree(trace[k]); }
    if (rc) { xfree(route.v); return rc; }
    route_reverse(&route);
    *out_score = maxScore; *out_ops = route.v; *out_nops = route.len;
    return OR_OK;
}

/* align/constGap_highMem.go:11-67 */
int or_const_gap_highmem(const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc, int64_t gapPen,
                         int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    if (n < 0 || m < 0) return OR_EINVAL;
    if (!bases_ok(alpha, n) || !bases_ok(beta, m)) return OR_EINVAL;
    int rc = OR_OK;
    const int64_t W = m + 1;
    int64_t *cur = (int64_t *)xcalloc((size_t)W, 8), *prev = (int64_t *)xcalloc((size_t)W, 8);
    uint8_t *trace = (uint8_t *)xcalloc((size_t)((n + 1) * W), 1);
    route_t route; route.v = NULL;
    if (!cur || !prev || !trace) rc = OR_ENOMEM;
    if (!rc) rc = route_init(&route);
    for (int64_t i = 0; !rc && i <= n; i++) {
        for (int64_t j = 0; j < W; j++) {
            if (i == 0 && j == 0) cur[j] = 0;
            else if (i == 0) { cur[j] = cur[j - 1] + gapPen; trace[i * W + j] = 1; }
            else if (j == 0) { cur[j] = prev[j] + gapPen; trace[i * W + j] = 2; }
            else cur[j] = tmt(prev[j - 1] + sc[alpha[i - 1] * 5 + beta[j - 1]], cur[j - 1] + gapPen, prev[j] + gapPen, &trace[i * W + j]);
        }
        if (i < n) { int64_t *t = prev; prev = cur; cur = t; }
    }
    int64_t score = rc ? 0 : cur[m];
    if (!rc) {
        int64_t i = n, j = m, ridx = 0;
        while (i > 0 || j > 0) {
            uint8_t t = trace[i * W + j];
            if (route.v[ridx].run == 0) { route.v[ridx].run = 1; route.v[ridx].op = t; }
            else if (route.v[ridx].op == t) route.v[ridx].run += 1;
            else { rc = route_append(&route, 1, t); if (rc) break; ridx++; }
            switch (t) {
            case 0: i--; j--; break;
            case 1: j--; break;
            case 2: i--; break;
            default: rc = OR_ETRACE;
            }
            if (rc) break;
        }
    }
    xfree(cur); xfree(prev); xfree(trace);
    if (rc) { xfree(route.v); return rc; }
    route_reverse(&route);
    *out_score = score; *out_ops = route.v; *out_nops = route.len;
    return OR_OK;
}

/* ------------------------------------------------------------------------------------------------
 * "Next" row N1: the chunk / multiple-alignment variants.  They are all the same full-matrix affine DP
 * (initAffineScoringAndTrace + affineTrace, affineGap_highMem.go:13-27,57-89) over a per-cell score:
 *   AffineGapChunk           affineGap_highMem.go:227-268   cell = ungappedRegionScore of two chunks (ungapped.go:7-13)
 *   multipleAffineGap        affineGap_highMem.go:270-306   cell = scoreColumnMatch (multiAlign.go:82-102)
 *   multipleAffineGapChunk   affineGap_highMem.go:308-353   cell = ungappedRegionColumnScore (multiAlign.go:104-110)
 * gapExtend is multiplied by chunkSize, gapOpen is not; run lengths are multiplied by chunkSize at the end
 * (expandCigarRunLength :91-95).  or_scored_affine() is that common loop over an explicit n x m score matrix.
 * ---------------------------------------------------------------------------------------------- */
static int or_scored_affine(int64_t n, int64_t m, const int64_t *S /* n*m row-major */, int64_t gapOpen, int64_t gapExt,
                            int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    int rc = OR_OK;
    int64_t *cur[3], *prev[3]; uint8_t *trace[3];
    const int64_t W = m + 1;
    for (int k = 0; k < 3; k++) {
        cur[k] = (int64_t *)xcalloc((size_t)W, 8); prev[k] = (int64_t *)xcalloc((size_t)W, 8);
        trace[k] = (uint8_t *)xcalloc((size_t)((n + 1) * W), 1);
        if (!cur[k] || !prev[k] || !trace[k]) rc = OR_ENOMEM;
    }
    route_t route; route.v = NULL;
    if (!rc) rc = route_init(&route);
    for (int64_t i = 0; !rc && i <= n; i++) {
        for (int64_t j = 0; j < W; j++) {
            if (i == 0 && j == 0) { cur[0][j] = 0; cur[1][j] = gapOpen; cur[2][j] = gapOpen; }
            else if (i == 0) { cur[0][j] = VNN; cur[1][j] = gapExt + cur[1][j - 1]; trace[1][i * W + j] = 1; cur[2][j] = VNN; }
            else if (j == 0) { cur[0][j] = VNN; cur[1][j] = VNN; cur[2][j] = gapExt + prev[2][j]; trace[2][i * W + j] = 2; }
            else {
                const int64_t s = S[(i - 1) * m + (j - 1)];
                cur[0][j] = tmt(s + prev[0][j - 1], s + prev[1][j - 1], s + prev[2][j - 1], &trace[0][i * W + j]);
                cur[1][j] = tmt(gapOpen + gapExt + cur[0][j - 1], gapExt + cur[1][j - 1], gapOpen + gapExt + cur[2][j - 1], &trace[1][i * W + j]);
                cur[2][j] = tmt(gapOpen + gapExt + prev[0][j], gapOpen + gapExt + prev[1][j], gapExt + prev[2][j], &trace[2][i * W + j]);
            }
        }
        if (i < n) for (int k = 0; k < 3; k++) { int64_t *t = prev[k]; prev[k] = cur[k]; cur[k] = t; }
    }
    int64_t maxScore = 0;
    if (!rc) { /* affineTrace */
        uint8_t k;
        maxScore = tmt(cur[0][m], cur[1][m], cur[2][m], &k);
        int64_t i = n, j = m, ridx = 0;
        while (i > 0 || j > 0) {
            if (route.v[ridx].run == 0) { route.v[ridx].run = 1; route.v[ridx].op = k; }
            else if (route.v[ridx].op == k) route.v[ridx].run += 1;
            else { rc = route_append(&route, 1, k); if (rc) break; ridx++; }
            if (i < 0 || j < 0) { rc = OR_EINVAL; break; }
            switch (k) {
            case 0: k = trace[0][i * W + j]; i--; j--; break;
            case 1: k = trace[1][i * W + j]; j--; break;
            case 2: k = trace[2][i * W + j]; i--; break;
            default: rc = OR_ETRACE;
            }
            if (rc) break;
        }
    }
    for (int k = 0; k < 3; k++) { xfree(cur[k]); xfree(prev[k]); xfree(trace[k]); }
    if (rc) { xfree(route.v); return rc; }
    route_reverse(&route);
    *out_score = maxScore; *out_ops = route.v; *out_nops = route.len;
    return OR_OK;
}

/* multiAlign.go:82-102; returns OR_EINVAL where Go would panic (index out of range / integer divide by zero).
 * A group is `g` sequences of `len` bases, sequence-major. */
static int score_column_match(const uint8_t *A, int ga, int64_t la, const uint8_t *B, int gb, int64_t lb,
                              int64_t acol, int64_t bcol, const int64_t *sc, int64_t *out) {
    int64_t sum = 0, count = 0;
    for (int x = 0; x < ga; x++) {
        uint8_t a = A[x * la + acol];
        if (a >= 5 && a <= 9) a -= 5;
        for (int y = 0; y < gb; y++) {
            uint8_t b = B[y * lb + bcol];
            if (b >= 5 && b <= 9) b -= 5;
            if (a != 10 && b != 10) {
                if (a >= 5 || b >= 5) return OR_EINVAL;
                sum += sc[a * 5 + b];
                count++;
            }
        }
    }
    if (count == 0) return OR_EINVAL;
    *out = sum / count; /* C and Go both truncate toward zero */
    return OR_OK;
}

/* AffineGapChunk (affineGap_highMem.go:227-268) */
int or_affine_gap_chunk(const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc,
                        int64_t gapOpen, int64_t gapExtend, int64_t chunk,
                        int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    if (chunk < 1 || n < 0 || m < 0 || n % chunk != 0 || m % chunk != 0) return OR_EINVAL; /* log.Fatalf in the reference */
    if (!bases_ok(alpha, n) || !bases_ok(beta, m)) return OR_EINVAL;
    const int64_t nc = n / chunk, mc = m / chunk;
    int64_t *S = (int64_t *)calloc((size_t)(nc * mc > 0 ? nc * mc : 1), 8);
    if (!S) return OR_ENOMEM;
    for (int64_t i = 0; i < nc; i++)
        for (int64_t j = 0; j < mc; j++) {
            int64_t a = 0;
            for (int64_t k = 0; k < chunk; k++) a += sc[alpha[i * chunk + k] * 5 + beta[j * chunk + k]];
            S[i * mc + j] = a;
        }
    int rc = or_scored_affine(nc, mc, S, gapOpen, gapExtend * chunk, out_score, out_ops, out_nops);
    free(S);
    if (!rc) for (int64_t x = 0; x < *out_nops; x++) (*out_ops)[x].run *= chunk;
    return rc;
}

/* multipleAffineGap (chunk == 1, :270-306) and multipleAffineGapChunk (:308-353) */
int or_multiple_affine_gap(const uint8_t *A, int ga, int64_t la, const uint8_t *B, int gb, int64_t lb, const int64_t *sc,
                           int64_t gapOpen, int64_t gapExtend, int64_t chunk,
                           int64_t *out_score, or_cigar **out_ops, int64_t *out_nops) {
    if (chunk < 1 || ga < 1 || gb < 1 || la < 0 || lb < 0 || la % chunk != 0 || lb % chunk != 0) return OR_EINVAL;
    const int64_t nc = la / chunk, mc = lb / chunk;
    int64_t *S = (int64_t *)calloc((size_t)(nc * mc > 0 ? nc * mc : 1), 8);
    if (!S) return OR_ENOMEM;
    int rc = OR_OK;
    for (int64_t i = 0; !rc && i < nc; i++)
        for (int64_t j = 0; !rc && j < mc; j++) {
            int64_t a = 0, v;
            for (int64_t k = 0; k < chunk; k++) { rc = score_column_match(A, ga, la, B, gb, lb, i * chunk + k, j * chunk + k, sc, &v); if (rc) break; a += v; }
            S[i * mc + j] = a;
        }
    if (!rc) rc = or_scored_affine(nc, mc, S, gapOpen, gapExtend * chunk, out_score, out_ops, out_nops);
    free(S);
    if (!rc) for (int64_t x = 0; x < *out_nops; x++) (*out_ops)[x].run *= chunk;
    return rc;
}

/* ---- "next" row N2: seed-extension DPs of the graph aligner ------------------------------------------------------------
 * genomeGraph/search.go:234-276 LeftDynamicAln, :278-321 RightDynamicAln, cigar/tools.go:58-66 TripleMaxTrace.
 * Parity UNPINNED: the reference's tests of this path only log; these are literal restatements.
 * resetDynamicScore (search.go:104-107) takes its argument by value, so it is a no-op: the route the caller passes in is kept
 * (route_in / n_in) and the loop's routeIdx still starts at 0 -- the first traced op is compared with route[0], not with the
 * last element.  With n_in = 0 the result is the plain run-length encoding in traceback order.  currMax likewise starts at
 * whatever the caller's copy holds (curr_max_in; the callers' value is always 0).
 * Ops here: 0/1/2 for cigar 'M'/'I'/'D'.  out_route = the whole route (carried-over elements included). */
static int gsw_trace_step(or_cigar **route, int64_t *len, int64_t *cap, int64_t *routeIdx, uint8_t op) {
    if (*len == 0 || (*route)[*routeIdx].op != op) {
        if (*len == 0) {
            /* route = append(route, Cigar{1, op}) on the empty route; routeIdx stays 0 */
        } else {
            (*routeIdx)++;
        }
        if (*len == *cap) {
            int64_t ncap = *cap ? 2 * *cap : 16;
            or_cigar *nr = (or_cigar *)realloc(*route, (size_t)ncap * sizeof(or_cigar));
            if (!nr) return OR_ENOMEM;
            *route = nr; *cap = ncap;
        }
        memset(&(*route)[*len], 0, sizeof(or_cigar));
        (*route)[*len].run = 1; (*route)[*len].op = op;
        (*len)++;
    } else {
        (*route)[*routeIdx].run += 1;
    }
    return OR_OK;
}

int or_gsw_extend(int side, const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc, int64_t gapPen,
                  const or_cigar *route_in, int64_t n_in, int64_t curr_max_in,
                  int64_t *out_score, int64_t *out_i, int64_t *out_j, or_cigar **out_route, int64_t *out_len) {
    if (!bases_ok(alpha, n) || !bases_ok(beta, m)) return OR_EINVAL;
    const int64_t W = m + 1;
    int64_t *mat = (int64_t *)calloc((size_t)((n + 1) * W), sizeof(int64_t));
    uint8_t *tr = (uint8_t *)calloc((size_t)((n + 1) * W), 1);
    int64_t cap = n_in > 16 ? 2 * n_in : 16, len = n_in, routeIdx = 0, i, j;
    or_cigar *route = (or_cigar *)malloc((size_t)cap * sizeof(or_cigar));
    if (!mat || !tr || !route) { free(mat); free(tr); free(route); return OR_ENOMEM; }
    if (n_in) memcpy(route, route_in, (size_t)n_in * sizeof(or_cigar));
    int rc = OR_OK;
    uint8_t k;
    if (side == 0) { /* LeftDynamicAln */
        /* m[i][0] = 0, m[0][j] = 0 (calloc) */
        for (i = 1; i < n + 1; i++) {
            for (j = 1; j < m + 1; j++) {
                mat[i * W + j] = tmt(mat[(i - 1) * W + j - 1] + sc[alpha[i - 1] * 5 + beta[j - 1]], mat[i * W + j - 1] + gapPen, mat[(i - 1) * W + j] + gapPen, &k);
                tr[i * W + j] = k;
                if (mat[i * W + j] < 0) mat[i * W + j] = 0;
            }
        }
        for (i = n, j = m; mat[i * W + j] > 0 && rc == OR_OK;) {
            const uint8_t op = tr[i * W + j];
            rc = gsw_trace_step(&route, &len, &cap, &routeIdx, op);
            if (op == 0) { i--; j--; } else if (op == 1) j--; else i--;
        }
        *out_score = mat[n * W + m]; *out_i = i; *out_j = j;
    } else { /* RightDynamicAln */
        int64_t maxI = 0, maxJ = 0, currMax = curr_max_in;
        for (i = 0; i < n + 1; i++) {
            for (j = 0; j < m + 1; j++) {
                if (i == 0 && j == 0) mat[0] = 0;
                else if (i == 0) { mat[j] = mat[j - 1] + gapPen; tr[j] = 1; }
                else if (j == 0) { mat[i * W] = mat[(i - 1) * W] + gapPen; tr[i * W] = 2; }
                else {
                    mat[i * W + j] = tmt(mat[(i - 1) * W + j - 1] + sc[alpha[i - 1] * 5 + beta[j - 1]], mat[i * W + j - 1] + gapPen, mat[(i - 1) * W + j] + gapPen, &k);
                    tr[i * W + j] = k;
                }
                if (mat[i * W + j] > currMax) { currMax = mat[i * W + j]; maxI = i; maxJ = j; }
            }
        }
        for (i = maxI, j = maxJ; (i > 0 || j > 0) && rc == OR_OK;) {
            const uint8_t op = tr[i * W + j];
            rc = gsw_trace_step(&route, &len, &cap, &routeIdx, op);
            if (op == 0) { i--; j--; } else if (op == 1) j--; else i--;
        }
        *out_score = mat[maxI * W + maxJ]; *out_i = maxI; *out_j = maxJ;
    }
    free(mat); free(tr);
    if (rc != OR_OK) { free(route); return rc; }
    *out_route = route; *out_len = len;
    return OR_OK;
}

void or_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------------
 * Batch driver = the reference's house pattern for parallelism: a pool of workers pulling independent
 * items from a queue (genomeGraph/routines.go:12-65, align/affineGap_highMem.go:120-179), restated
 * with pthreads + an atomic work counter.  Used for (i) oracle batches in tests, (ii) bench.py's
 * cpu_baseline ("port") leg.
 *   mode 0 = AffineGap_customizeCheckersize   mode 1 = ConstGap_customizeCheckersize
 *   mode 2 = AffineGap_highMem                mode 3 = AffineGapLocal    mode 4 = ConstGap_highMem
 * Results: out_score[p]; cigars are concatenated per pair in out_ops with out_off[n_pairs+1].
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int mode; const int64_t *sc; int64_t gap_open, gap_extend, ci, cj;
    int64_t n_pairs; const uint8_t *a_cat; const int64_t *a_off; const uint8_t *b_cat; const int64_t *b_off;
    int64_t *out_score; or_cigar **ops; int64_t *nops; int *rcs;
    volatile int64_t next;
} batch_ctx;

static int run_one(batch_ctx *c, int64_t p) {
    const uint8_t *a = c->a_cat + c->a_off[p], *b = c->b_cat + c->b_off[p];
    int64_t n = c->a_off[p + 1] - c->a_off[p], m = c->b_off[p + 1] - c->b_off[p];
    switch (c->mode) {
    case 0: return or_affine_gap_checkersize(a, n, b, m, c->sc, c->gap_open, c->gap_extend, c->ci, c->cj, &c->out_score[p], &c->ops[p], &c->nops[p]);
    case 1: return or_const_gap_checkersize(a, n, b, m, c->sc, c->gap_open, c->ci, c->cj, &c->out_score[p], &c->ops[p], &c->nops[p]);
    case 2: return or_affine_gap_highmem(a, n, b, m, c->sc, c->gap_open, c->gap_extend, 0, &c->out_score[p], &c->ops[p], &c->nops[p]);
    case 3: return or_affine_gap_highmem(a, n, b, m, c->sc, c->gap_open, c->gap_extend, 1, &c->out_score[p], &c->ops[p], &c->nops[p]);
    case 4: return or_const_gap_highmem(a, n, b, m, c->sc, c->gap_open, &c->out_score[p], &c->ops[p], &c->nops[p]);
    default: return OR_EINVAL;
    }
}

static void *worker(void *arg) {
    batch_ctx *c = (batch_ctx *)arg;
    /* arena sized for the largest pair of the batch: 3 trace planes + prep arrays + rows, see the functions above */
    size_t need = 1 << 20;
    for (int64_t p = 0; p < c->n_pairs; p++) {
        const int64_t n = c->a_off[p + 1] - c->a_off[p], m = c->b_off[p + 1] - c->b_off[p];
        const int64_t ti = (c->mode <= 1) ? (n < c->ci ? n : c->ci) : n + 1, tj = (c->mode <= 1) ? (m < c->cj ? m : c->cj) : m + 1;
        const int64_t ni = (c->mode <= 1) ? n / c->ci + 1 : 0, nj = (c->mode <= 1) ? m / c->cj + 1 : 0;
        size_t b = (size_t)(3 * ti * tj) + (size_t)(3 * 8 * (ni * (m + 1) + nj * (n + 1))) + (size_t)(16 * 8 * (m + 1)) + (1 << 16);
        if (b > need) need = b;
    }
    if (need <= ((size_t)1 << 30)) { tl_base = (char *)malloc(need); tl_cap = tl_base ? need : 0; }
    for (;;) {
        int64_t p = __sync_fetch_and_add(&c->next, 1);
        if (p >= c->n_pairs) break;
        tl_used = 0; tl_on = (tl_base != NULL);
        c->rcs[p] = run_one(c, p);
        tl_on = 0;
        if (c->rcs[p] == 0 && in_arena(c->ops[p])) { /* the route lives in the arena: move it out */
            or_cigar *keep = (or_cigar *)malloc((size_t)(c->nops[p] > 0 ? c->nops[p] : 1) * sizeof(or_cigar));
            if (!keep) c->rcs[p] = OR_ENOMEM; else memcpy(keep, c->ops[p], (size_t)c->nops[p] * sizeof(or_cigar));
            c->ops[p] = keep;
        }
    }
    free(tl_base); tl_base = NULL; tl_cap = 0;
    return NULL;
}

int or_align_batch(int mode, const int64_t *sc, int64_t gap_open, int64_t gap_extend, int64_t ci, int64_t cj,
                   int64_t n_pairs, const uint8_t *a_cat, const int64_t *a_off, const uint8_t *b_cat, const int64_t *b_off,
                   int n_threads, int64_t *out_score, or_cigar **out_ops, int64_t *out_off /* n_pairs+1 */) {
    batch_ctx c; memset(&c, 0, sizeof(c));
    c.mode = mode; c.sc = sc; c.gap_open = gap_open; c.gap_extend = gap_extend; c.ci = ci; c.cj = cj;
    c.n_pairs = n_pairs; c.a_cat = a_cat; c.a_off = a_off; c.b_cat = b_cat; c.b_off = b_off; c.out_score = out_score;
    c.ops = (or_cigar **)calloc((size_t)(n_pairs > 0 ? n_pairs : 1), sizeof(or_cigar *));
    c.nops = (int64_t *)calloc((size_t)(n_pairs > 0 ? n_pairs : 1), 8);
    c.rcs = (int *)calloc((size_t)(n_pairs > 0 ? n_pairs : 1), sizeof(int));
    if (!c.ops || !c.nops || !c.rcs) { free(c.ops); free(c.nops); free(c.rcs); return OR_ENOMEM; }
    if (n_threads < 1) n_threads = 1;
    if (n_threads == 1) worker(&c);
    else {
        pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
        int started = 0;
        for (int t = 0; th && t < n_threads; t++) if (pthread_create(&th[t], NULL, worker, &c) == 0) started++; else break;
        if (started == 0) worker(&c);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
        free(th);
    }
    int rc = OR_OK; int64_t total = 0;
    for (int64_t p = 0; p < n_pairs; p++) { if (c.rcs[p] && !rc) rc = c.rcs[p]; total += c.nops[p]; }
    or_cigar *blob = NULL;
    if (!rc) {
        blob = (or_cigar *)calloc((size_t)(total > 0 ? total : 1), sizeof(or_cigar));
        if (!blob) rc = OR_ENOMEM;
    }
    if (!rc) {
        int64_t off = 0;
        for (int64_t p = 0; p < n_pairs; p++) {
            out_off[p] = off;
            if (c.nops[p]) memcpy(blob + off, c.ops[p], (size_t)c.nops[p] * sizeof(or_cigar));
            off += c.nops[p];
        }
        out_off[n_pairs] = off;
        *out_ops = blob;
    }
    for (int64_t p = 0; p < n_pairs; p++) free(c.ops[p]);
    free(c.ops); free(c.nops); free(c.rcs);
    return rc;
}
