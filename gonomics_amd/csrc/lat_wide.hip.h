// lat_wide.hip.h -- the int64 form of the latency geometry: the fallback for what the int32 keys of every other kernel cannot hold
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.10.
#pragma once
#include "lat_fill.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// The reference computes in int64 throughout (align/align.go:8, affineGap.go:151-207, constGap.go:129-176).  The kernels of this library
// hold 4 * score + tag in int32; what leaves that range is taken by the snapshot path on moving bases (REBASE, const_long.hip.h) -- for the
// global functions with gapOpen <= 0 and scores small enough that a strip's band fits int32 around its base.  Everything else that is out of
// range -- AffineGapLocal, gapOpen > 0, scores in the millions -- comes here: lat_fill_kernel's mapping (one pair per wave, 64 lanes x 2
// rows, strips piped through sentinel-marked row buffers) with int64 keys 4 * score + tag, the LITERAL three-candidate recurrences
// (affineGap_highMem.go:181-223, constGap_highMem.go:30-45: no rebasing, no h-form, any sign of gapOpen), the same direction words, so
// traceback_kernel<.., 64, 2> walks the matrix unchanged (it takes the score from an int64 array instead of hcol, which keeps the tags).
// A 64-bit max is a compare and two selects where the int32 kernels spend one v_max3: ~3 x the instructions per cell -- a correctness
// fallback, measured in profiles/r5_experiments.md.  Limit: the stored matrix (1 B per cell) must fit the workspace.
// ------------------------------------------------------------------------------------------------------
typedef long long k64;
constexpr k64 NEG64 = -(1LL << 60);          // "veryNegNum" for int64 keys: 4 * |score| stays below 2^59 (host check)
constexpr k64 SENT64 = (k64)0x8000000080000000ULL; // "not written yet" in the row buffer (hipMemsetD32 of LAT_SENT): far below NEG64, never a key

__device__ __forceinline__ k64 mk64(int lo, int hi) { return (k64)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo); }
__device__ __forceinline__ k64 wave_shr1_64(k64 oldv, k64 src) {
    return mk64(wave_shr1((int)oldv, (int)src), wave_shr1((int)(oldv >> 32), (int)(src >> 32)));
}
__device__ __forceinline__ k64 row_shl1_64(k64 oldv, k64 src) {
    return mk64(dpp_shl1((int)oldv, (int)src), dpp_shl1((int)(oldv >> 32), (int)(src >> 32)));
}
__device__ __forceinline__ k64 max3k(k64 a, k64 b, k64 c) { const k64 x = a > b ? a : b; return x > c ? x : c; }
__device__ __forceinline__ void wide_store(k64 *p, k64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ k64 wide_load(const k64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// rowbuf: k64 entries, AFFINE: {dn, h} per column (2 entries), constant gap: 1 entry; pl.rowbuf_off in entries
// SCORED (round 6: the chunk / multiple-alignment variants beyond the int32 range, align/affineGap_highMem.go:227-353 is int64 there too): the substitution score of
// a cell comes from the pair's explicit matrix in HBM (column-major int32 entries 4 * s, no bias: S[s_off + (j-1) s_pitch + (i-1)]), loaded sixteen steps ahead into a
// register ring exactly as lat_fill_kernel<.., SCORED> does; sequences are not read.
template <bool AFFINE, bool LOCAL, bool SCORED = false>
__global__ __launch_bounds__(64) void lat_wide_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                      const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                      const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                      KParams kp, long long o4w, long long e4w, long long d00w, long long ecolw,
                                                      uint4 *__restrict__ trace, int *__restrict__ hcol, int64_t *__restrict__ score64,
                                                      k64 *__restrict__ rowbuf, unsigned *__restrict__ dcol, int *__restrict__ err,
                                                      const int2 *__restrict__ strip_map, int *__restrict__ claims, const int *__restrict__ smat = nullptr) {
    static_assert(!SCORED || (AFFINE && !LOCAL), "the scored variants have AffineGap_highMem semantics");
    // o4w / e4w / d00w / ecolw: 4 * gapOpen (constant gap: 4 * gapPen), 4 * gapExtend, 4 * D(0,0), 4 * (column-0 extension) in int64; kp.sc4 = 4 * scores (int32: |score| <= 2^26)
    static_assert(AFFINE || !LOCAL, "free end gaps are an affine mode");
    constexpr int TI = 2, TD = 1;
    constexpr int BST = LG * LR;
    constexpr int NACC = AFFINE ? 3 * LR : LR;
    constexpr int Q = AFFINE ? LQA : LQC;
    constexpr int RBW = AFFINE ? 2 : 1;
    __shared__ int lds[32 + 5 * BST];
    const int l = threadIdx.x;
    if (l < 25) lds[l] = kp.sc4[l] + 3; // the diagonal candidate's tag rides on the profile entry
    int *prof = &lds[32];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LR);
    const int s_own = strip_map[blockIdx.x].y;
    const int n_stolen = claim_items(claims, 1, s_own);
    if (n_stolen < 0) return;
    const int p = strip_map[blockIdx.x].x;
    const PairPlan pl = plans[p];
    const uint8_t *ap = SCORED ? nullptr : a_buf + a_start[pl.src];
    BetaBytes bp;
    bp.init(b_buf, kp, SCORED ? 0 : b_start[pl.src], SCORED ? 0 : pl.m);
    const int m = pl.m;
    const int Tend = (m + (LG - 1) + 15) & ~15;
    const k64 O4 = o4w, E4 = e4w, OE4 = o4w + e4w;
    int bad = 0;
    const int64_t rb_pitch = ((int64_t)m + 1) * RBW;

    for (int s = s_own - n_stolen; s <= s_own; s++) {
        const bool store_row = s + 1 < pl.strips;
        const int row0 = s * LH + l * LR;
        if (!SCORED) {
            int a5[LR];
#pragma unroll
            for (int r = 0; r < LR; r++) {
                const int i0 = row0 + r;
                int a = 0;
                if (i0 < pl.n) { a = ap[i0]; if (a >= 5) { bad = 1; a = 4; } }
                a5[r] = a * 5;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LR; k++) prof[b * BST + l * LR + k] = lds[a5[k] + b];
            }
            __syncthreads();
        }
        k64 rt[LR], hold[LR];
        unsigned acc[NACC];
#pragma unroll
        for (int r = 0; r < LR; r++) {
            const k64 i = row0 + r + 1;
            if (AFFINE) {
                const k64 D1c = d00w + i * ecolw + TD; // D(i, 0)
                hold[r] = max3k(NEG64 + 3, NEG64 + TI, D1c);
                rt[r] = max3k(NEG64 + 3 + OE4, NEG64 + TI + E4, D1c + OE4); // I(i, 1)
            } else { hold[r] = i * O4; rt[r] = 0; } // column 0: i * gapPen
        }
#pragma unroll
        for (int d = 0; d < NACC; d++) acc[d] = 0;
        k64 diag0;
        if (AFFINE) diag0 = (row0 == 0) ? max3k(3, O4 + TI, d00w + TD) : max3k(NEG64 + 3, NEG64 + TI, d00w + (k64)row0 * ecolw + TD);
        else diag0 = (k64)row0 * O4;
        k64 dn_out = 0, h_out = 0, sq_dn = 0, sq_h = 0;
        int b_out = 0;
        k64 qdn = 0, qh = 0, ndn = 0, nh = 0;
        int qb = 0, nb = 0;
        auto row0_boundary = [&](int c, k64 &odn, k64 &oh) {
            if (AFFINE) {
                const k64 M3 = NEG64 + 3, I2 = O4 + (k64)c * E4 + TI, D1 = NEG64 + TD; // row 0: I(0, c) = gapOpen + c gapExtend
                const k64 h0 = max3k(M3, I2, D1);
                odn = (LOCAL && c == m) ? h0 : max3k(M3 + OE4, I2 + OE4, D1 + E4);
                oh = h0;
            } else { odn = (k64)c * O4; oh = 0; } // row 0: j * gapPen
        };
        auto issue = [&](int c, k64 &odn, k64 &oh, int &ob) {
            odn = 0; oh = 0; ob = 0;
            if (l < 16 && c >= 1 && c <= m) {
                if (s == 0) row0_boundary(c, odn, oh);
                else {
                    const k64 *src = &rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + (int64_t)c * RBW];
                    odn = wide_load(src); if (AFFINE) oh = wide_load(src + 1);
                }
                if (!SCORED) ob = bp.raw(c - 1);
            }
        };
        auto settle = [&](int c, k64 &odn, k64 &oh) {
            if (s > 0) {
                const bool mine = l < 16 && c >= 1 && c <= m;
                auto missing = [&]() { return mine && (odn == SENT64 || (AFFINE && oh == SENT64)); };
                if (__any(missing())) {
                    const long long t_begin = wall_clock64();
                    while (true) {
                        if (missing()) {
                            const k64 *src = &rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + (int64_t)c * RBW];
                            odn = wide_load(src); if (AFFINE) oh = wide_load(src + 1);
                        }
                        if (!__any(missing())) break;
                        __builtin_amdgcn_s_sleep(4);
                        if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); if (missing()) { odn = 0; oh = 0; } break; }
                    }
                }
            }
        };
        auto base_off = [&](int raw, int c) { if (SCORED) return 0; int b = (l < 16 && c >= 1 && c <= m) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); };
        // SCORED: the ring of score entries, slot u = the step at position u of a block (lat_fill.hip.h: no branch around the load, a lane outside its columns loads
        // the entry of the nearest one it has and never uses it)
        int ring[16][LR];
        auto ring_load = [&](int t, int u) {
            const int j = min(max(t - l, 1), m);
            const int2 v = *reinterpret_cast<const int2 *>(smat + pl.s_off + (int64_t)(j - 1) * pl.s_pitch + row0);
            ring[u][0] = v.x + 3; ring[u][1] = v.y + 3; // (the diagonal candidate's tag rides on the entry, like on the profile's)
        };
        static_assert(LR == 2, "the ring loads two adjacent rows as one 8-byte entry");
        if (SCORED) {
#pragma unroll
            for (int u = 0; u < 16; u++) ring_load(u + 1, u);
        }
        issue(l + 1, qdn, qh, qb);
        settle(l + 1, qdn, qh);
        qb = base_off(qb, l + 1);
        int wq[LR], pb_cur;
        auto fetch = [&](int pbv, int *w) {
            const int *pw = reinterpret_cast<const int *>(prof_lane + pbv);
#pragma unroll
            for (int k = 0; k < LR; k++) w[k] = pw[k];
        };
        pb_cur = 0;
        if (!SCORED) {
            pb_cur = wave_shr1(qb, b_out);
            qb = dpp_shl1(qb, qb);
            fetch(pb_cur, wq);
        }
        auto step = [&](const int t, const bool take, const int nqv, const int u = 0) {
            const k64 up_dn = wave_shr1_64(qdn, dn_out);
            const k64 up_h = AFFINE ? wave_shr1_64(qh, h_out) : 0;
            qdn = row_shl1_64(qdn, qdn);
            if (AFFINE) qh = row_shl1_64(qh, qh);
            int wn[LR], pb_next = 0;
            if (!SCORED) {
                if (take) qb = nqv;
                pb_next = wave_shr1(qb, pb_cur);
                qb = dpp_shl1(qb, qb);
                fetch(pb_next, wn);
            } else {
#pragma unroll
                for (int k = 0; k < LR; k++) { wq[k] = ring[u][k]; wn[k] = 0; }
            }
            const int j = t - l;
            if (j >= 1 && j <= m) {
                if (AFFINE) {
                    k64 hd = diag0, dnu = up_dn;
#pragma unroll
                    for (int r = 0; r < LR; r++) {
                        acc[r] = alignbit2((unsigned)hd, acc[r]);
                        acc[LR + r] = alignbit2((unsigned)rt[r], acc[LR + r]);
                        acc[2 * LR + r] = alignbit2((unsigned)dnu, acc[2 * LR + r]);
                        const k64 M3 = (hd & ~3LL) + (k64)wq[r]; // profile entry = 4 * s + 3
                        const k64 I2 = (rt[r] & ~3LL) | TI;
                        const k64 D1 = (dnu & ~3LL) | TD;
                        const k64 hnew = max3k(M3, I2, D1);
                        const k64 Moe = M3 + OE4;
                        rt[r] = max3k(Moe, I2 + E4, D1 + OE4);
                        k64 dnn = max3k(Moe, I2 + OE4, D1 + E4);
                        if (LOCAL) dnn = (j == m) ? hnew : dnn;
                        hd = hold[r];
                        hold[r] = hnew;
                        dnu = dnn;
                    }
                    diag0 = up_h;
                    dn_out = dnu;
                    h_out = hold[LR - 1];
                } else {
                    k64 vd = diag0, vu = up_dn;
#pragma unroll
                    for (int r = 0; r < LR; r++) {
                        const k64 k = max3k(vd + (k64)wq[r], hold[r] + O4 + 2, vu + O4 + 1);
                        acc[r] = alignbit2((unsigned)k, acc[r]);
                        vd = hold[r];
                        hold[r] = k & ~3LL;
                        vu = hold[r];
                    }
                    diag0 = up_dn;
                    dn_out = vu;
                }
            }
            sq_dn = row_shl1_64(dn_out, sq_dn);
            if (AFFINE) sq_h = row_shl1_64(h_out, sq_h);
            if (!SCORED) {
#pragma unroll
                for (int k = 0; k < LR; k++) wq[k] = wn[k];
                pb_cur = pb_next;
            }
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            issue(t0 + 16 + l + 1, ndn, nh, nb);
            if (SCORED) { // (the ring is indexed by the position in the block: unrolled; the slot of step u is re-loaded right after step u has used it)
#pragma unroll
                for (int u = 0; u < 16; u++) { step(t0 + u + 1, false, 0, u); ring_load(t0 + 16 + u + 1, u); }
            } else {
#pragma unroll 4
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, u == 15, nb); }
            }
            settle(t0 + 16 + l + 1, ndn, nh);
            qdn = ndn; qh = nh;
            const int w = t0 >> 4;
            if (w < pl.words) {
                const int miss = (t0 + 16 - l) - m;
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
                if (t0 + 16 > m) {
#pragma unroll
                    for (int d = 0; d < NACC; d++) acc[d] >>= sh;
                }
                uint4 *dst = trace + pl.trace_off + ((int64_t)(s * pl.words + w) * Q) * LG + l;
                if (AFFINE) {
                    trace_store(&dst[0], acc[0], acc[1], acc[2], acc[3], true);
                    trace_store(&dst[LG], acc[4], acc[5], 0u, 0u, true);
                } else trace_store(&dst[0], acc[0], acc[1], 0u, 0u, true);
            }
            if (store_row) {
                const int c = t0 + (l - (LG - 16)) + 1 - (LG - 1);
                if (l >= LG - 16 && c >= 1 && c <= m) {
                    k64 *dst = &rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + (int64_t)c * RBW];
                    wide_store(dst, sq_dn); if (AFFINE) wide_store(dst + 1, sq_h);
                }
            }
        }
        if (m >= 1) {
#pragma unroll
            for (int r = 0; r < LR; r++) if (row0 + r < pl.n) {
                hcol[pl.hcol_off + row0 + r] = (int)(hold[r] & 3); // the argmax tag of h(i, m); the score of (n, m) goes out in int64
                if (row0 + r + 1 == pl.n) score64[p] = hold[r] >> 2;
            }
            const int t0f = ((m + l - 1) >> 4) << 4, missf = t0f + 16 - l - m;
            unsigned dw = 0;
#pragma unroll
            for (int r = 0; r < LR; r++) dw |= ((acc[(AFFINE ? 2 * LR : 0) + r] >> (30 - 2 * missf)) & 3u) << (2 * r);
            dcol[pl.dcol_off + s * LG + l] = dw;
        }
    }
    if (bad) atomicOr(err, 1);
}

// the scores of the wide path leave in int64 (traceback_kernel reads its start state from the tags in hcol and writes tag >> 2 as "score")
__global__ __launch_bounds__(256) void wide_scores_kernel(const PairPlan *__restrict__ plans, const int64_t *__restrict__ score64, int64_t *__restrict__ score_out, int n) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && plans[p].n > 0 && plans[p].m > 0) score_out[p] = score64[p]; // (an empty sequence: traceback_kernel's closed form stays)
}

} // namespace
