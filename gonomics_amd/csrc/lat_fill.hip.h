// lat_fill.hip.h -- the LATENCY geometry of the general path: one pair per wave, 64 lanes x 2 rows, for launches of few long pairs
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.9.
#pragma once
#include "fill_affine.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// What a single align.AffineGap / align.ConstGap call is (cmd/cigarToBed/cigarToBed.go:86, cmd/globalAlignment/globalAlignment.go:84,
// every round of align/multiAlign.go:70-78): ONE pair, or a handful.  fill_affine_kernel / fill_const_kernel map a pair to 16 lanes x 10
// rows, four pairs per wave -- built for batches: a launch of one pair leaves 48 of 64 lanes idle, and its time is (columns + strips x
// hand-over lag) x the issue time of ~126 instructions per step of ONE wave per strip (9 673 x 10 000: 6.4 ms).  A lone wave is paced by
// its own instruction stream (~7 cycles per dependent VALU instruction), so the lever is instructions per step:
//   lat_fill_kernel   the same recording recurrences (rebased h-form; constant gap: rebased one-matrix form), same keys and tags, with
//                     the whole wave on ONE pair: lane l owns rows 2 l + 1, 2 l + 2 of a 128-row strip, the lane-to-lane moves are
//                     `wave_shr:1` DPP moves (gfx9 has them across the full wave), ~40 instructions per step instead of 126.
//   Strips of a pair  run as separate workgroups (one wave each), pipelined through the row buffer -- claimed like every piped launch
//                     (claim_items) -- but WITHOUT a progress word: the row buffer is preset to a sentinel (INT_MIN, never a key), the
//                     producer's 64-bit write-through stores ARE the signal, the consumer re-loads the 16 columns of its next block
//                     until none is the sentinel.  No s_waitcnt vmcnt(0) + publish on the producer side: a lone wave would pay for each
//                     with a full memory round trip.
//   Direction matrix  the layout of the general path with (lanes, rows) = (64, 2): per strip, 16-step word, chunk q and lane one
//                     uint4 -- 6 / 2 bits per cell, every 1 KB line written once; traceback_kernel<.., 64, 2> walks it (quirks Q1 / Q2,
//                     run merging: unchanged code, the geometry is a template parameter of load_word).
// Routing (run_device): batches whose 128-row strips number at most ~2 per SIMD of the device, inside the static int32 key range.
// ------------------------------------------------------------------------------------------------------
constexpr int LG = 64;                 // lanes per pair
constexpr int LR = 2;                  // rows per lane
constexpr int LH = LG * LR;            // rows per strip
__host__ __device__ constexpr int trace_q(bool affine, int rows) { return ((affine ? 3 : 1) * rows + 3) / 4; } // uint4 per lane and 16-step word
constexpr int LQA = trace_q(true, LR), LQC = trace_q(false, LR);
static_assert(trace_q(true, R) == QA && trace_q(false, R) == QC, "the general path's layout is the same formula");
constexpr int LAT_SENT = (int)0x80000000; // "not written yet" in the row buffer: below every key (finite keys > -2^29, the sentinel NEG4 = -2^30)

#define DPP_WAVE_SHR1 0x138
__device__ __forceinline__ int wave_shr1(int oldv, int src) { return __builtin_amdgcn_update_dpp(oldv, src, DPP_WAVE_SHR1, 0xf, 0xf, false); }

// SCORED (chunk / multiple-alignment variants, "next" row N1): the substitution score of a cell comes from the pair's explicit matrix in
// HBM (column-major, S[s_off + (j-1) s_pitch + (i-1)], int32 or -- S16 -- int16 entries carrying the -2e of the rebased diagonal move)
// instead of the LDS profile; sequences are not read.  A lone wave cannot hide a memory round trip behind other waves, so the two entries
// of a step are loaded SIXTEEN steps ahead into a register ring: the slot of step u is re-loaded right after step u has used it.
template <bool AFFINE, bool LOCAL, bool SCORED = false, bool S16 = false>
__global__ __launch_bounds__(64) void lat_fill_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                      const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                      const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                      KParams kp, uint4 *__restrict__ trace, int *__restrict__ hcol,
                                                      int2 *__restrict__ rowbuf, unsigned *__restrict__ dcol, int *__restrict__ err,
                                                      const int2 *__restrict__ strip_map, int *__restrict__ claims, const int *__restrict__ smat = nullptr) {
    static_assert(AFFINE || !LOCAL, "free end gaps are an affine mode");
    static_assert(!SCORED || (AFFINE && !LOCAL), "the scored variants have AffineGap_highMem semantics");
    static_assert(!S16 || SCORED, "S16 is a layout of the score matrix");
    constexpr int TI = 2, TD = 1;
    constexpr int BST = LG * LR; // dwords per base plane of the profile
    constexpr int NACC = AFFINE ? 3 * LR : LR;
    constexpr int Q = AFFINE ? LQA : LQC;
    __shared__ int lds[32 + 5 * BST];
    const int l = threadIdx.x;
    if (l < 25) lds[l] = AFFINE ? kp.sc4[l] - 2 * kp.e4 : kp.sc4[l] - 2 * kp.g4 + 1; // rebased diagonal move (constant gap: pre-tagged, see fill_const_kernel)
    int *prof = &lds[32];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LR);
    // this workgroup runs strip strip_map[blockIdx].y of pair strip_map[blockIdx].x -- and first every strip above it nobody has claimed (claim_items)
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
    const int E4 = kp.e4, OE4 = kp.oe4, RB = AFFINE ? kp.e4 : kp.g4;
    int vO4, vE4;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(vO4), "=v"(vE4) : "s"(kp.o4), "s"(kp.e4));
    int bad = 0;
    const int64_t rb_pitch = (int64_t)m + 1;

    for (int s = s_own - n_stolen; s <= s_own; s++) {
        const bool store_row = s + 1 < pl.strips;
        const int row0 = s * LH + l * LR; // 0-based index of this lane's first row == 1-based index of the row above it
        if (!SCORED) {
            int a5[LR];
#pragma unroll
            for (int r = 0; r < LR; r++) {
                const int i0 = row0 + r;
                int a = 0;
                if (i0 < pl.n) { a = ap[i0]; if (a >= 5) { bad = 1; a = 4; } }
                a5[r] = a * 5;
            }
            __syncthreads(); // table visible; the previous strip's profile no longer read
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LR; k++) prof[b * BST + l * LR + k] = lds[a5[k] + b];
            }
            __syncthreads();
        }
        // state: affine rt = I'(i, j+1), hold = h'(i, j) (keys with argmax tags, rebased V' = V - e (i + j)); constant gap hold = V' (tag 2)
        int rt[LR], hold[LR];
        unsigned acc[NACC];
#pragma unroll
        for (int r = 0; r < LR; r++) {
            if (AFFINE) {
                const int i = row0 + r + 1;
                const int D1c = kp.d00_4 + i * kp.ecol4 + TD - RB * i; // D(i, 0), rebased with j = 0
                hold[r] = max3i(NEG4 + 3, NEG4 + TI, D1c);
                rt[r] = max3i(NEG4 + 3 + OE4, NEG4 + TI + E4, D1c + OE4) - RB;
            } else { hold[r] = 2; rt[r] = 0; }
        }
#pragma unroll
        for (int d = 0; d < NACC; d++) acc[d] = 0;
        int diag0;
        if (AFFINE) diag0 = (row0 == 0) ? max3i(3, kp.o4 + TI, kp.d00_4 + TD) : max3i(NEG4 + 3, NEG4 + TI, kp.d00_4 + row0 * kp.ecol4 + TD - RB * row0);
        else diag0 = 2;
        int dn_out = 0, h_out = 0, b_out = 0, sq_dn = 0, sq_h = 0;
        // boundary queues: lanes 0 .. 15 hold columns t0 + l + 1 of the current 16-step block (row above the strip + beta); lane 0 consumes
        int qdn = 0, qh = 0, qb = 0, ndn = 0, nh = 0, nb = 0;
        auto row0_boundary = [&](int c, int &odn, int &oh) {
            if (AFFINE) {
                const int M3 = NEG4 + 3, I2 = kp.o4 + TI, D1 = NEG4 + TD; // row 0: I(0, c) = gapOpen + c gapExtend, rebased with i = 0
                const int h0 = max3i(M3, I2, D1);
                odn = ((LOCAL && c == m) ? h0 : max3i(M3 + OE4, I2 + OE4, D1 + E4)) - RB;
                oh = h0;
            } else { odn = 2; oh = 0; } // row 0, rebased: 0 (tag 2)
        };
        auto issue = [&](int c, int &odn, int &oh, int &ob) { // the loads of column c (lanes 0 .. 15), not waited for
            odn = 0; oh = 0; ob = 0;
            if (l < 16 && c >= 1 && c <= m) {
                if (s == 0) row0_boundary(c, odn, oh);
                else { const int2 v = rb_load(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], true); odn = v.x; oh = v.y; }
                if (!SCORED) ob = bp.raw(c - 1);
            }
        };
        auto settle = [&](int c, int &odn, int &oh) { // ... until the strip above has written them
            if (s > 0) {
                const bool mine = l < 16 && c >= 1 && c <= m;
                if (__any(mine && odn == LAT_SENT)) {
                    const long long t_begin = wall_clock64();
                    while (true) {
                        if (mine && odn == LAT_SENT) { const int2 v = rb_load(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], true); odn = v.x; oh = v.y; }
                        if (!__any(mine && odn == LAT_SENT)) break;
                        __builtin_amdgcn_s_sleep(4);
                        if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); if (mine && odn == LAT_SENT) { odn = 0; oh = 0; } break; } // 5 s: a bug trap, not part of the protocol
                    }
                }
            }
        };
        auto base_off = [&](int raw, int c) { if (SCORED) return 0; int b = (l < 16 && c >= 1 && c <= m) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); };
        // SCORED: the ring of score entries, slot u = the step at position u of a block; the two rows of a lane are adjacent in a column of the matrix
        int ring[16][LR];
        // (no branch around the load: a lane outside its columns loads the entry of the nearest one it has and never uses it -- behind a
        // branch the compiler cannot count the loads in flight and waits for ALL of them, i.e. a memory round trip per step)
        auto ring_load = [&](int t, int u) { // entries of step t (column t - l) into slot u
            const int j = min(max(t - l, 1), m);
            if (S16) {
                const int v = *reinterpret_cast<const int *>(reinterpret_cast<const short *>(smat) + pl.s_off + (int64_t)(j - 1) * pl.s_pitch + row0);
                ring[u][0] = (int)(short)(v & 0xffff); ring[u][1] = v >> 16;
            } else {
                const int2 v = *reinterpret_cast<const int2 *>(smat + pl.s_off + (int64_t)(j - 1) * pl.s_pitch + row0);
                ring[u][0] = v.x; ring[u][1] = v.y;
            }
        };
        static_assert(LR == 2, "the ring loads two adjacent rows as one 8-byte (S16: 4-byte) entry");
        if (SCORED) {
#pragma unroll
            for (int u = 0; u < 16; u++) ring_load(u + 1, u);
        }
        issue(l + 1, qdn, qh, qb);
        settle(l + 1, qdn, qh);
        qb = base_off(qb, l + 1);
        // profile entries one step ahead (the base a lane needs at step t + 1 is the one its left neighbour has at step t)
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
        auto step = [&](const int t, auto chk, const bool take, const int nqv, const int u = 0) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_dn = wave_shr1(qdn, dn_out);
            const int up_h = AFFINE ? wave_shr1(qh, h_out) : 0;
            qdn = dpp_shl1(qdn, qdn);
            if (AFFINE) qh = dpp_shl1(qh, qh);
            int wn[LR], pb_next = 0;
            if (!SCORED) {
                if (take) qb = nqv; // (last step of a block: the base queue of the next one takes over)
                pb_next = wave_shr1(qb, pb_cur);
                qb = dpp_shl1(qb, qb);
                fetch(pb_next, wn);
                asm volatile("" ::: "memory"); // the reads stay HERE, ahead of the arithmetic
            } else {
#pragma unroll
                for (int k = 0; k < LR; k++) wq[k] = ring[u][k];
            }
            const int j = t - l;
            if (SCORED) { // straight-line: in the ramps (CHECK) a lane outside its columns computes and keeps its old state by selects
                const bool act = !CHECK || (j >= 1 && j <= m);
                int hd = diag0, dnu = up_dn;
#pragma unroll
                for (int r = 0; r < LR; r++) {
                    const int S4 = wq[r];
                    const unsigned a0 = alignbit2((unsigned)hd, acc[r]), a1 = alignbit2((unsigned)rt[r], acc[LR + r]), a2 = alignbit2((unsigned)dnu, acc[2 * LR + r]);
                    const int M3 = (hd | 3) + S4;
                    const int I2 = (rt[r] & ~3) | TI;
                    const int D1 = (dnu & ~3) | TD;
                    const int hnew = max3i(M3, I2, D1);
                    const int ho = hnew + vO4;
                    const int rtn = max(ho, I2), dnn = max(ho, D1);
                    hd = hold[r];
                    acc[r] = act ? a0 : acc[r]; acc[LR + r] = act ? a1 : acc[LR + r]; acc[2 * LR + r] = act ? a2 : acc[2 * LR + r];
                    rt[r] = act ? rtn : rt[r];
                    hold[r] = act ? hnew : hold[r];
                    dnu = dnn;
                }
                diag0 = act ? up_h : diag0;
                dn_out = act ? dnu : dn_out;
                h_out = act ? hold[LR - 1] : h_out;
            } else if (!CHECK || (j >= 1 && j <= m)) {
                if (AFFINE) {
                    int hd = diag0, dnu = up_dn;
#pragma unroll
                    for (int r = 0; r < LR; r++) {
                        const int S4 = wq[r];
                        acc[r] = alignbit2((unsigned)hd, acc[r]);
                        acc[LR + r] = alignbit2((unsigned)rt[r], acc[LR + r]);
                        acc[2 * LR + r] = alignbit2((unsigned)dnu, acc[2 * LR + r]);
                        const int M3 = (hd | 3) + S4;
                        const int I2 = (rt[r] & ~3) | TI;
                        const int D1 = (dnu & ~3) | TD;
                        const int hnew = max3i(M3, I2, D1);
                        const int ho = hnew + vO4;
                        rt[r] = max(ho, I2);
                        int dnn = max(ho, D1);
                        if (LOCAL) dnn = (j == m) ? hnew - vE4 : dnn; // last column: D(i+1, m) = tmt(M, I, D)(i, m), no penalty
                        hd = hold[r];
                        hold[r] = hnew;
                        dnu = dnn;
                    }
                    diag0 = up_h;
                    dn_out = dnu;
                    h_out = hold[LR - 1];
                } else {
                    int vd = diag0, vu = up_dn;
#pragma unroll
                    for (int r = 0; r < LR; r++) {
                        const int k = max3i(vd + wq[r], hold[r], vu - 1);
                        acc[r] = alignbit2((unsigned)k, acc[r]);
                        vd = hold[r];
                        hold[r] = (k & ~3) | 2;
                        vu = hold[r];
                    }
                    diag0 = up_dn;
                    dn_out = vu;
                }
            }
            sq_dn = dpp_shl1(dn_out, sq_dn); // (row 3 of the wave: lane 63 inserts, lanes 48 .. 63 hold the last 16 columns of the bottom row)
            if (AFFINE) sq_h = dpp_shl1(h_out, sq_h);
            if (!SCORED) {
#pragma unroll
                for (int k = 0; k < LR; k++) wq[k] = wn[k];
                pb_cur = pb_next;
            } else ring_load(t + 16, u); // this slot's next use is sixteen steps away
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            issue(t0 + 16 + l + 1, ndn, nh, nb);
            if (SCORED) { // (the ring is indexed by the position in the block, so the block is unrolled in the ramps too)
                if (t0 >= LG && t0 + 16 <= m) {
#pragma unroll
                    for (int u = 0; u < 16; u++) step(t0 + u + 1, std::false_type{}, false, 0, u);
                } else {
#pragma unroll
                    for (int u = 0; u < 16; u++) step(t0 + u + 1, std::true_type{}, false, 0, u);
                }
            } else if (t0 >= LG && t0 + 16 <= m) {
#pragma unroll
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::false_type{}, u == 15, nb); }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, u == 15, nb); }
            }
            settle(t0 + 16 + l + 1, ndn, nh);
            qdn = ndn; qh = nh;
            // flush 16 steps of direction bits: word t0 / 16 of this strip
            const int w = t0 >> 4;
            if (w < pl.words) {
                const int miss = (t0 + 16 - l) - m; // steps this lane sat idle after its last column
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
                if (t0 + 16 > m) { // drain: a lane that finished early right-aligns its last fields (it never shifts again)
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
                const int c = t0 + (l - (LG - 16)) + 1 - (LG - 1); // lanes 48 .. 63: slot x = l - 48 holds what lane 63 handed down at step t0 + 1 + x, i.e. column t0 + 1 + x - 63
                if (l >= LG - 16 && c >= 1 && c <= m) rb_store(&rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + c], sq_dn, AFFINE ? sq_h : 0, true);
            }
        }
        if (m >= 1) {
#pragma unroll
            for (int r = 0; r < LR; r++) if (row0 + r < pl.n) hcol[pl.hcol_off + row0 + r] = AFFINE ? hold[r] + RB * (row0 + r + 1 + m) : (hold[r] & ~3) + RB * (row0 + r + 1 + m); // plain h(i, m)
            // last-column fields of this lane's rows (D plane / the only plane), packed: vertical runs in column m (traceback_kernel)
            const int t0f = ((m + l - 1) >> 4) << 4, missf = t0f + 16 - l - m;
            unsigned dw = 0;
#pragma unroll
            for (int r = 0; r < LR; r++) dw |= ((acc[(AFFINE ? 2 * LR : 0) + r] >> (30 - 2 * missf)) & 3u) << (2 * r);
            dcol[pl.dcol_off + s * LG + l] = dw;
        }
    }
    if (bad) atomicOr(err, 1);
}

} // namespace
