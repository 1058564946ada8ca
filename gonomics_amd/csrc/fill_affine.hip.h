// fill_affine.hip.h -- affine-gap fill kernel (16 lanes x 10 rows per pair): full direction matrix, strips, window re-fills, SCORED variant
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.
#pragma once
#include "gnx_common.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// Affine fill.
//   LOCAL = free end gaps (AffineGapLocal, affineGap_highMem.go:188-210)
//   MULTI = some pair of the launch has more than one 160-row strip.  When the launch has too few pairs to fill the GPU with
//           one wave per 4 pairs (host decision: strip_map != nullptr), the strips of a pair run as separate workgroups,
//           pipelined: block (group of 4 pairs, strip s) reads the bottom row of strip s-1 from the row buffer once the block
//           before it has published it (strip_prog, every RB_PUB steps; rows and progress word are agent-scope atomics, see rb_store).  A 20 kb x 100 kb
//           pair is up to 125 concurrent waves instead of one.  A workgroup runs the strip of its block index -- and, before
//           it, every strip above it that nobody has claimed yet (claim_items, gnx_common.hip.h): nobody waits for work that has
//           not been taken, whatever the dispatch order; the 5 s timeout on the spin is a bug trap (error flag instead of a
//           hang), not part of the protocol.  Such launches store their direction
//           words non-temporally (full lines that the fill never reads back).
//   P16   = 4*score fits int16: the per-row score profile is stored as packed int16 pairs in LDS
//   HFORM = gapOpen <= 0 (every caller): with h = max3(M,I,D) the recurrences collapse to
//           rt = max(h+oe, I+e), dn = max(h+oe, D+e) with identical values AND identical argmax tags
//           (oe <= e makes the dropped candidate I+oe / D+oe never a strict winner; ties keep M > I > D).
//           The h-form runs on REBASED keys V' = V - e*(i+j) (i = row of the pair, j = column of this launch's window), like
//           fp_sweep_kernel: both extensions cost nothing, both opens share h' + o:
//               M' = (h'diag | 3) + (s - 2e)   I' = max(h' + o, I')   D' = max(h' + o, D')
//           i.e. per cell  or, add, and_or, and_or, max3, add, max, max  + 3 v_alignbit = 11 VALU instructions (13 with the
//           un-rebased h-form, 15 with the literal three-candidate form kept for gapOpen > 0).  Every max compares candidates
//           of one cell (same offset), so values and tags are the plain recurrence's; the row buffer carries V', hcol is
//           un-rebased when stored, checkpoints of the sweep are rebased when loaded.
// LDS (dwords): [0,32) 4*score table; then per pair g a profile  prof[b][lane][LW]  (b-stride BST, pair offset
// ProfCfg::pair_off(g)).  BST = 0 and the second pair's offset = 16 (mod 32) make the 32 lanes of a ds_read_b32 group hit 32 distinct banks whatever
// bases they look up (lane stride 5 or 10 dwords is odd/2*odd -> a permutation within a pair, +16 for the
// second pair of the group fills the complement).
// ------------------------------------------------------------------------------------------------------
template <bool P16> struct ProfCfg {
    static constexpr int LW = P16 ? R / 2 : R;       // dwords per lane per base
    // int32 entries: one pair per 160-dword plane, pair stride 5 planes + 16.  int16 entries (16 lanes x 5 dwords = 80 per plane): the
    // two pairs of a 32-lane group interleave plane by plane -- pair 0 in [0, 80) of a 160-dword plane, pair 1 in [80, 160), 80 == 16
    // (mod 32) -- so nothing is padding: 6 528 B per workgroup instead of 8 064, i.e. 6 instead of 7 of the 1280-byte granules LDS is
    // handed out in on gfx950: 21 workgroups per CU instead of 18 (the piped constant-gap sweep holds 5 waves per SIMD by registers).
    static constexpr int BST = 160;                  // dwords per base plane (P16: of a duo of pairs); multiple of 32
    static constexpr int TOTAL = P16 ? 2 * 5 * BST : 4 * (5 * BST + 16); // dwords of the four pairs of a workgroup
    static __device__ __forceinline__ int pair_off(int g) { return P16 ? (g >> 1) * (5 * BST) + (g & 1) * (G * (R / 2)) : g * (5 * BST + 16); }
};

template <bool LOCAL, bool MULTI, bool P16, bool HFORM, bool WIN = false, bool SCORED = false, bool XP = false>
__global__ __launch_bounds__(64) void fill_affine_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                         const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                         const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                         KParams kp, uint4 *__restrict__ trace, int *__restrict__ hcol,
                                                         int2 *__restrict__ rowbuf, unsigned *__restrict__ dcol, const int2 *__restrict__ ckpt,
                                                         int *__restrict__ err, const int *__restrict__ smat = nullptr,
                                                         const int2 *__restrict__ strip_map = nullptr, int *__restrict__ strip_prog = nullptr) {
    // SCORED: the substitution score of a cell comes from an explicit per-pair matrix in HBM (chunk / multiple-alignment
    //      variants, "next" row N1) instead of the LDS profile of alpha x the base of the column; sequences are not read.
    // WIN: window / tile re-fill of the fast path: the left boundary comes from a column checkpoint written by
    //      fp_sweep_kernel (pl.col_off > 0), the row-0 boundary and the beta window start at column col_off.
    //      The window holds ONE row block of the sweep: rows pl.s_off + 1 .. pl.s_off + pl.n of the pair (pl.s_off = "row base",
    //      pl.s_pitch = rows of the whole pair = pitch of its checkpoints).  With a row base the row above the window is what the
    //      block above handed down in the sweep's row buffer, rowbuf[pl.rowbuf_off + column of the pair] = {D'(base+1,j), h'(base,j)}
    //      with their argmax tags, rebased by the sweep with the pair's (i + j): + e*(row base + col_off) makes them this launch's.
    // XP:  window re-fill of the TRANSPOSED free-end-gap fast path (fp_sweep_kernel<.., true>): the gap chains swap their tags
    //      (tie order M >= D' >= I'), row 0 is free (I'(0,j) = 0) and so is the horizontal step in the last row.
    static_assert(!XP || (WIN && HFORM && !LOCAL && !MULTI && !SCORED), "XP is a window re-fill variant");
    constexpr int TI = XP ? 1 : 2, TD = XP ? 2 : 1;
    using PC = ProfCfg<P16>;
    constexpr int LW = PC::LW, BST = PC::BST, PTOT = PC::TOTAL;
    __shared__ int lds[32 + PTOT];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    if (lane < 25) lds[lane] = kp.sc4[lane] - (HFORM ? 2 * kp.e4 : 0); // rebased diagonal: s - 2e
    int *prof = &lds[32 + PC::pair_off(g)];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);

    const bool piped = MULTI && strip_map != nullptr; // else one wave walks the strips of its 4 pairs one after the other
    constexpr bool PF = !SCORED && !WIN; // the look-ahead form of the step (LDS reads one step ahead, base conversion deferred), see below
    // piped: this workgroup runs strip strip_map[blockIdx].y of its group -- and first every strip above it that nobody has claimed yet
    // (claim_items: forward progress without any assumption about dispatch order); none in the normal case
    int n_stolen = 0;
    if (piped) { n_stolen = claim_items(strip_prog + gridDim.x, 1, strip_map[blockIdx.x].y); if (n_stolen < 0) return; }
    const int pbase = (piped ? strip_map[blockIdx.x].x : (int)blockIdx.x) * 4;
    int S_max = 0, m_max = 0;
    for (int q = 0; q < 4; q++) {
        if (pbase + q < n_pairs) { S_max = max(S_max, plans[pbase + q].strips); m_max = max(m_max, plans[pbase + q].m); }
    }
    const int p = pbase + g;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const int row_base = WIN ? (int)pl.s_off : 0;                          // rows of the pair above this window
    const int ck_pitch = (WIN && pl.s_pitch) ? (int)pl.s_pitch : pl.n;    // rows of the whole pair
    const uint8_t *ap = SCORED ? nullptr : a_buf + (valid ? a_start[pl.src] + row_base : 0);
    BetaSrcT<WIN ? 2 : 0> bp; // (SCORED: unused; WIN: re-fills of the fast path, whose beta may be windows of the packed resident reference)
    bp.init(b_buf, kp, (valid && !SCORED) ? b_start[pl.src] + (WIN ? pl.col_off : 0) : 0, (valid && !SCORED) ? pl.m : 0);
    const int Tend = (m_max + 15 + 15) & ~15;
    const int OE4 = kp.oe4, E4 = kp.e4;
    constexpr bool REB = HFORM;   // rebased keys (see above)
    const int RB = REB ? E4 : 0;  // V' = V - RB*(i + j)
    int vOE4, vE4, vO4; // constants pinned in VGPRs
    asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=v"(vOE4), "=v"(vE4), "=v"(vO4) : "s"(kp.oe4), "s"(kp.e4), "s"(kp.o4));
    int bad = 0;

    const int s_own = piped ? strip_map[blockIdx.x].y : 0;
    const int s_lo = piped ? s_own - n_stolen : 0, s_hi = piped ? s_own + 1 : S_max;
    const int64_t rb_pitch = (int64_t)pl.m + 1; // row-buffer entries per strip of this pair
    for (int s = s_lo; s < s_hi; s++) {
        const int bid = (int)blockIdx.x - s_own + s; // piped: block index of strip s of this group = its slot in strip_prog
        const bool gact = valid && s < pl.strips;
        const int m_eff = gact ? pl.m : 0;
        int m_min = 0x7fffffff; // over the 4 pairs of the wave, this strip (wave-uniform)
        for (int q = 0; q < 4; q++) m_min = min(m_min, (pbase + q < n_pairs && s < plans[pbase + q].strips) ? plans[pbase + q].m : 0);
        const bool store_row = MULTI && gact && (s + 1 < pl.strips);
        const int row0 = s * H + l * R; // 0-based index of this lane's first row == 1-based index of the row above it
        const int r_last = (XP && row_base + pl.n == ck_pitch) ? pl.n - 1 - row0 : -1; // XP: slot of the pair's last row in this lane (if 0 <= r_last < R; windows of the bottom row block only)
        int rt[R], hold[R];
        unsigned acc[3 * R]; // direction accumulators: [0,R) M, [R,2R) I, [2R,3R) D
        if (!SCORED) { // score profile of this lane's rows: prof[b][lane][k]
            int a5[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i0 = row0 + r;
                int a = 0;
                if (gact && i0 < pl.n) { a = ap[i0]; if (a >= 5) { bad = 1; a = 4; } }
                a5[r] = a * 5;
            }
            __syncthreads(); // table visible; previous strip's profile no longer read
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) {
                    int v;
                    if (P16) v = (lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16);
                    else v = lds[a5[k] + b];
                    prof[b * BST + l * LW + k] = v;
                }
            }
            __syncthreads();
        }
        const int2 *ck0 = nullptr; // window re-fill: left boundary = column checkpoint col_off / CKW
        if (WIN && pl.col_off > 0) ck0 = ckpt + pl.ckpt_off + (int64_t)(pl.col_off / CKW - 1) * ck_pitch + row_base;
        static_assert(!WIN || HFORM, "the checkpoints are rebased keys");
        const int Kck = WIN ? RB * (row_base + pl.col_off) : 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = row0 + r + 1; // column-0 cell of row i: M = I = -inf, D = D00 + i*ecol
            const int D1c = kp.d00_4 + (i + row_base) * kp.ecol4 + TD - RB * i; // D(i,0) (i + row base = row of the pair), rebased with j = 0
            hold[r] = max3i(NEG4 + 3, NEG4 + TI, D1c);
            rt[r] = max3i(NEG4 + 3 + OE4, NEG4 + TI + E4, D1c + OE4) - RB;      // I(i,1) = D(i,0) + oe, rebased with j = 1
            if (XP && r == r_last) rt[r] = D1c - RB; // I'(n,1) = h(n,0), no penalty
            // checkpoint {I'(i,j+1), h'(i,j)} of the sweep, rebased with the pair's (i + j): + e*(row base + col_off) makes them this launch's
            if (WIN && ck0 && gact && i <= pl.n) { const int2 v = ck0[i - 1]; rt[r] = v.x + Kck; hold[r] = v.y + Kck; }
            acc[r] = 0; acc[R + r] = 0; acc[2 * R + r] = 0;
        }
        const int grow0 = row0 + row_base; // row of the pair above this lane's first row
        int diag0 = (grow0 == 0) ? max3i(3, kp.o4 + TI, kp.d00_4 + TD) : max3i(NEG4 + 3, NEG4 + TI, kp.d00_4 + grow0 * kp.ecol4 + TD - RB * row0);
        if (WIN && ck0) {
            if (grow0 == 0) diag0 = max3i(NEG4 + 3, (XP ? 0 : kp.o4 + pl.col_off * E4) + TI, NEG4 + TD); // h(0, col_off)
            else if (gact && row0 <= pl.n) diag0 = ck0[row0 - 1].y + Kck;                              // (row0 = 0: the checkpoint of the row above the window)
        }
        int dn_out = 0, h_out = 0, b_out = 0;
        int sq_dn = 0, sq_h = 0;
        // boundary queues (row above the strip + beta): lane u holds column t0+u+1 of the current 16-step block
        int qdn, qh, qb, ndn = 0, nh = 0, nb = 0;
        auto boundary = [&](int c, int &odn, int &oh, int &ob) {
            if (WIN && row_base > 0) { // the row the block above handed down in the sweep
                if (c >= 1 && c <= m_eff) {
                    const int2 v = rowbuf[pl.rowbuf_off + pl.col_off + c];
                    const int K = RB * (row_base + pl.col_off);
                    odn = v.x + K; oh = v.y + K;
                } else { odn = 0; oh = 0; }
            } else if (!MULTI || s == 0) {
                const int M3 = NEG4 + 3, I2 = (XP ? 0 : kp.o4 + ((WIN ? pl.col_off : 0) + c) * E4) + TI - RB * c, D1 = NEG4 + TD; // row 0: I(0,c) = gapOpen + c*gapExtend (XP: free), rebased with i = 0
                const int h0 = max3i(M3, I2, D1);
                odn = ((LOCAL && c == m_eff) ? h0 : max3i(M3 + OE4, I2 + OE4, D1 + E4)) - RB; // D(1,c): one row further down
                oh = h0;
            } else if (c >= 1 && c <= m_eff) {
                const int2 v = rb_load(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], piped);
                odn = v.x; oh = v.y; // rebased like everything else
            } else { odn = 0; oh = 0; }
            ob = (!SCORED && c >= 1 && c <= m_eff) ? bp.raw(c - 1) : 0; // PF: the RAW base -- base_off() turns it into the LDS offset where the queue is needed (no wait on the load here)
            if (!PF && !SCORED) { int b = (c >= 1 && c <= m_eff) ? bp.value(ob, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } ob = b * (BST * 4); }
        };
        auto base_off = [&](int raw, int c) { int b = (c >= 1 && c <= m_eff) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); }; // LDS byte offset of the base's profile plane (c: the column the raw base was loaded for)
        // wait until the block of strip s-1 (the previous block of the grid) has published the row-buffer columns <= cmax;
        // strip_prog holds the number of columns it has published so far (INT_MAX when it is done)
        int rb_seen = 0;
        auto wait_rows = [&](int cmax) {
            if (piped && s > 0 && rb_seen < cmax) {
                const long long t_begin = wall_clock64();
                while ((rb_seen = rb_progress(&strip_prog[bid - 1])) < cmax) {
                    __builtin_amdgcn_s_sleep(32);
                    if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); break; } // 5 s at 100 MHz
                }
            }
        };
        if (MULTI && !piped && s > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        wait_rows(G);
        boundary(l + 1, qdn, qh, qb);
        if (PF) qb = base_off(qb, l + 1);

        // LDS profile (!SCORED): the entries of a step are read ONE STEP AHEAD -- the base a lane needs at step t + 1 is the one its left
        // neighbour has at step t, so the DPP move and the reads for t + 1 are issued before the arithmetic of step t and land while it
        // runs (at the top of their own step they cost the wave an LDS round trip per step; see cl_sweep_kernel).
        // (PF = the un-windowed fill only: the window re-fills of the fast path are launches of a few ten thousand short waves whose time is
        // set by how many of them a SIMD holds -- the look-ahead's registers cost them 0.9 ms per headline step, 31.6 -> 32.5 ms)
        int wq[LW], pb_cur = 0;
        auto fetch = [&](int pbv, int *w) {
            const int *pw = reinterpret_cast<const int *>(prof_lane + pbv);
#pragma unroll
            for (int k = 0; k < LW; k++) w[k] = pw[k];
        };
        if (PF) {
            pb_cur = dpp_shr1(qb, b_out);
            qb = dpp_shl1(qb, qb);
            fetch(pb_cur, wq);
        }
        // one anti-diagonal step; CHECK=false is the steady state (every lane of the wave has a live column)
        // take / nqv (!SCORED): the base queue of the next 16-step block, taken over at the last step of this one
        auto step = [&](const int t, auto chk, const bool take, const int nqv) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_dn = dpp_shr1(qdn, dn_out);
            const int up_h = dpp_shr1(qh, h_out);
            qdn = dpp_shl1(qdn, qdn);
            qh = dpp_shl1(qh, qh);
            int wn[LW], pb_next = 0, pb = 0;
            if (PF) {
                if (take) qb = nqv;
                pb_next = dpp_shr1(qb, pb_cur);
                qb = dpp_shl1(qb, qb);
                fetch(pb_next, wn);
                asm volatile("" ::: "memory"); // the reads stay HERE, ahead of the arithmetic (the scheduler would sink them next to their first use)
            } else if (!SCORED) {
                pb = dpp_shr1(qb, b_out);
                qb = dpp_shl1(qb, qb);
                b_out = pb;
            }
            const int j = t - l;
            if (!CHECK || (j >= 1 && j <= m_eff)) {
                int w[LW];
                if (SCORED) {
                    // the lane's R rows of column j are contiguous in the score matrix (int32 entries, or int16 with P16: s_off, s_pitch and row0 are even)
                    const int *pw = P16 ? reinterpret_cast<const int *>(reinterpret_cast<const short *>(smat) + pl.s_off + (int64_t)(j - 1) * pl.s_pitch + row0)
                                        : smat + pl.s_off + (int64_t)(j - 1) * pl.s_pitch + row0;
#pragma unroll
                    for (int k = 0; k < LW; k++) w[k] = pw[k];
                } else if (PF) {
#pragma unroll
                    for (int k = 0; k < LW; k++) w[k] = wq[k];
                } else fetch(pb, w);
                int hd = diag0, dnu = up_dn;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int S4;
                    if (P16) S4 = (r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff);
                    else S4 = w[r];
                    acc[r] = alignbit2((unsigned)hd, acc[r]);
                    acc[R + r] = alignbit2((unsigned)rt[r], acc[R + r]);
                    acc[2 * R + r] = alignbit2((unsigned)dnu, acc[2 * R + r]);
                    int hnew, dnn;
                    if (HFORM) { // rebased keys
                        const int M3 = (hd | 3) + S4;              // S4 = 4*(s - 2e)
                        const int I2 = (rt[r] & ~3) | TI;
                        const int D1 = (dnu & ~3) | TD;
                        hnew = max3i(M3, I2, D1);
                        const int ho = hnew + vO4;                 // both opens
                        rt[r] = max(ho, I2);
                        dnn = max(ho, D1);
                        if (LOCAL) dnn = (j == m_eff) ? hnew - vE4 : dnn; // last column: D(i+1,m) = tmt(M,I,D)(i,m), no penalty (the -e is the rebase of the next row)
                        if (XP) rt[r] = (r == r_last) ? hnew - vE4 : rt[r]; // transposed: last row, I'(n,j+1) = tmt(M,I',D')(n,j)
                    } else {
                        const int M3 = (hd | 3) + S4;
                        const int I2 = (rt[r] & ~3) | 2;
                        const int D1 = (dnu & ~3) | 1;
                        hnew = max3i(M3, I2, D1);
                        const int Moe = M3 + vOE4;
                        rt[r] = max3i(Moe, I2 + vE4, D1 + vOE4);
                        dnn = max3i(Moe, I2 + vOE4, D1 + vE4);
                        if (LOCAL) dnn = (j == m_eff) ? hnew : dnn;
                    }
                    hd = hold[r];
                    hold[r] = hnew;
                    dnu = dnn;
                }
                diag0 = up_h;
                dn_out = dnu;
                h_out = hold[R - 1];
            }
            if (MULTI) { sq_dn = dpp_shl1(dn_out, sq_dn); sq_h = dpp_shl1(h_out, sq_h); }
            if (PF) {
#pragma unroll
                for (int k = 0; k < LW; k++) wq[k] = wn[k];
                pb_cur = pb_next;
            }
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            wait_rows(t0 + 2 * G);
            boundary(t0 + 16 + l + 1, ndn, nh, nb); // prefetch the next block's boundary
            const bool steady = t0 >= 16 && t0 + 16 <= m_min;
            if (steady && SCORED) { // the per-step HBM loads of the score matrix do better with the shallow unroll (3.2 -> 2.3 ms on tools/bench_n1.py)
#pragma unroll 2
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::false_type{}, false, 0);
            } else if (steady) {
#pragma unroll
                for (int u = 0; u < 16; u++) { if (PF && u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::false_type{}, u == 15, nb); }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (PF && u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, u == 15, nb); }
            }
            // consume the loads issued at the top of this block BEFORE the stores below are issued (exact wait, nothing newer in flight;
            // left to their first real use -- the DPP moves of the next block -- the wait becomes a vmcnt(0) behind those stores)
            if (PF) asm volatile("" :: "v"(ndn), "v"(nh));
            qdn = ndn; qh = nh;
            if (!PF) qb = nb;
            // flush 16 steps of direction bits: word w of this strip
            const int w = t0 >> 4;
            if (gact && w < pl.words) {
                const int miss = (t0 + 16 - l) - m_eff; // steps this lane sat idle after its last column
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
                if (t0 + 16 > m_min) { // drain: a lane that finished early right-aligns its last fields (it never shifts again)
#pragma unroll
                    for (int d = 0; d < 3 * R; d++) acc[d] >>= sh;
                }
                uint4 *dst = trace + pl.trace_off + ((int64_t)(s * pl.words + w) * QA) * G + l;
#pragma unroll
                for (int q = 0; q < QA - 1; q++) trace_store(&dst[q * G], acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], piped);
                trace_store(&dst[(QA - 1) * G], acc[4 * (QA - 1)], acc[4 * (QA - 1) + 1], 0u, 0u, piped);
            }
            if (store_row) {
                const int c = t0 + l - 14;
                if (c >= 1 && c <= m_eff) rb_store(&rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + c], sq_dn, sq_h, piped);
            }
            if (piped && ((t0 + 16) & (RB_PUB - 1)) == 0) rb_publish(&strip_prog[bid], t0 + 1, lane); // the bottom row of this strip is out up to column t0 + 1
        }
        if (gact && m_eff >= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) if (row0 + r < pl.n) hcol[pl.hcol_off + row0 + r] = hold[r] + RB * (row0 + r + 1 + m_eff); // plain h(i, m)
            // last-column D-plane fields of this lane's rows, packed (field r at bits 2r): lets the traceback skip
            // vertical runs in column m (free end gaps of AffineGapLocal, trailing gaps when alpha is the long one)
            const int t0f = ((m_eff + l - 1) >> 4) << 4, missf = t0f + 16 - l - m_eff; // 0..15: in-place drain shift
            unsigned dw = 0;
#pragma unroll
            for (int r = 0; r < R; r++) dw |= ((acc[2 * R + r] >> (30 - 2 * missf)) & 3u) << (2 * r);
            dcol[pl.dcol_off + s * G + l] = dw;
        }
        if (piped) rb_publish(&strip_prog[bid], 0x7fffffff, lane);
        else if (MULTI) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    if (bad) atomicOr(err, 1);
}

} // namespace
