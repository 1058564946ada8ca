// fpc_sweep.hip.h -- fast-path forward sweep for the CONSTANT-gap functions (align.ConstGap*, constGap.go:146-157): the geometry and the
// row blocks of fp_sweep.hip.h (8 lanes x RR rows per pair, 8 pairs per wave64, right-aligned rows, column checkpoints, planes of the
// rows n .. n-3, row blocks that follow each other through the row buffer) with the one-matrix recurrence.
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.5.
#pragma once
#include "fp_sweep.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// Keys 4*H' + tag with the REBASED value H'(i,j) = H(i,j) - g*(i+j) (fill_const_kernel's form): both gap moves cost nothing, every
// border (row 0, column 0) is 0, and a cell is
//     H'(i,j) = max3( H'(i-1,j-1) + (s - 2g),  H'(i,j-1),  H'(i-1,j) )          = add (SDWA, int16 profile) + max3
// -- 2 VALU instructions where the affine sweep needs 5.  Tags: the argmax of a cell IS its direction (diag 3, left 2, up 1: the
// tie order of tripleMaxTrace), a stateless walk needs nothing else; only the rows whose directions are kept (the planes of rows
// n .. n-3 in the bottom block) pay for them: or, add, and_or, and_or, max3, alignbit.  All other rows run on keys whose tag bits are
// junk < 4 (every increment is a multiple of 4, so a tag never changes the value of a max).
// Padding slots above row 1 reproduce row 0 (value 0: profile entry -32768, the diagonal candidate never wins; left = up = 0).
// Outputs (what fpc_walk_kernel and the window re-fills fill_const_kernel<.., WIN> consume): plain (un-rebased) column checkpoints
// 4*H(i,j) of every row every CKW columns (one int per row: tag bits junk), the direction planes of rows n .. n-3 (word = step >> 4,
// field = step & 15 with step = j + 7), h(n,m), and -- row blocks that hand their bottom row down -- H'(r, j) per column (one int).
// ROLE / `below` / piped / prog_*: as in fp_sweep_body.
// ------------------------------------------------------------------------------------------------------
template <int RR, int ROLE>
__device__ __forceinline__ void fpc_sweep_body(int *__restrict__ lds, const int wblk, const PairPlan *__restrict__ plans, int n_pairs,
                                               const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                               const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                               const KParams &kp, int *__restrict__ hcol, int *__restrict__ ckpt,
                                               unsigned *__restrict__ rowi, int *__restrict__ err,
                                               int *__restrict__ rowbuf, const int below, const bool piped, const int *prog_in, int *prog_out) {
    static_assert(RR <= 2 * FP8_LW && RR > FP_PLANES, "rows per lane");
    static_assert(ROLE < 2 || RR == 2 * FP8_LW, "the bottom and middle row blocks are full: 8 x 20 slots");
    constexpr bool BOTTOM = (ROLE == 0 || ROLE == 2);
    constexpr bool HANDS = (ROLE == 1 || ROLE == 3);
    constexpr bool TAKES = (ROLE >= 2);
    constexpr int BOT = G8 * 2 * FP8_LW;
    const int lane = threadIdx.x;
    const int g = lane >> 3;
    const int lp = (lane & 8) ? 15 - (lane & 15) : (lane & 7);
    const int G4 = kp.g4;
    if (lane < 25) lds[lane] = kp.sc4[lane] - 2 * G4;
    else if (lane < 32) lds[lane] = -32768; // padding rows: the diagonal candidate never wins
    int *prof = &lds[32 + g * FP8_PST];
    const char *prof_lane = reinterpret_cast<const char *>(prof + lp * FP8_LW);

    const int pbase = wblk * 8;
    int m_max = 0, m_min = 0x7fffffff;
    for (int q = 0; q < 8; q++) {
        const int mq = (pbase + q < n_pairs) ? plans[pbase + q].m : 0;
        m_max = max(m_max, mq); m_min = min(m_min, mq);
    }
    const int p = pbase + g;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const int row_base = TAKES ? pl.n - BOT * (below + 1) : 0;
    const int n_loc = ROLE == 1 ? pl.n - BOT * below : (TAKES ? BOT : pl.n);
    const uint8_t *ap = a_buf + (valid ? a_start[pl.src] + row_base : 0);
    const uint8_t *bp = b_buf + (valid ? b_start[pl.src] : 0);
    const int m_eff = valid ? pl.m : 0;
    const int P = G8 * RR - n_loc;
    const int q0 = lp * RR;
    int bad = 0;
    int cH; // the row-0 boundary (0) pinned in a VGPR: the DPP `old` operand of the first lane
    asm volatile("v_mov_b32 %0, 0" : "=v"(cH));

    { // int16 profile of this lane's rows: prof[b][lp][r] = 4*(scores[alpha[row]][b] - 2g), padding -32768
        int a5[2 * FP8_LW];
#pragma unroll
        for (int r = 0; r < 2 * FP8_LW; r++) {
            int a = 5;
            const int q = q0 + r;
            if (r < RR && q >= P) { a = ap[q - P]; if (a >= 5) { bad = 1; a = 4; } }
            a5[r] = a * 5;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 5; b++) {
#pragma unroll
            for (int k = 0; k < FP8_LW; k++) prof[b * FP8_BST + lp * FP8_LW + k] = (lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16);
        }
        __syncthreads();
    }
    int val[RR]; // H'(i, j-1) of this lane's rows: the left candidate of the next column
#pragma unroll
    for (int r = 0; r < RR; r++) val[r] = 0; // column 0 (and every padding row): 0
    unsigned accR[FP_PLANES] = {0u, 0u, 0u, 0u};
    int diag0 = 0;
    int h_out = 0, b_out = 0;
    int up_h = cH;
    auto base_of = [&](int c) {
        int b = 0;
        if (c >= 1 && c <= m_eff) { b = bp[c - 1]; if (b >= 5) { bad = 1; b = 4; } }
        return b * (FP8_BST * 4);
    };
    int qb = base_of(lp), nb = 0;

    const int level = pl.strips - 1 - below; // 0 = top block
    const int *rb_in = (TAKES && valid) ? rowbuf + pl.rowbuf_off + (int64_t)(level - 1) * (pl.m + 1) : nullptr;
    int *rb_out = (HANDS && valid) ? rowbuf + pl.rowbuf_off + (int64_t)level * (pl.m + 1) : nullptr;
    auto rb_at = [&](int c) -> int {
        if (!(TAKES && valid && c >= 1 && c <= m_eff)) return 0;
        return piped ? __hip_atomic_load(&rb_in[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rb_in[c];
    };
    int rq = 0, rqn = 0; // the first lane's queue of handed-down values (see fp_sweep_body)
    int *hand = lds + 32 + 8 * FP8_PST + g * 8;
    int rb_seen = 0;
    auto wait_cols = [&](int c) {
        if (TAKES && piped && rb_seen < c + G8 - 1) {
            const long long t_begin = wall_clock64();
            while ((rb_seen = rb_progress(prog_in)) < c + G8 - 1) {
                __builtin_amdgcn_s_sleep(32);
                if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); break; }
            }
        }
    };
    auto step = [&](const int t, auto chk, const bool ckflag) {
        constexpr bool CHECK = decltype(chk)::value;
        up_h = dpp_prev8(TAKES ? rq : up_h, h_out);
        if (TAKES) rq = dpp_next8(rq);
        const int pb = dpp_prev8(qb, b_out);
        qb = dpp_next8(qb);
        const int j = t - lp;
        b_out = pb;
        if (!CHECK || (j >= 1 && j <= m_eff)) {
            const int2 *pw = reinterpret_cast<const int2 *>(prof_lane + pb);
            int w[FP8_LW];
#pragma unroll
            for (int k = 0; k < FP8_LW / 2; k++) { const int2 v = pw[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
            int hd = diag0, upv = up_h;
#pragma unroll
            for (int r = 0; r < RR; r++) {
                const int S4 = (r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff);
                int hnew;
                if (!BOTTOM || r < RR - FP_PLANES) hnew = max3i(hd + S4, val[r], upv);
                else { // a plane row: the candidates carry their direction
                    hnew = max3i((hd | 3) + S4, (val[r] & ~3) | 2, (upv & ~3) | 1);
                    accR[RR - 1 - r] = alignbit2((unsigned)hnew, accR[RR - 1 - r]);
                }
                hd = val[r];
                val[r] = hnew;
                upv = hnew;
            }
            diag0 = up_h;
            h_out = upv;
            if (HANDS) { if (lp == G8 - 1) hand[t & 7] = h_out; }
            if (ckflag) {
            asm volatile("" ::: "memory");
            if ((j & (CKW - 1)) == 0 && j < m_eff && valid) { // column checkpoint: plain 4*H(i,j) (tag bits junk)
                int *ck = ckpt + pl.ckpt_off + (int64_t)(j / CKW - 1) * pl.n;
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    const int i = q0 + r - P + 1 + row_base; // row of the pair
                    if (i - row_base >= 1) ck[i - 1] = val[r] + G4 * (i + j);
                }
            }
            }
        }
    };

    const int Tend = ((m_max + G8 - 1) / 16 + 1) * 16;
    auto flush = [&](int t0) {
        if (BOTTOM && (t0 & 8) && lp == G8 - 1 && valid) {
            const int w = t0 >> 4;
            if (w < pl.words) {
                const int miss = (t0 + 7) - (m_eff + G8 - 1);
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
                for (int d = 0; d < FP_PLANES; d++) {
                    accR[d] >>= sh;
                    if (pl.n - d >= 1) rowi[pl.rowi_off + (int64_t)d * pl.words + w] = accR[d];
                }
            }
        }
    };
    auto hand_down = [&](int t0) {
        if (HANDS) {
            __syncthreads();
            const int c = t0 - (G8 - 1) + lp;
            const int v = hand[lp];
            if (valid && c >= 1 && c <= m_eff) {
                if (piped) __hip_atomic_store(&rb_out[c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else rb_out[c] = v;
            }
            __syncthreads();
            if (piped && ((t0 + 8) & 31) == 0) rb_publish(prog_out, t0 + 7, lane);
        }
    };
    auto edge_half_block = [&](int t0) {
        nb = base_of(t0 + 8 + lp);
        if (TAKES) { wait_cols(t0 + 15); rqn = rb_at(t0 + 8 + lp); }
#pragma unroll 1
        for (int u = 0; u < 8; u++) step(t0 + u, std::true_type{}, true);
        qb = nb;
        if (TAKES) rq = rqn;
        flush(t0);
        hand_down(t0);
    };
    int t0 = 0;
    if (TAKES) { wait_cols(7); rq = rb_at(lp); }
    for (; t0 < Tend && !(t0 >= 8 && t0 + 7 <= m_min); t0 += 8) edge_half_block(t0);
    for (; t0 + 7 <= m_min; t0 += 8) {
        nb = base_of(t0 + 8 + lp);
        if (TAKES) { wait_cols(t0 + 15); rqn = rb_at(t0 + 8 + lp); }
        const bool ckflag = (t0 & (CKW - 1)) == 0;
#pragma unroll
        for (int u = 0; u < 8; u++) step(t0 + u, std::false_type{}, ckflag);
        qb = nb;
        if (TAKES) rq = rqn;
        flush(t0);
        hand_down(t0);
    }
    for (; t0 < Tend; t0 += 8) edge_half_block(t0);
    if (HANDS && piped) rb_publish(prog_out, 0x7fffffff, lane);
    if (BOTTOM && lp == G8 - 1 && valid && m_eff >= 1) hcol[pl.hcol_off] = (val[RR - 1] & ~3) + G4 * (pl.n + m_eff); // plain 4*H(n, m)
    if (bad) atomicOr(err, 1);
}

// reads of one row block (n <= 8 * RR)
template <int RR>
__global__ __launch_bounds__(64) void fpc_sweep_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                       const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                       const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                       KParams kp, int *__restrict__ hcol, int *__restrict__ ckpt,
                                                       unsigned *__restrict__ rowi, int *__restrict__ err) {
    __shared__ int lds[32 + 8 * FP8_PST];
    fpc_sweep_body<RR, 0>(lds, (int)blockIdx.x, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, hcol, ckpt, rowi, err, nullptr, 0, false, nullptr, nullptr);
}

// reads of S >= 2 row blocks: levels of W waves, level-major (see fp_sweep_levels_kernel)
template <int RRTOP>
__global__ __launch_bounds__(64) void fpc_sweep_levels_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                              const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                              const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                              KParams kp, int *__restrict__ hcol, int *__restrict__ ckpt,
                                                              unsigned *__restrict__ rowi, int *__restrict__ err,
                                                              int *__restrict__ rowbuf, int S, int W, int level0, int piped, int *__restrict__ prog) {
    __shared__ int lds[32 + 8 * FP8_PST + 8 * 8];
    const int lv = (int)blockIdx.x / W, w = (int)blockIdx.x - lv * W, level = level0 + lv, below = S - 1 - level;
    int *po = prog + (int64_t)level * W + w;
    const int *pi = po - W;
    if (level == 0) fpc_sweep_body<RRTOP, 1>(lds, w, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, hcol, ckpt, rowi, err, rowbuf, below, piped != 0, nullptr, po);
    else if (below == 0) fpc_sweep_body<2 * FP8_LW, 2>(lds, w, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, hcol, ckpt, rowi, err, rowbuf, 0, piped != 0, pi, nullptr);
    else fpc_sweep_body<2 * FP8_LW, 3>(lds, w, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, hcol, ckpt, rowi, err, rowbuf, below, piped != 0, pi, po);
}

} // namespace
