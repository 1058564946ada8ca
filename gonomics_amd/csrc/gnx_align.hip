// gnx_align.hip -- MI355X (gfx950 / CDNA4) implementation of the gonomics `align` pairwise DP hot path.
//
// What is computed (reference semantics, bit-exact):
//   align/affineGap.go:59-344, align/constGap.go:13-311, align/affineGap_highMem.go:57-223,
//   align/constGap_highMem.go:11-67, align/align.go:76-90 (tripleMaxTrace tie order M >= I >= D).
//
// How (MI355X-first, nothing here is a translation of the Go loops):
//   * FILL kernel: one 16-lane DPP row per pair (4 pairs per wave64), each lane owns R consecutive DP rows
//     (alpha), the wave sweeps the columns (beta) as an anti-diagonal wavefront: lane l works on column
//     t-l at step t.  The only cross-lane traffic is two `row_shr:1` DPP moves per step.  Cells are int32
//     "keys" = 4*score + tag (tag 3/2/1 = came-from M/I/D), so v_max3_i32 returns value AND argmax with the
//     reference's tie order, and the 2-bit direction is the low bits of the winner.  Per cell:
//       h  = max3(M,I,D)            (feeds M of the lower-right neighbour and is the cell's argmax)
//       rt = max3(M+oe, I+e, D+oe)  (feeds I of the right neighbour)
//       dn = max3(M+oe, I+oe, D+e)  (feeds D of the lower neighbour)
//     Direction bits are shifted into 3 accumulators per row with v_alignbit and flushed every 16 steps
//     as 8 coalesced 16-byte stores per lane (6 bits/cell of HBM write traffic, the algorithmic minimum).
//     Sequences longer than 16*R rows are processed as strips with a row buffer in HBM in between.
//   * TRACEBACK kernel (separate launch, one lane per pair): walks the bit-packed direction matrix,
//     emulating the reference's checkerboard walk (state reset when a tile is left through its top edge,
//     dropped leading gap on a corner exit) from global coordinates, run-length encodes, two passes
//     (count, exclusive scan, write) so CIGARs are emitted densely in input order.
//   * No MFMA: this is an integer max-plus recurrence.  No CPU fallback: every entry point needs the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "gnx_align.h"

namespace {

constexpr int G = 16;            // lanes per pair (one DPP row)
constexpr int R = 10;            // DP rows per lane
constexpr int H = G * R;         // rows per strip
constexpr int NEG4 = -(1 << 30); // scaled "veryNegNum" (align/align.go:8); finite keys stay above -(1<<29)
constexpr int QA = 8;            // uint4 stores per lane per flush, affine (3*R=30 dwords -> 32)
constexpr int QC = 3;            // const gap (R=10 dwords -> 12)

#define DPP_ROW_SHR1 0x111
#define DPP_ROW_SHL1 0x101

struct PairPlan {
    int32_t n, m;
    int32_t words;      // 16-column direction words per strip
    int32_t strips;     // ceil(n / H)
    int64_t trace_off;  // in uint4 units, relative to the chunk's trace buffer
    int64_t hcol_off;   // ints
    int64_t rowbuf_off; // int2
    int64_t dcol_off;   // dwords: per strip and lane one word with the last-column direction fields of the lane's R rows
    // fast path (short alpha) / window re-fill:
    int32_t src;        // index of the pair in the a_start / b_start tables (== own index except for window plans)
    int32_t col_off;    // first column of a window re-fill minus one (0 = whole matrix); multiple of CKW
    int64_t ckpt_off;   // int2: column checkpoints of the pair, [c-1][row] for column c*CKW
    int64_t rowi_off;   // dwords: I-plane of row n (one word per 16 steps of the owner lane)
    int64_t s_off;      // SCORED kernels: explicit 4*score matrix of the pair, column-major: S[s_off + (j-1)*s_pitch + (i-1)]
    int64_t s_pitch;
};

constexpr int CKW = 128;     // column checkpoint spacing of the fast path
constexpr int FP_SPAN = 192; // a re-fill window is at least this wide (>= 160 rows + typical indels)
constexpr int FP_PLANES = 4; // rows n .. n-3 keep their I-plane: a trailing gap sits on row n-d when the last d bases match the chunk end
constexpr int FP_CAP = 64;   // CIGAR runs staged per pair on the fast path (more -> general path)
constexpr int FP_WWORDS = (FP_SPAN + CKW + 15 + 15) / 16 + 1; // direction words of the widest window
constexpr int FP_TILE = 1024;                                  // straggler tiles: columns (c*FP_TILE, (c+1)*FP_TILE]
constexpr int FP_TWORDS = (FP_TILE + CKW + 15 + 15) / 16;      // direction words of a tile (plus the checkpoint interval it starts early)

struct KParams {
    int sc4[25]; // 4*scores
    int oe4, e4, o4;
    int d00_4;   // 4*D(0,0): gapOpen, or 0 with free end gaps
    int ecol4;   // 4*(column-0 extension): gapExtend, or 0 with free end gaps
    int g4;      // const gap: 4*gapPen
};

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int dpp_shr1(int oldv, int src) { return __builtin_amdgcn_update_dpp(oldv, src, DPP_ROW_SHR1, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_shl1(int oldv, int src) { return __builtin_amdgcn_update_dpp(oldv, src, DPP_ROW_SHL1, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned alignbit2(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 2); }

// ------------------------------------------------------------------------------------------------------
// Affine fill.
//   LOCAL = free end gaps (AffineGapLocal, affineGap_highMem.go:188-210)
//   MULTI = some pair of the launch has more than one 160-row strip (row buffer hand-over code compiled in)
//   P16   = 4*score fits int16: the per-row score profile is stored as packed int16 pairs in LDS
//   HFORM = gapOpen <= 0 (every caller): with h = max3(M,I,D) the recurrences collapse to
//           rt = max(h+oe, I+e), dn = max(h+oe, D+e) with identical values AND identical argmax tags
//           (oe <= e makes the dropped candidate I+oe / D+oe never a strict winner; ties keep M > I > D).
//           gfx950 issues v_max_i32/v_max3_i32/v_and_or/v_alignbit/DPP/SDWA and any VALU op with an SGPR
//           operand at 4 cycles per wave64 but VGPR/immediate add/or at 2 (tools/valu_ubench*.hip), so the
//           penalties are kept in VGPRs and the h-form trades 2 max3 + 2 adds for 2 max.
// LDS (dwords): [0,32) 4*score table; then per pair g a profile  prof[b][lane][LW]  (b-stride BST, pair stride
// PST).  BST = 0 and PST = 16 (mod 32) make the 32 lanes of a ds_read_b32 group hit 32 distinct banks whatever
// bases they look up (lane stride 5 or 10 dwords is odd/2*odd -> a permutation within a pair, +16 for the
// second pair of the group fills the complement).
// ------------------------------------------------------------------------------------------------------
template <bool P16> struct ProfCfg {
    static constexpr int LW = P16 ? R / 2 : R;       // dwords per lane per base
    static constexpr int BST = P16 ? 96 : 160;       // dwords per base (>= 16*LW, multiple of 32)
    static constexpr int PST = 5 * BST + 16;         // dwords per pair
};

template <bool LOCAL, bool MULTI, bool P16, bool HFORM, bool WIN = false, bool SCORED = false>
__global__ __launch_bounds__(64) void fill_affine_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                         const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                         const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                         KParams kp, uint4 *__restrict__ trace, int *__restrict__ hcol,
                                                         int2 *__restrict__ rowbuf, unsigned *__restrict__ dcol, const int2 *__restrict__ ckpt,
                                                         int *__restrict__ err, const int *__restrict__ smat = nullptr) {
    // SCORED: the substitution score of a cell comes from an explicit per-pair matrix in HBM (chunk / multiple-alignment
    //      variants, "next" row N1) instead of the LDS profile of alpha x the base of the column; sequences are not read.
    // WIN: window / tile re-fill of the fast path: the left boundary comes from a column checkpoint written by
    //      fp_sweep_kernel (pl.col_off > 0), the row-0 boundary and the beta window start at column col_off.
    using PC = ProfCfg<P16>;
    constexpr int LW = PC::LW, BST = PC::BST, PST = PC::PST;
    __shared__ int lds[32 + 4 * PST];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    if (lane < 25) lds[lane] = kp.sc4[lane];
    int *prof = &lds[32 + g * PST];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);

    const int pbase = blockIdx.x * 4;
    int S_max = 0, m_max = 0;
    for (int q = 0; q < 4; q++) {
        if (pbase + q < n_pairs) { S_max = max(S_max, plans[pbase + q].strips); m_max = max(m_max, plans[pbase + q].m); }
    }
    const int p = pbase + g;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const uint8_t *ap = SCORED ? nullptr : a_buf + (valid ? a_start[pl.src] : 0);
    const uint8_t *bp = SCORED ? nullptr : b_buf + (valid ? b_start[pl.src] + (WIN ? pl.col_off : 0) : 0);
    const int Tend = (m_max + 15 + 15) & ~15;
    const int OE4 = kp.oe4, E4 = kp.e4;
    // h-form carries X = h + e instead of h (XE = e): then I+e and D+e are one 2-cycle `and` + one 2-cycle add
    // each and no separate retag is needed; (X|3) + s == M + e because e is a multiple of 4.
    const int XE = HFORM ? kp.e4 : 0;
    int vOE4, vE4, vO4, vE4p2, vE4p1; // constants pinned in VGPRs (2-cycle adds)
    asm volatile("v_mov_b32 %0, %5\n\tv_mov_b32 %1, %6\n\tv_mov_b32 %2, %7\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9"
                 : "=v"(vOE4), "=v"(vE4), "=v"(vO4), "=v"(vE4p2), "=v"(vE4p1)
                 : "s"(kp.oe4), "s"(kp.e4), "s"(kp.o4), "s"(kp.e4 + 2), "s"(kp.e4 + 1));
    int bad = 0;

    for (int s = 0; s < S_max; s++) {
        const bool gact = valid && s < pl.strips;
        const int m_eff = gact ? pl.m : 0;
        int m_min = 0x7fffffff; // over the 4 pairs of the wave, this strip (wave-uniform)
        for (int q = 0; q < 4; q++) m_min = min(m_min, (pbase + q < n_pairs && s < plans[pbase + q].strips) ? plans[pbase + q].m : 0);
        const bool store_row = MULTI && gact && (s + 1 < pl.strips);
        const int row0 = s * H + l * R; // 0-based index of this lane's first row == 1-based index of the row above it
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
        if (WIN && pl.col_off > 0) ck0 = ckpt + pl.ckpt_off + (int64_t)(pl.col_off / CKW - 1) * pl.n;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = row0 + r + 1; // column-0 cell of row i: M = I = -inf, D = D00 + i*ecol
            const int D1c = kp.d00_4 + i * kp.ecol4 + 1;
            hold[r] = max3i(NEG4 + 3, NEG4 + 2, D1c) + XE;
            rt[r] = max3i(NEG4 + 3 + OE4, NEG4 + 2 + E4, D1c + OE4);
            if (WIN && ck0 && gact && i <= pl.n) { const int2 v = ck0[i - 1]; rt[r] = v.x; hold[r] = v.y; }
            acc[r] = 0; acc[R + r] = 0; acc[2 * R + r] = 0;
        }
        int diag0 = ((row0 == 0) ? max3i(3, kp.o4 + 2, kp.d00_4 + 1) : max3i(NEG4 + 3, NEG4 + 2, kp.d00_4 + row0 * kp.ecol4 + 1)) + XE;
        if (WIN && ck0) {
            if (row0 == 0) diag0 = max3i(NEG4 + 3, kp.o4 + pl.col_off * E4 + 2, NEG4 + 1) + XE; // h(0, col_off)
            else if (gact && row0 <= pl.n) diag0 = ck0[row0 - 1].y;
        }
        int dn_out = 0, h_out = 0, b_out = 0;
        int sq_dn = 0, sq_h = 0;
        // boundary queues (row above the strip + beta): lane u holds column t0+u+1 of the current 16-step block
        int qdn, qh, qb, ndn = 0, nh = 0, nb = 0;
        auto boundary = [&](int c, int &odn, int &oh, int &ob) {
            if (!MULTI || s == 0) {
                const int M3 = NEG4 + 3, I2 = kp.o4 + ((WIN ? pl.col_off : 0) + c) * E4 + 2, D1 = NEG4 + 1; // row 0: I(0,c) = gapOpen + c*gapExtend
                const int h0 = max3i(M3, I2, D1);
                odn = (LOCAL && c == m_eff) ? h0 : max3i(M3 + OE4, I2 + OE4, D1 + E4);
                oh = h0 + XE;
            } else if (c >= 1 && c <= m_eff) {
                const int2 v = rowbuf[pl.rowbuf_off + c];
                odn = v.x; oh = v.y; // already in the X domain
            } else { odn = 0; oh = 0; }
            int b = 0;
            if (!SCORED && c >= 1 && c <= m_eff) { b = bp[c - 1]; if (b >= 5) { bad = 1; b = 4; } }
            ob = b * (BST * 4); // LDS byte offset of the base's profile plane
        };
        if (MULTI && s > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        boundary(l + 1, qdn, qh, qb);

        // one anti-diagonal step; CHECK=false is the steady state (every lane of the wave has a live column)
        auto step = [&](const int t, auto chk) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_dn = dpp_shr1(qdn, dn_out);
            const int up_h = dpp_shr1(qh, h_out);
            const int pb = dpp_shr1(qb, b_out);
            qdn = dpp_shl1(qdn, qdn);
            qh = dpp_shl1(qh, qh);
            qb = dpp_shl1(qb, qb);
            const int j = t - l;
            b_out = pb;
            if (!CHECK || (j >= 1 && j <= m_eff)) {
                const int *pw = SCORED ? smat + pl.s_off + (int64_t)(j - 1) * pl.s_pitch + row0
                                       : reinterpret_cast<const int *>(prof_lane + pb);
                int w[LW];
#pragma unroll
                for (int k = 0; k < LW; k++) w[k] = pw[k];
                int hd = diag0, dnu = up_dn;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int S4;
                    if (P16) S4 = (r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff);
                    else S4 = w[r];
                    acc[r] = alignbit2((unsigned)hd, acc[r]);
                    acc[R + r] = alignbit2((unsigned)rt[r], acc[R + r]);
                    acc[2 * R + r] = alignbit2((unsigned)dnu, acc[2 * R + r]);
                    int hnew, dnn; // hnew is in the X domain (h + XE)
                    if (HFORM) {
                        const int M3e = (hd | 3) + S4;             // M + e
                        const int Ie = (rt[r] & ~3) + vE4p2;       // I + e, tag 2
                        const int De = (dnu & ~3) + vE4p1;         // D + e, tag 1
                        hnew = max3i(M3e, Ie, De);                 // h + e
                        const int hoe = hnew + vO4;                // h + oe
                        rt[r] = max(hoe, Ie);
                        dnn = max(hoe, De);
                        if (LOCAL) dnn = (j == m_eff) ? hnew - vE4 : dnn; // last column: D(i+1,m) = tmt(M,I,D)(i,m), no penalty
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
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            boundary(t0 + 16 + l + 1, ndn, nh, nb); // prefetch the next block's boundary
            const bool steady = t0 >= 16 && t0 + 16 <= m_min;
            if (steady) {
#pragma unroll 2
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::false_type{});
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::true_type{});
            }
            qdn = ndn; qh = nh; qb = nb;
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
                for (int q = 0; q < QA - 1; q++) dst[q * G] = make_uint4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                dst[(QA - 1) * G] = make_uint4(acc[4 * (QA - 1)], acc[4 * (QA - 1) + 1], 0u, 0u);
            }
            if (store_row) {
                const int c = t0 + l - 14;
                if (c >= 1 && c <= m_eff) rowbuf[pl.rowbuf_off + c] = make_int2(sq_dn, sq_h);
            }
        }
        if (gact && m_eff >= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) if (row0 + r < pl.n) hcol[pl.hcol_off + row0 + r] = hold[r] - XE;
            // last-column D-plane fields of this lane's rows, packed (field r at bits 2r): lets the traceback skip
            // vertical runs in column m (free end gaps of AffineGapLocal, trailing gaps when alpha is the long one)
            const int t0f = ((m_eff + l - 1) >> 4) << 4, missf = t0f + 16 - l - m_eff; // 0..15: in-place drain shift
            unsigned dw = 0;
#pragma unroll
            for (int r = 0; r < R; r++) dw |= ((acc[2 * R + r] >> (30 - 2 * missf)) & 3u) << (2 * r);
            dcol[pl.dcol_off + s * G + l] = dw;
        }
        if (MULTI) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    if (bad) atomicOr(err, 1);
}

// ------------------------------------------------------------------------------------------------------
// Fast-path forward sweep: 8 lanes x RR rows per pair, 8 pairs per wave64 (global affine, gapOpen <= 0, n <= 8*RR).
// Same anti-diagonal wavefront as fill_affine_kernel, re-cut for the short-alpha shape:
//  * REBASED keys.  Every cell quantity V(i,j) in {M, I, D, h} is carried as V' = V - e*(i+j).  The recurrences keep
//    all their comparisons (each max compares candidates of the same cell, i.e. with the same offset) and become
//        M'(i,j) = h'(i-1,j-1) + (s - 2e)    I'(i,j+1) = max(h'(i,j) + o, I'(i,j))    D'(i+1,j) = max(h'(i,j) + o, D'(i,j))
//    (h-form, see fill_affine_kernel): both extensions cost nothing, both opens share h' + o, and every boundary
//    (row 0, column 0) is a constant: per cell  add, max3, add, max, max  = 16 issue cycles instead of 20.
//  * rows are RIGHT-ALIGNED: the pair's 8*RR slots end at row n, the first P = 8*RR - n slots are padding that reproduces
//    row 0 (profile entry -32768 so M never wins; I' = h' = o, D' = 2o are fixed points of the recurrences when o <= 0).
//    So rows n .. n-3 (whose I-planes are kept, FP_PLANES) are always the last four slots of the last lane: one kernel
//    for every n, pairs of different length mix freely, and only 4 of RR rows per lane pay the tag arithmetic.
//  * two pairs per 16-lane DPP row, the second one mirrored (lane 15 is its first lane), so that "value of the previous
//    lane of my pair" is row_shr:1 on banks 0-1 plus row_shl:1 on banks 2-3 and the lanes without a source keep the
//    boundary constant passed as `old`.
//  * int16 score profile (4*(s-2e)) read as 5 ds_read_b64 per step: an LDS read costs the issuing SIMD ~2 cycles + 2 per
//    returned dword (tools/lds_ubench.hip), so 20 rows cost 30 cycles for 8 pairs instead of 60 for 4.
// Outputs (what fp_walk_kernel and the window re-fills of fill_affine_kernel<.., WIN> consume): un-rebased, tagged column
// checkpoints {rt = I(i,j+1), X = h(i,j)+e} of every row every CKW columns, the I-plane words of rows n..n-3
// (word = step >> 4, field = step & 15 with step = j + 7), and h(n,m).
// ------------------------------------------------------------------------------------------------------
constexpr int G8 = 8;
constexpr int FP8_BST = 96;                // dwords per base plane (>= 8 lanes * 10 dwords, multiple of 32)
constexpr int FP8_PST = 5 * FP8_BST + 16;  // dwords per pair: == 16 (mod 32), the two pairs of a 16-lane group hit disjoint banks
constexpr int FP8_LW = 10;                 // dwords per lane per base (20 int16 entries)

// lanes 0-7 of a DPP row: from lane-1; lanes 8-15 (mirrored pair): from lane+1; the first lane of each pair keeps oldv
__device__ __forceinline__ int dpp_prev8(int oldv, int src) {
    const int v = __builtin_amdgcn_update_dpp(oldv, src, DPP_ROW_SHR1, 0xf, 0x3, false);
    return __builtin_amdgcn_update_dpp(v, src, DPP_ROW_SHL1, 0xf, 0xc, false);
}
// the opposite direction (queue rotation towards the first lane of the pair)
__device__ __forceinline__ int dpp_next8(int oldv, int src) {
    const int v = __builtin_amdgcn_update_dpp(oldv, src, DPP_ROW_SHL1, 0xf, 0x3, false);
    return __builtin_amdgcn_update_dpp(v, src, DPP_ROW_SHR1, 0xf, 0xc, false);
}

// 2 waves per SIMD by choice: capping the kernel at 168 VGPRs for a third wave makes the compiler shuffle registers in the
// unrolled loop and costs 20 % (measured: 32.2 ms vs 38.6-40.6 ms per 100 k pairs); 16 000 B of LDS allow 10 waves per CU.
template <int RR>
__global__ __launch_bounds__(64) void fp_sweep_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                      const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                      const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                      KParams kp, int *__restrict__ hcol, int2 *__restrict__ ckpt,
                                                      unsigned *__restrict__ rowi, unsigned *__restrict__ tail, int *__restrict__ err) {
    static_assert(RR <= 2 * FP8_LW && RR > FP_PLANES, "rows per lane");
    __shared__ int lds[32 + 8 * FP8_PST];
    const int lane = threadIdx.x;
    const int g = lane >> 3;
    const int lp = (lane & 8) ? 15 - (lane & 15) : (lane & 7); // position of the lane inside its pair
    const int E4 = kp.e4;
    if (lane < 25) lds[lane] = kp.sc4[lane] - 2 * E4;
    else if (lane < 32) lds[lane] = -32768; // padding rows: the diagonal candidate never wins
    int *prof = &lds[32 + g * FP8_PST];
    const char *prof_lane = reinterpret_cast<const char *>(prof + lp * FP8_LW);

    const int pbase = blockIdx.x * 8;
    int m_max = 0, m_min = 0x7fffffff;
    for (int q = 0; q < 8; q++) {
        const int mq = (pbase + q < n_pairs) ? plans[pbase + q].m : 0;
        m_max = max(m_max, mq); m_min = min(m_min, mq);
    }
    const int p = pbase + g;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const uint8_t *ap = a_buf + (valid ? a_start[pl.src] : 0);
    const uint8_t *bp = b_buf + (valid ? b_start[pl.src] : 0);
    const int m_eff = valid ? pl.m : 0;
    const int P = G8 * RR - pl.n; // padding slots above row 1
    const int q0 = lp * RR;       // first slot of this lane; slot q holds row q - P + 1
    int bad = 0;
    int vO4, cH, cDN; // constants pinned in VGPRs (2-cycle adds, DPP `old` operands)
    asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=v"(vO4), "=v"(cH), "=v"(cDN) : "s"(kp.o4), "s"(kp.o4 + 2), "s"(2 * kp.o4 + 2));

    { // int16 profile of this lane's rows: prof[b][lp][r] = 4*(scores[alpha[row]][b] - 2e), padding -32768
        int a5[2 * FP8_LW];
#pragma unroll
        for (int r = 0; r < 2 * FP8_LW; r++) {
            int a = 5; // padding
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
    int rt[RR], hold[RR];
#pragma unroll
    for (int r = 0; r < RR; r++) {
        const int q = q0 + r;
        // column 0: real row i: h'(i,0) = D'(i,0) = o (tag 1), I'(i,1) = 2o (from D); padding: h' = I' = o; the slot above row 1 is h(0,0) = 0 (tag 3)
        hold[r] = (q >= P) ? kp.o4 + 1 : (q == P - 1 ? 3 : kp.o4 + 2);
        rt[r] = (q >= P) ? 2 * kp.o4 + 1 : kp.o4 + 2;
    }
    unsigned accR[FP_PLANES] = {0u, 0u, 0u, 0u}; // I-planes of rows n-d = slots RR-1-d of the last lane
    unsigned tailw = 0; // argmax tags of h(n-d, m-x), d, x = 0..3, field 4x + d: lets the walk take its first diagonal steps without a window
    int diag0 = (q0 == 0) ? (P == 0 ? 3 : kp.o4 + 2) : ((q0 - 1 >= P) ? kp.o4 + 1 : (q0 - 1 == P - 1 ? 3 : kp.o4 + 2));
    int dn_out = 0, h_out = 0, b_out = 0;
    auto base_of = [&](int c) { // LDS byte offset of the profile plane of beta[c] (column c, 1-based)
        int b = 0;
        if (c >= 1 && c <= m_eff) { b = bp[c - 1]; if (b >= 5) { bad = 1; b = 4; } }
        return b * (FP8_BST * 4);
    };
    int qb = base_of(lp), nb = 0; // base queue: lane lp holds column t0 + lp of the current 8-step half block

    auto step = [&](const int t, auto chk, auto ckt) {
        constexpr bool CHECK = decltype(chk)::value;
        constexpr bool CKPT = decltype(ckt)::value; // this half block crosses a checkpoint column
        const int up_dn = dpp_prev8(cDN, dn_out);
        const int up_h = dpp_prev8(cH, h_out);
        const int pb = dpp_prev8(qb, b_out);
        qb = dpp_next8(qb, qb);
        const int j = t - lp;
        b_out = pb;
        if (!CHECK || (j >= 1 && j <= m_eff)) {
            const int2 *pw = reinterpret_cast<const int2 *>(prof_lane + pb);
            int w[FP8_LW];
#pragma unroll
            for (int k = 0; k < FP8_LW / 2; k++) { const int2 v = pw[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
            int hd = diag0, dnu = up_dn;
#pragma unroll
            for (int r = 0; r < RR; r++) {
                const int S4 = (r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff);
                if (r >= RR - FP_PLANES) accR[RR - 1 - r] = alignbit2((unsigned)rt[r], accR[RR - 1 - r]);
                int hnew, dnn;
                if (r < RR - FP_PLANES) { // tag bits are junk < 4 here; they never change the value of a max
                    const int M = hd + S4;
                    hnew = max3i(M, rt[r], dnu);
                    const int ho = hnew + vO4;
                    rt[r] = max(ho, rt[r]);
                    dnn = max(ho, dnu);
                } else {
                    const int M3 = (hd | 3) + S4;
                    const int I2 = (rt[r] & ~3) | 2;
                    const int D1 = (dnu & ~3) | 1;
                    hnew = max3i(M3, I2, D1);
                    const int ho = hnew + vO4;
                    rt[r] = max(ho, I2);
                    dnn = max(ho, D1);
                }
                hd = hold[r];
                hold[r] = hnew;
                dnu = dnn;
            }
            diag0 = up_h;
            dn_out = dnu;
            h_out = hold[RR - 1];
            if (CHECK && j + 3 >= m_eff) { // the last four columns (always in a CHECK half block)
#pragma unroll
                for (int d = 0; d < FP_PLANES; d++) tailw |= (unsigned)(hold[RR - 1 - d] & 3) << (8 * (m_eff - j) + 2 * d);
            }
            if (CKPT && (j & (CKW - 1)) == 0 && j < m_eff && valid) { // column checkpoint, un-rebased values {I(i,j+1), h(i,j)+e} (tag bits junk)
                int2 *ck = ckpt + pl.ckpt_off + (int64_t)(j / CKW - 1) * pl.n;
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    const int i = q0 + r - P + 1;
                    const int off = E4 * (i + j + 1);
                    if (i >= 1) ck[i - 1] = make_int2(rt[r] + off, hold[r] + off);
                }
            }
        }
    };

    // half blocks of 8 steps t0 .. t0+7 (step t: lane lp is at column t - lp); a plane word is two half blocks
    const int Tend = ((m_max + G8 - 1) / 16 + 1) * 16;
    for (int t0 = 0; t0 < Tend; t0 += 8) {
        nb = base_of(t0 + 8 + lp); // prefetch the next half block's bases
        const bool ckblk = (t0 & (CKW - 1)) == 0; // steps t0 + lp are the lanes' checkpoint columns
        const bool steady = t0 >= 8 && t0 + 7 <= m_min;
        if (steady && !ckblk) {
#pragma unroll 2
            for (int u = 0; u < 8; u++) step(t0 + u, std::false_type{}, std::false_type{});
        } else if (steady) {
#pragma unroll 1
            for (int u = 0; u < 8; u++) step(t0 + u, std::false_type{}, std::true_type{});
        } else {
#pragma unroll 1
            for (int u = 0; u < 8; u++) step(t0 + u, std::true_type{}, std::true_type{});
        }
        qb = nb;
        if ((t0 & 8) && lp == G8 - 1 && valid) { // the last lane owns rows n..n-3: flush the plane word of steps t0-8 .. t0+7
            const int w = t0 >> 4;
            if (w < pl.words) {
                const int miss = (t0 + 7) - (m_eff + G8 - 1); // steps this lane sat idle after its last column
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
                for (int d = 0; d < FP_PLANES; d++) {
                    accR[d] >>= sh;
                    if (pl.n - d >= 1) rowi[pl.rowi_off + (int64_t)d * pl.words + w] = accR[d];
                }
            }
        }
    }
    if (lp == G8 - 1 && valid && m_eff >= 1) { hcol[pl.hcol_off] = hold[RR - 1] + E4 * (pl.n + m_eff); tail[pl.hcol_off] = tailw; } // h(n, m)
    if (bad) atomicOr(err, 1);
}

// ------------------------------------------------------------------------------------------------------
// Constant-gap fill (align/constGap.go:146-157 recurrence), same wavefront mapping as the affine kernel.
// Keys: diag+s -> tag 3, left+g -> tag 2, up+g -> tag 1; the stored value is the clean (tag-free) key, so the three
// candidates are three 2-cycle VGPR adds (the profile holds 4*s+3, the penalties 4*g+2 / 4*g+1 live in VGPRs),
// one v_max3, one v_and and one v_alignbit per cell.
// ------------------------------------------------------------------------------------------------------
// GSW (the seed-extension DP of the graph aligner, "next" row N2, /root/reference/genomeGraph/search.go:234-321):
//   1 = LeftDynamicAln: zero borders, cell values clamped at 0 (the trace keeps its direction);
//   2 = RightDynamicAln: the ordinary borders plus, per row, the running maximum of (score << 12 | 4095 - column), i.e. the first
//       column of the row's best score; hcol receives that key instead of the last-column value.
template <bool MULTI, int GSW = 0>
__global__ __launch_bounds__(64) void fill_const_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                        const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                        const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                        KParams kp, uint4 *__restrict__ trace, int *__restrict__ hcol,
                                                        int2 *__restrict__ rowbuf, unsigned *__restrict__ dcol, int *__restrict__ err) {
    using PC = ProfCfg<false>;
    constexpr int LW = PC::LW, BST = PC::BST, PST = PC::PST;
    __shared__ int lds[32 + 4 * PST];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    if (lane < 25) lds[lane] = kp.sc4[lane] + 3; // pre-tagged diagonal candidate
    int *prof = &lds[32 + g * PST];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    const int pbase = blockIdx.x * 4;
    int S_max = 0, m_max = 0;
    for (int q = 0; q < 4; q++) {
        if (pbase + q < n_pairs) { S_max = max(S_max, plans[pbase + q].strips); m_max = max(m_max, plans[pbase + q].m); }
    }
    const int p = pbase + g;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const uint8_t *ap = a_buf + (valid ? a_start[p] : 0);
    const uint8_t *bp = b_buf + (valid ? b_start[p] : 0);
    const int Tend = (m_max + 15 + 15) & ~15;
    int vGL, vGU;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(vGL), "=v"(vGU) : "s"(kp.g4 + 2), "s"(kp.g4 + 1));
    int bad = 0;

    for (int s = 0; s < S_max; s++) {
        const bool gact = valid && s < pl.strips;
        const int m_eff = gact ? pl.m : 0;
        int m_min = 0x7fffffff;
        for (int q = 0; q < 4; q++) m_min = min(m_min, (pbase + q < n_pairs && s < plans[pbase + q].strips) ? plans[pbase + q].m : 0);
        const bool store_row = MULTI && gact && (s + 1 < pl.strips);
        const int row0 = s * H + l * R;
        int val[R];
        unsigned acc[R];
        {
            int a5[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i0 = row0 + r;
                int a = 0;
                if (gact && i0 < pl.n) { a = ap[i0]; if (a >= 5) { bad = 1; a = 4; } }
                a5[r] = a * 5;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = lds[a5[k] + b];
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < R; r++) { val[r] = GSW == 1 ? 0 : (row0 + r + 1) * kp.g4; acc[r] = 0; } // column 0: i*gapPen
        int best[R];
#pragma unroll
        for (int r = 0; r < R; r++) best[r] = 4095; // score 0: only a positive score replaces it (currMax starts at 0)
        int diag0 = GSW == 1 ? 0 : row0 * kp.g4; // V(row above, 0)
        int v_out = 0, b_out = 0, sq_v = 0;
        int qv, qb, nv = 0, nb = 0;
        auto boundary = [&](int c, int &ov, int &ob) {
            if (!MULTI || s == 0) ov = GSW == 1 ? 0 : c * kp.g4; // row 0: j*gapPen
            else if (c >= 1 && c <= m_eff) ov = rowbuf[pl.rowbuf_off + c].x;
            else ov = 0;
            int b = 0;
            if (c >= 1 && c <= m_eff) { b = bp[c - 1]; if (b >= 5) { bad = 1; b = 4; } }
            ob = b * (BST * 4);
        };
        if (MULTI && s > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        boundary(l + 1, qv, qb);

        auto step = [&](const int t, auto chk) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_v = dpp_shr1(qv, v_out);
            const int pb = dpp_shr1(qb, b_out);
            qv = dpp_shl1(qv, qv);
            qb = dpp_shl1(qb, qb);
            const int j = t - l;
            b_out = pb;
            if (!CHECK || (j >= 1 && j <= m_eff)) {
                const int *pw = reinterpret_cast<const int *>(prof_lane + pb);
                int w[LW];
#pragma unroll
                for (int k = 0; k < LW; k++) w[k] = pw[k];
                int vd = diag0, vu = up_v;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int k = max3i(vd + w[r], val[r] + vGL, vu + vGU);
                    acc[r] = alignbit2((unsigned)k, acc[r]);
                    vd = val[r];
                    val[r] = k & ~3;
                    if (GSW == 1) val[r] = max(val[r], 0);
                    if (GSW == 2) best[r] = max(best[r], (int)((unsigned)val[r] << 10) + (4095 - j));
                    vu = val[r];
                }
                diag0 = up_v;
                v_out = vu;
            }
            if (MULTI) sq_v = dpp_shl1(v_out, sq_v);
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            boundary(t0 + 16 + l + 1, nv, nb);
            if (t0 >= 16 && t0 + 16 <= m_min) {
#pragma unroll 2
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::false_type{});
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::true_type{});
            }
            qv = nv; qb = nb;
            const int w = t0 >> 4;
            if (gact && w < pl.words) {
                const int miss = (t0 + 16 - l) - m_eff;
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
                if (t0 + 16 > m_min) {
#pragma unroll
                    for (int d = 0; d < R; d++) acc[d] >>= sh;
                }
                uint4 *dst = trace + pl.trace_off + ((int64_t)(s * pl.words + w) * QC) * G + l;
                dst[0] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
                dst[G] = make_uint4(acc[4], acc[5], acc[6], acc[7]);
                dst[2 * G] = make_uint4(acc[8], acc[9], 0u, 0u);
            }
            if (store_row) {
                const int c = t0 + l - 14;
                if (c >= 1 && c <= m_eff) rowbuf[pl.rowbuf_off + c] = make_int2(sq_v, 0);
            }
        }
        if (gact && m_eff >= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) if (row0 + r < pl.n) hcol[pl.hcol_off + row0 + r] = GSW == 2 ? best[r] : val[r];
            const int t0f = ((m_eff + l - 1) >> 4) << 4, missf = t0f + 16 - l - m_eff;
            unsigned dw = 0;
#pragma unroll
            for (int r = 0; r < R; r++) dw |= ((acc[r] >> (30 - 2 * missf)) & 3u) << (2 * r);
            dcol[pl.dcol_off + s * G + l] = dw;
        }
        if (MULTI) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    if (bad) atomicOr(err, 1);
}

// ------------------------------------------------------------------------------------------------------
// Traceback.  One lane per pair.  WRITE=false counts CIGAR runs, WRITE=true emits them (reversed into
// alignment order) at ops[ops_off[p] ..).  ci/cj = checkerboard sizes (huge for the highMem modes).
// ------------------------------------------------------------------------------------------------------
struct TbParams {
    int64_t ci, cj;
    int64_t d00, ecol, gap_open, gap_extend; // unscaled, for the empty-sequence closed forms
    int affine;
};

// direction word of cell (i,j) (1-based) for plane k (affine: 0/1/2 = M/I/D; const: 0) and the field position
// of the cell inside it -- see the flush layout in the fill kernels.  Fields of lower columns sit at lower positions.
template <bool AFFINE>
__device__ __forceinline__ unsigned load_word(const uint4 *trace, const PairPlan &pl, int k, int i, int j, int &pos) {
    const int i0 = i - 1;
    const int s = i0 / H, rem = i0 - s * H;
    const int l = rem / R, r = rem - l * R;
    const int t1 = j + l - 1;
    const int w = t1 >> 4;
    pos = t1 & 15;
    const int d = AFFINE ? k * R + r : r;
    const int Q = AFFINE ? QA : QC;
    const unsigned *base = reinterpret_cast<const unsigned *>(trace + pl.trace_off + ((int64_t)(s * pl.words + w) * Q + (d >> 2)) * G + l);
    return base[d & 3];
}

template <bool AFFINE, bool WRITE>
__global__ __launch_bounds__(64) void traceback_kernel(const PairPlan *__restrict__ plans, int n_pairs, const uint4 *__restrict__ trace,
                                                       const int *__restrict__ hcol, const unsigned *__restrict__ dcol, TbParams tp,
                                                       int64_t *__restrict__ score_out,
                                                       int64_t *__restrict__ nops, const int64_t *__restrict__ ops_off,
                                                       gnx_cigar *__restrict__ ops, int64_t ops_capacity, int *__restrict__ err) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const PairPlan pl = plans[p];
    int i = pl.n, j = pl.m;
    int64_t score;
    int k;
    if (i > 0 && j > 0) {
        const int hc = hcol[pl.hcol_off + pl.n - 1];
        score = (int64_t)(hc >> 2);
        if (AFFINE) k = 3 - (hc & 3);
        else k = 0;
    } else { // highMem modes with an empty sequence: closed forms of row 0 / column 0
        if (AFFINE) {
            if (i == 0 && j == 0) { // tmt(0, gapOpen, D00)
                const int64_t a = 0, b = tp.gap_open, c = tp.d00;
                if (a >= b && a >= c) { score = a; k = 0; } else if (b >= c) { score = b; k = 1; } else { score = c; k = 2; }
            } else if (i == 0) { score = tp.gap_open + (int64_t)j * tp.gap_extend; k = 1; }
            else { score = tp.d00 + (int64_t)i * tp.ecol; k = 2; }
        } else { score = (int64_t)(i + j) * tp.gap_open; k = 0; }
    }
    const int po = pl.src; // output slot (== p except for sub-batches routed here by the fast path)
    if (!WRITE) score_out[po] = score;

    int64_t cnt = 0;           // runs emitted so far (traceback order)
    int cur_op = -1;
    int64_t cur_run = 0;
    const int64_t total = WRITE ? nops[po] : 0;
    const int64_t obase = WRITE ? ops_off[po] : 0;
    const bool fits = WRITE ? (obase + total <= ops_capacity) : false;
    auto flush_run = [&]() {
        if (cur_op >= 0) {
            if (WRITE && fits) {
                gnx_cigar c; c.run_length = cur_run; c.op = (uint8_t)cur_op;
                for (int z = 0; z < 7; z++) c._pad[z] = 0;
                ops[obase + (total - 1 - cnt)] = c;
            }
            cnt++;
        }
    };
    auto emit = [&](int op, int64_t run) {
        if (op == cur_op) cur_run += run;
        else { flush_run(); cur_op = op; cur_run = run; }
    };

    // The checkerboard walk in global coordinates.  A tile is left through its top edge when the new row index
    // is a multiple of checkersize_i, through its left edge when the new column index is a multiple of
    // checkersize_j (affineGap.go:121-127).
    int64_t li = (i > 0) ? (int64_t)(i - 1) % tp.ci : 0; // tile-local row of the current cell
    int last_op = -1;
    const bool walked = (i > 0 && j > 0);
    while (i > 0 && j > 0) {
        if (j == pl.m && (!AFFINE || k == 2)) {
            // Vertical run in the last column: the packed per-lane word holds the fields of R consecutive rows.
            const int i0 = i - 1, sl = i0 / R, r = i0 - sl * R; // sl = strip*16 + lane
            const unsigned w = dcol[pl.dcol_off + sl];
            int tag = (int)((w >> (2 * r)) & 3u);
            if (tag == 0) { atomicOr(err, 2); break; }
            if (AFFINE || tag == 1) {
                int avail = min(r + 1, i);
                if (li + 1 < (int64_t)avail) avail = (int)(li + 1); // do not run past the tile's top edge (quirk Q1 applies there)
                unsigned x = w ^ 0x55555555u;                        // fields "from D" (tag 1) become 0
                if (r < 15) x &= (1u << (2 * r + 2)) - 1u;
                const int lowcut = r + 1 - avail;
                if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
                int steps;
                bool cont = false; // the walk is still in a D cell after the run
                if (x == 0) { steps = avail; cont = true; }
                else {
                    const int rnz = (31 - __clz((int)x)) >> 1;
                    tag = (int)((w >> (2 * rnz)) & 3u);
                    if (tag == 0) { atomicOr(err, 2); break; }
                    if (AFFINE) steps = r - rnz + 1; else { steps = r - rnz; cont = true; }
                }
                if (steps > 0) {
                    emit(2, steps); i -= steps; last_op = 2;
                    li -= steps;
                    const bool up_exit = li < 0;
                    if (up_exit) li += tp.ci;
                    if (AFFINE) {
                        k = cont ? 2 : 3 - tag;
                        if (up_exit && i > 0) k = 3 - (hcol[pl.hcol_off + i - 1] & 3); // quirk Q1, entry cell (i, m)
                    }
                    continue;
                }
            }
        }
        int pos;
        const unsigned w = load_word<AFFINE>(trace, pl, AFFINE ? k : 0, i, j, pos);
        int tag = (int)((w >> (2 * pos)) & 3u);
        const int op = AFFINE ? k : 3 - tag;
        if (tag == 0) { atomicOr(err, 2); break; } // impossible direction: the Go code would log.Fatalf
        if (op == 1) {
            // Horizontal run: every cell visited in state I emits one I and moves left; the walk stays in this word
            // while the fields read "came from I" (tag 2).  Count them with one xor + clz instead of 16 iterations.
            const int avail = min(pos + 1, j);            // fields of columns >= 1 at positions pos .. pos-avail+1
            unsigned x = w ^ 0xAAAAAAAAu;
            if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
            const int lowcut = pos + 1 - avail;
            if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
            int steps;
            if (x == 0) { steps = avail; if (AFFINE) k = 1; }
            else {
                const int pnz = (31 - __clz((int)x)) >> 1;  // highest field that is not "from I"
                tag = (int)((w >> (2 * pnz)) & 3u);
                if (tag == 0) { atomicOr(err, 2); break; }
                if (AFFINE) { steps = pos - pnz + 1; k = 3 - tag; } // that cell is still in state I; its source decides the next state
                else steps = pos - pnz;                            // const gap: that cell is not an I cell
            }
            if (steps > 0) { emit(1, steps); j -= steps; last_op = 1; }
            continue;
        }
        emit(op, 1);
        last_op = op;
        bool up_exit = false;
        if (op != 1) { up_exit = (li == 0); li = up_exit ? tp.ci - 1 : li - 1; i--; }
        if (op != 2) j--;
        if (AFFINE) {
            k = 3 - tag;
            if (up_exit && i > 0 && j > 0) {
                // quirk Q1 (affineGap.go:305): entering a tile from below restarts in the argmax state of the entry cell
                int ht;
                if (j < pl.m) { int p2; ht = (int)((load_word<true>(trace, pl, 0, i + 1, j + 1, p2) >> (2 * p2)) & 3u); }
                else ht = hcol[pl.hcol_off + i - 1] & 3;
                k = 3 - ht;
            }
        }
    }
    // Step 4 (affineGap.go:135-139 / constGap.go:59-63) and the highMem border walks
    if (walked) {
        const bool up_exit = (last_op != 1) && ((int64_t)i % tp.ci == 0);
        const bool left_exit = (last_op != 2) && ((int64_t)j % tp.cj == 0);
        if (!up_exit && left_exit) emit(2, i);
        else if (up_exit && !left_exit) emit(1, j);
        // both: corner exit -> nothing (quirk Q2 when it is not the origin)
    } else { // empty sequence (highMem modes only)
        if (i == 0 && j > 0) emit(1, j);
        else if (j == 0 && i > 0) emit(2, i);
        else { cur_op = 0; cur_run = 0; } // Go: route == [{0 0}]
    }
    flush_run();
    if (!WRITE) nops[po] = cnt;
    else if (!fits) atomicOr(err, 4);
}

// ------------------------------------------------------------------------------------------------------
// Fast path (alpha fits one strip, global affine, gapOpen <= 0): traceback by stages.
//   fp_walk: one lane per pair.  On row n in state I it follows the stored I-plane of row n (the long trailing
//   gap of a short read against a long chunk) a word at a time; anywhere else it needs full direction bits and
//   requests a re-fill of the <= FP_SPAN+CKW-1 columns left of the current cell from the nearest column
//   checkpoint (window plan appended to a list), which fill_affine_kernel<.., WIN> computes with the normal
//   recording; the next fp_walk call continues inside that window.  Row 0 / column 0 end the walk (Step 4).
//   CIGAR runs are staged per pair in traceback order and reversed into place by fp_compact.
// Same checkerboard-walk emulation (Q1/Q2) as traceback_kernel.
// ------------------------------------------------------------------------------------------------------
struct FpState {
    int32_t i, j, k, last_op;
    int32_t cur_op, cnt, status, slot; // status 0 = needs a window, 1 = done; slot = window slot of the last request
    int64_t cur_run;
    int64_t li;
    int32_t j_hi, jc_lo;
};

template <bool FIRST, bool TILED = false>
__global__ __launch_bounds__(64) void fp_walk_kernel(const PairPlan *__restrict__ plans, const int *__restrict__ active, int n_active,
                                                     FpState *__restrict__ states, const int *__restrict__ hcol_fwd,
                                                     const unsigned *__restrict__ rowi, const unsigned *__restrict__ tail,
                                                     const PairPlan *__restrict__ wplans, const uint4 *__restrict__ wtrace,
                                                     const int *__restrict__ whcol, TbParams tp, gnx_cigar *__restrict__ stage,
                                                     int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                     int *__restrict__ next_active, int *__restrict__ next_count,
                                                     PairPlan *__restrict__ next_wplans, int *__restrict__ err, int p_base) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_active) return;
    const int p = FIRST ? a + p_base : active[a];
    const PairPlan pl = plans[p];
    FpState st;
    PairPlan wp;
    if (FIRST) {
        const int hc = hcol_fwd[pl.hcol_off];
        score_out[p] = (int64_t)(hc >> 2);
        st.i = pl.n; st.j = pl.m; st.k = 3 - (hc & 3); st.last_op = -1;
        st.cur_op = -1; st.cnt = 0; st.status = 0; st.slot = -1; st.cur_run = 0;
        st.li = (int64_t)(pl.n - 1) % tp.ci;
        st.j_hi = 0; st.jc_lo = 0; // empty window
        wp = pl;
    } else {
        st = states[p];
        wp = wplans[TILED ? 0 : a];
        if (TILED) { st.j_hi = 0; st.jc_lo = 0; }
    }
    // TILED: `a` indexes the straggler; its tiles c = 0.. are the plans [a*tiles_per + c] (tiles_per in wplans[0].rowi_off)
    const int tiles_per = TILED ? (int)wplans[0].rowi_off : 0;
    int i = st.i, j = st.j, k = st.k, last_op = st.last_op, cur_op = st.cur_op, cnt = st.cnt;
    int64_t cur_run = st.cur_run, li = st.li;
    gnx_cigar *stg = stage + (int64_t)p * FP_CAP;
    auto flush_run = [&]() {
        if (cur_op >= 0) {
            if (cnt < FP_CAP) { gnx_cigar c; c.run_length = cur_run; c.op = (uint8_t)cur_op; for (int z = 0; z < 7; z++) c._pad[z] = 0; stg[cnt] = c; }
            cnt++;
        }
    };
    auto emit = [&](int op, int64_t run) {
        if (op == cur_op) cur_run += run;
        else { flush_run(); cur_op = op; cur_run = run; }
    };
    bool done = false;
    // A re-fill that starts from a column checkpoint reproduces every VALUE, but the checkpoint carries no argmax tags
    // (the sweep computes them only on its plane rows), so the M- and I-plane fields of the re-fill's first column are
    // not usable: the walk uses a window / tile from its second column on (all of it when it starts at column 0).
    auto lo_ok = [](int jc_lo) { return jc_lo + (jc_lo > 0 ? 2 : 1); };
    // argmax tag of h(ii, jj) kept by the sweep for the 4 x 4 cells at the bottom right corner
    const unsigned tailw = tail[pl.hcol_off];
    auto tail_ok = [&](int ii, int jj) { return ii >= 1 && jj >= 1 && pl.n - ii < FP_PLANES && pl.m - jj < 4; };
    auto tail_tag = [&](int ii, int jj) { return (tailw >> (8 * (pl.m - jj) + 2 * (pl.n - ii))) & 3u; };
    while (true) {
        if (i == 0 || j == 0) { done = true; break; }
        unsigned w;
        int pos;
        const bool on_plane = (k == 1) && (pl.n - i) < FP_PLANES;
        if (on_plane) { // stored I-plane of row n-d
            const int t1 = j + G8 - 1; // step at which the owner lane (the pair's last) was at column j
            w = rowi[pl.rowi_off + (int64_t)(pl.n - i) * pl.words + (t1 >> 4)];
            pos = t1 & 15;
        } else if (j >= lo_ok(st.jc_lo) && j <= st.j_hi) { // inside the usable part of the current window
            w = load_word<true>(wtrace, wp, k, i, j - st.jc_lo, pos);
        } else if (k == 0 && tail_ok(i - 1, j - 1)) { // trM(i,j) = argmax of h(i-1,j-1): a diagonal step in the corner needs no window
            w = tail_tag(i - 1, j - 1); pos = 0;
        } else if (TILED) { // switch to the tile holding column j (all tiles of a straggler are filled)
            const int c = (j - 1) / FP_TILE;
            wp = wplans[(int64_t)a * tiles_per + c];
            st.jc_lo = wp.col_off; st.j_hi = st.jc_lo + wp.m; // tile c starts one checkpoint before column c*FP_TILE
            if (j > st.j_hi || j < lo_ok(st.jc_lo)) { atomicOr(err, 2); done = true; break; }
            continue;
        } else break; // needs a (new) window
        int tag = (int)((w >> (2 * pos)) & 3u);
        if (tag == 0) { atomicOr(err, 2); done = true; break; }
        if (k == 1) {
            int avail = min(pos + 1, j);
            if (!on_plane) avail = min(avail, j - lo_ok(st.jc_lo) + 1); // do not run past the window's usable left edge
            unsigned x = w ^ 0xAAAAAAAAu;
            if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
            const int lowcut = pos + 1 - avail;
            if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
            int steps;
            if (x == 0) steps = avail;
            else {
                const int pnz = (31 - __clz((int)x)) >> 1;
                tag = (int)((w >> (2 * pnz)) & 3u);
                if (tag == 0) { atomicOr(err, 2); done = true; break; }
                steps = pos - pnz + 1;
                k = 3 - tag;
            }
            emit(1, steps); j -= steps; last_op = 1;
            if (on_plane && x == 0 && steps == pos + 1) {
                // the run continues below field 0 of this word: take whole 16-column words while they are all-I, four loads in
                // flight (a 10 kb trailing gap is 600 dependent loads otherwise).  Only on the stored planes, where the lanes of
                // a wave are in this state together; inside windows / tiles the extra control flow costs more than it saves.
                const unsigned *wbase = rowi + pl.rowi_off + (int64_t)(pl.n - i) * pl.words;
                int wi = ((j + steps + G8 - 1) >> 4) - 1;
                bool more = true;
                while (more && wi >= 0 && j >= 16) {
                    unsigned q[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) q[u] = (wi - u >= 0) ? wbase[wi - u] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (more && q[u] == 0xAAAAAAAAu && j >= 16) { emit(1, 16); j -= 16; wi--; }
                        else more = false;
                    }
                }
            }
            continue;
        }
        emit(k, 1);
        last_op = k;
        bool up_exit = false;
        if (k != 1) { up_exit = (li == 0); li = up_exit ? tp.ci - 1 : li - 1; i--; }
        if (k != 2) j--;
        k = 3 - tag;
        if (up_exit && i > 0 && j > 0) { // quirk Q1: restart in the argmax state of the entry cell (i, j)
            int ht;
            if (tail_ok(i, j)) ht = (int)tail_tag(i, j);
            else if (j <= st.jc_lo) { atomicOr(err, 2); done = true; break; } // cannot happen: column jc_lo + 1 is never walked
            else if (j < st.j_hi) { int p2; ht = (int)((load_word<true>(wtrace, wp, 0, i + 1, j + 1 - st.jc_lo, p2) >> (2 * p2)) & 3u); }
            else ht = whcol[wp.hcol_off + i - 1] & 3;
            k = 3 - ht;
        }
    }
    if (done) {
        // Step 4 (affineGap.go:135-139)
        const bool up_exit = (last_op != 1) && ((int64_t)i % tp.ci == 0);
        const bool left_exit = (last_op != 2) && ((int64_t)j % tp.cj == 0);
        if (!up_exit && left_exit) emit(2, i);
        else if (up_exit && !left_exit) emit(1, j);
        flush_run();
        cur_op = -1;
        nops[p] = cnt;
        if (cnt > FP_CAP) atomicOr(err, 8);
        st.status = 1;
    } else {
        // request the window (jc_lo, j] : at least FP_SPAN wide, starting on a checkpoint column (or column 0)
        const int slot = atomicAdd(next_count, 1);
        int jc = j - FP_SPAN;
        jc = jc <= 0 ? 0 : (jc / CKW) * CKW;
        st.j_hi = j; st.jc_lo = jc; st.slot = slot; st.status = 0;
        next_active[slot] = p;
        PairPlan q;
        q.n = pl.n; q.m = j - jc; q.words = (q.m + 15 + 15) / 16; q.strips = 1;
        q.trace_off = (int64_t)slot * FP_WWORDS * QA * G; q.hcol_off = (int64_t)slot * H; q.rowbuf_off = 0; q.dcol_off = (int64_t)slot * G;
        q.src = pl.src; q.col_off = jc; q.ckpt_off = pl.ckpt_off; q.rowi_off = 0;
        next_wplans[slot] = q;
    }
    st.i = i; st.j = j; st.k = k; st.last_op = last_op; st.cur_op = cur_op; st.cnt = cnt; st.cur_run = cur_run; st.li = li;
    states[p] = st;
}

// stragglers (the path keeps needing windows, e.g. a long gap on a row without a stored plane): every remaining
// column of such a pair is re-filled as independent FP_TILE-column tiles from the column checkpoints -- one launch,
// a tile deep instead of a matrix deep -- and fp_walk_kernel<false, true> finishes the walk through them.
__global__ __launch_bounds__(256) void fp_straggler_plans_kernel(const PairPlan *__restrict__ plans, const int *__restrict__ active, int n_active,
                                                                  int tiles_per, const FpState *__restrict__ states, PairPlan *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_active * tiles_per) return;
    const int a = x / tiles_per, c = x - a * tiles_per;
    const int p = active[a];
    const PairPlan pl = plans[p];
    const int j_cur = states[p].j;
    PairPlan q = pl;
    const int lo = c * FP_TILE, lo2 = max(0, lo - CKW); // start one checkpoint early: the first re-filled column has no usable tags
    q.m = (j_cur > lo) ? min(FP_TILE, j_cur - lo) + (lo - lo2) : 0;
    q.words = (q.m + 15 + 15) / 16; q.strips = q.m > 0 ? 1 : 0;
    q.trace_off = (int64_t)x * FP_TWORDS * QA * G; q.hcol_off = (int64_t)x * H; q.rowbuf_off = 0; q.dcol_off = (int64_t)x * G;
    q.src = pl.src; q.col_off = lo2;
    q.rowi_off = (x == 0) ? tiles_per : 0; // plan 0 carries tiles_per for the walk kernel
    out[x] = q;
}

__global__ __launch_bounds__(256) void fp_compact_kernel(int n_pairs, const FpState *__restrict__ states, const gnx_cigar *__restrict__ stage, const int64_t *__restrict__ nops,
                                                          const int64_t *__restrict__ ops_off, gnx_cigar *__restrict__ ops, int64_t ops_capacity,
                                                          int *__restrict__ err) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const int64_t cnt = nops[p], base = ops_off[p];
    if (base + cnt > ops_capacity) { atomicOr(err, 4); return; }
    const int64_t m = cnt < FP_CAP ? cnt : FP_CAP;
    for (int64_t x = 0; x < m; x++) ops[base + (cnt - 1 - x)] = stage[(int64_t)p * FP_CAP + x];
}

// ------------------------------------------------------------------------------------------------------
// N1: per-cell score matrices for the chunk / multiple-alignment variants, one thread per (chunk) cell, written
// column-major as 4*score.  A "group" is an alignment block: nseq sequences of len bases, sequence-major.
//   pairwise (AffineGapChunk):      cell = sum_k scores[a[i*c+k]][b[j*c+k]]                       (ungapped.go:7-13)
//   groups (multipleAffineGap*):    cell = sum_k scoreColumnMatch(column i*c+k, column j*c+k)     (multiAlign.go:82-110)
//     scoreColumnMatch = (sum over sequence pairs, lower case folded, gap columns skipped) / count, Go integer division
// ------------------------------------------------------------------------------------------------------
struct GroupDesc { int64_t off; int32_t nseq; int32_t len; };
struct ScorePair { int64_t a_off, b_off; int32_t a_nseq, b_nseq, a_len, b_len; int32_t nc, mc; int64_t s_off, s_pitch; };

__global__ __launch_bounds__(256) void score_matrix_kernel(const ScorePair *__restrict__ sp, const uint8_t *__restrict__ bases, KParams kp, int chunk,
                                                           int groups, int *__restrict__ smat, int *__restrict__ err) {
    const ScorePair q = sp[blockIdx.y];
    const int64_t cells = (int64_t)q.nc * q.mc;
    for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < cells; cell += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(cell % q.nc), j = (int)(cell / q.nc);
        int64_t total = 0;
        for (int k = 0; k < chunk; k++) {
            const int64_t ac = (int64_t)i * chunk + k, bc = (int64_t)j * chunk + k;
            if (!groups) {
                const int a = bases[q.a_off + ac], b = bases[q.b_off + bc];
                if (a >= 5 || b >= 5) { atomicOr(err, 1); continue; }
                total += kp.sc4[a * 5 + b] / 4;
            } else {
                int64_t sum = 0, count = 0;
                for (int x = 0; x < q.a_nseq; x++) {
                    int a = bases[q.a_off + (int64_t)x * q.a_len + ac];
                    if (a >= 5 && a <= 9) a -= 5;
                    for (int y = 0; y < q.b_nseq; y++) {
                        int b = bases[q.b_off + (int64_t)y * q.b_len + bc];
                        if (b >= 5 && b <= 9) b -= 5;
                        if (a != 10 && b != 10) {
                            if (a >= 5 || b >= 5) { atomicOr(err, 1); continue; }
                            sum += kp.sc4[a * 5 + b] / 4;
                            count++;
                        }
                    }
                }
                if (count == 0) { atomicOr(err, 16); continue; } // Go: integer divide by zero
                total += sum / count;
            }
        }
        smat[q.s_off + (int64_t)j * q.s_pitch + i] = (int)(4 * total);
    }
}

__global__ __launch_bounds__(256) void scale_runs_kernel(gnx_cigar *__restrict__ ops, int64_t total, int64_t factor) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < total) ops[x].run_length *= factor; // expandCigarRunLength, affineGap_highMem.go:91-95
}

// exclusive scan of nops[0..n) + carry[0] -> off[0..n], off[n]; carry[0] = off[n] afterwards.  One block.
__global__ __launch_bounds__(1024) void scan_kernel(const int64_t *__restrict__ nops, int n, int64_t *__restrict__ off, int64_t *__restrict__ carry) {
    // exclusive scan of the run counts, 1024 elements per round: wave-level scans by __shfl_up, then the 16 wave totals
    // (a one-pass version with a contiguous slice per thread measured slower: 227 us vs 139 us per 100 k elements)
    __shared__ int64_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t base = carry[0];
    for (int start = 0; start < n; start += 1024) {
        const int idx = start + threadIdx.x;
        const int64_t v = idx < n ? nops[idx] : 0;
        int64_t sum = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int64_t t = __shfl_up(sum, d, 64); if (lane >= d) sum += t; }
        if (lane == 63) wsum[wave] = sum;
        __syncthreads();
        if (wave == 0) {
            int64_t t = lane < 16 ? wsum[lane] : 0;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) { const int64_t u = __shfl_up(t, d, 64); if (lane >= d) t += u; }
            if (lane < 16) wsum[lane] = t;
        }
        __syncthreads();
        if (idx < n) off[idx] = base + (wave > 0 ? wsum[wave - 1] : 0) + sum - v;
        base += wsum[15];
        __syncthreads();
    }
    if (threadIdx.x == 0) { off[n] = base; carry[0] = base; }
}

// ------------------------------------------------------------------------------------------------------
// Traceback of the gsw seed-extension DPs (search.go:252-275, 298-320).  One lane per pair, runs in TRACEBACK order (the
// reference appends them that way; its callers reverse).  Ops use the GNX_COL_* codes (M/I/D).
//   LEFT : from (n, m) while the cell value is > 0.  Values are not stored: the walk rebuilds them from the final value
//          (an unclamped cell is its predecessor plus the score of the move; a clamped cell is 0 and ends the walk).
//   RIGHT: from the first row-major maximum (row scan of the keys the fill kernel left in hcol) back to (0, 0).
// ------------------------------------------------------------------------------------------------------
template <bool RIGHT, bool WRITE>
__global__ __launch_bounds__(64) void gsw_traceback_kernel(const PairPlan *__restrict__ plans, int n_pairs, const uint4 *__restrict__ trace,
                                                           const int *__restrict__ hcol,
                                                           const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                           const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start, KParams kp,
                                                           int64_t *__restrict__ score_out, int2 *__restrict__ endpos,
                                                           int64_t *__restrict__ nops, const int64_t *__restrict__ ops_off,
                                                           gnx_cigar *__restrict__ ops, int64_t ops_capacity, int *__restrict__ err) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    const uint8_t *bp = b_buf + b_start[p];
    int i = pl.n, j = pl.m;
    int cur = 0; // LEFT: value of the current cell
    if (RIGHT) {
        int bestv = 0, bi = 0, bj = 0;
        if (pl.m >= 1) {
            for (int r = 1; r <= pl.n; r++) {
                const int key = hcol[pl.hcol_off + r - 1];
                const int v = key >> 12;
                if (v > bestv) { bestv = v; bi = r; bj = 4095 - (key & 4095); }
            }
        }
        i = bi; j = bj;
        if (!WRITE) { score_out[p] = bestv; endpos[p] = make_int2(bi, bj); }
    } else {
        if (pl.n >= 1 && pl.m >= 1) cur = hcol[pl.hcol_off + pl.n - 1] >> 2;
        if (!WRITE) score_out[p] = cur;
    }
    const int64_t base = WRITE ? ops_off[p] : 0;
    int64_t cnt = 0, run = 0;
    int cur_op = -1;
    auto emit = [&](int op, int64_t len) {
        if (op == cur_op) { run += len; return; }
        if (cur_op >= 0) {
            if (WRITE) { if (base + cnt < ops_capacity) { gnx_cigar c; c.run_length = run; c.op = (uint8_t)cur_op; for (int z = 0; z < 7; z++) c._pad[z] = 0; ops[base + cnt] = c; } else atomicOr(err, 4); }
            cnt++;
        }
        cur_op = op; run = len;
    };
    while (RIGHT ? (i > 0 || j > 0) : (cur > 0)) {
        if (RIGHT && i == 0) { emit(GNX_COL_I, j); j = 0; break; } // trace[0][j] = 'I'
        if (RIGHT && j == 0) { emit(GNX_COL_D, i); i = 0; break; } // trace[i][0] = 'D'
        if (i < 1 || j < 1) { atomicOr(err, 2); break; }
        int pos;
        const unsigned w = load_word<false>(trace, pl, 0, i, j, pos);
        const int tag = (int)((w >> (2 * pos)) & 3u);
        if (tag == 3) { emit(GNX_COL_M, 1); if (!RIGHT) cur -= kp.sc4[min((int)ap[i - 1], 4) * 5 + min((int)bp[j - 1], 4)] >> 2; i--; j--; }
        else if (tag == 2) { emit(GNX_COL_I, 1); if (!RIGHT) cur -= kp.g4 >> 2; j--; }
        else if (tag == 1) { emit(GNX_COL_D, 1); if (!RIGHT) cur -= kp.g4 >> 2; i--; }
        else { atomicOr(err, 2); break; }
    }
    emit(-2, 0); // flush
    if (!WRITE) { nops[p] = cnt; if (!RIGHT) endpos[p] = make_int2(i, j); }
}

// ------------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";
void set_err(const char *fmt, const char *a = "", long long b = 0) { snprintf(g_err, sizeof(g_err), fmt, a, b); }

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            snprintf(g_err, sizeof(g_err), "HIP error %s at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__); \
            return GNX_EDEVICE;                                                                        \
        }                                                                                              \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return GNX_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; set_err("device allocation of %s%lld bytes failed", "", (long long)bytes); return GNX_ENOMEM; }
            want = bytes;
        }
        cap = want;
        return GNX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Ctx {
    std::mutex mu;
    bool inited = false;
    int device = -1;
    int64_t ws_limit = 0;
    hipStream_t own_stream = nullptr;
    DevBuf trace, hcol, rowbuf, dcol, plans, nops, misc;
    DevBuf fp_tail, fp_rowi, fp_ckpt, fp_states, fp_stage, fp_wplans[2], fp_active[2], fp_thcol, fp_ttrace;
    DevBuf in_a, in_b, in_as, in_al, in_bs, in_bl, out_score, out_off, out_ops, out_end, sc_pairs, sc_mat, sc_err;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    gnx_timing timing = {};
};
Ctx g_ctx;

int ensure_init() {
    if (g_ctx.inited) return GNX_OK;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { set_err("no HIP device available (this library has no CPU fallback)%s", ""); return GNX_EDEVICE; }
    int dev = 0;
    const char *lr = getenv("LOCAL_RANK");
    if (lr && *lr) dev = atoi(lr) % cnt;
    HIPCHK(hipSetDevice(dev));
    g_ctx.device = dev;
    HIPCHK(hipStreamCreate(&g_ctx.own_stream));
    for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&g_ctx.ev[i]));
    if (g_ctx.ws_limit <= 0) {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        g_ctx.ws_limit = (int64_t)std::min<size_t>(fr / 2, (size_t)64 << 30);
    }
    g_ctx.inited = true;
    return GNX_OK;
}

int check_params(const gnx_params *p, KParams &kp, TbParams &tp, bool &affine, bool &local, bool &lowmem) {
    if (!p) { set_err("null params%s", ""); return GNX_EINVAL; }
    affine = (p->mode == GNX_AFFINE_GAP || p->mode == GNX_AFFINE_GAP_HIGHMEM || p->mode == GNX_AFFINE_GAP_LOCAL);
    const bool cst = (p->mode == GNX_CONST_GAP || p->mode == GNX_CONST_GAP_HIGHMEM);
    if (!affine && !cst) { set_err("unknown mode %s%lld", "", (long long)p->mode); return GNX_EINVAL; }
    local = (p->mode == GNX_AFFINE_GAP_LOCAL);
    lowmem = (p->mode == GNX_AFFINE_GAP || p->mode == GNX_CONST_GAP);
    if (lowmem && (p->checkersize_i < 1 || p->checkersize_j < 1)) { set_err("checkersize must be >= 1%s", ""); return GNX_EINVAL; }
    const int64_t lim = (int64_t)1 << 26;
    for (int x = 0; x < 25; x++) if (p->scores[x] > lim || p->scores[x] < -lim) { set_err("score out of int32 kernel range%s", ""); return GNX_ERANGE; }
    if (p->gap_open > lim || p->gap_open < -lim || (affine && (p->gap_extend > lim || p->gap_extend < -lim))) { set_err("gap penalty out of int32 kernel range%s", ""); return GNX_ERANGE; }
    for (int x = 0; x < 25; x++) kp.sc4[x] = (int)(4 * p->scores[x]);
    kp.o4 = (int)(4 * p->gap_open);
    kp.e4 = affine ? (int)(4 * p->gap_extend) : 0;
    kp.oe4 = kp.o4 + kp.e4;
    kp.d00_4 = local ? 0 : kp.o4;
    kp.ecol4 = local ? 0 : kp.e4;
    kp.g4 = kp.o4;
    tp.ci = lowmem ? p->checkersize_i : ((int64_t)1 << 62);
    tp.cj = lowmem ? p->checkersize_j : ((int64_t)1 << 62);
    tp.d00 = local ? 0 : p->gap_open;
    tp.ecol = local ? 0 : p->gap_extend;
    tp.gap_open = p->gap_open;
    tp.gap_extend = affine ? p->gap_extend : 0;
    tp.affine = affine ? 1 : 0;
    return GNX_OK;
}

int64_t max_abs_pen(const gnx_params *p, bool affine) {
    int64_t mx = 0;
    for (int x = 0; x < 25; x++) mx = std::max<int64_t>(mx, llabs((long long)p->scores[x]));
    if (affine) mx = std::max<int64_t>(mx, llabs((long long)(p->gap_open + p->gap_extend)));
    mx = std::max<int64_t>(mx, llabs((long long)p->gap_open));
    if (affine) mx = std::max<int64_t>(mx, llabs((long long)p->gap_extend));
    return mx;
}

// Fast path for batches of short-alpha global affine alignments (see fp_walk_kernel).  Returns GNX_OK, an error,
// or -1 when the batch should go through the general path after all (workspace too small / staging overflow).
// `first` = first sub-batch of a call: later sub-batches keep the error flags and the CIGAR offset carry.
int run_device_fp(const KParams &kp, const TbParams &tp, int64_t n_pairs,
                  const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
                  const int64_t *h_alen, const int64_t *h_blen, int rows_per_lane,
                  int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off,
                  int64_t *out_total, hipStream_t stream, bool first) {
    // rows_per_lane: 19 (every n <= 152) or 20 (n <= 160) -> fp_sweep_kernel<19 / 20>
    Ctx &c = g_ctx;
    int rc;
    const int np = (int)n_pairs;
    std::vector<PairPlan> plans((size_t)n_pairs);
    int64_t roff = 0, coff = 0, cells = 0, m_maxb = 1;
    for (int64_t p = 0; p < n_pairs; p++) {
        PairPlan &pl = plans[(size_t)p];
        const int64_t n = h_alen[p], m = h_blen[p];
        pl.n = (int32_t)n; pl.m = (int32_t)m; pl.words = (int32_t)((m + 15 + 15) / 16); pl.strips = 1;
        pl.trace_off = 0; pl.hcol_off = p; pl.rowbuf_off = 0; pl.dcol_off = 0;
        pl.src = (int32_t)p; pl.col_off = 0; pl.ckpt_off = coff; pl.rowi_off = roff; pl.s_off = 0; pl.s_pitch = 0;
        roff += (int64_t)FP_PLANES * pl.words; coff += ((m - 1) / CKW) * n; cells += n * m;
        m_maxb = std::max(m_maxb, m);
    }
    const size_t wtrace_b = (size_t)np * FP_WWORDS * QA * G * 16;
    const size_t need = wtrace_b + (size_t)coff * 8 + (size_t)roff * 4 + (size_t)np * (FP_CAP * sizeof(gnx_cigar) + sizeof(FpState) + 3 * sizeof(PairPlan) + H * 4 + G * 4 + 32);
    if ((int64_t)need > c.ws_limit) { if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] working set %zu B exceeds the workspace limit -> general path\n", need); return -1; }
    if ((rc = c.trace.ensure(wtrace_b))) return rc;
    if ((rc = c.hcol.ensure((size_t)np * (H + 1) * 4))) return rc;   // [0,np) h(n,m) of the forward sweep, then the window hcol slots
    if ((rc = c.dcol.ensure((size_t)np * G * 4))) return rc;
    if ((rc = c.plans.ensure((size_t)np * sizeof(PairPlan)))) return rc;
    if ((rc = c.nops.ensure((size_t)np * 8))) return rc;
    if ((rc = c.misc.ensure(64))) return rc;
    if ((rc = c.fp_rowi.ensure((size_t)std::max<int64_t>(roff, 1) * 4))) return rc;
    if ((rc = c.fp_tail.ensure((size_t)np * 4))) return rc;
    if ((rc = c.fp_ckpt.ensure((size_t)std::max<int64_t>(coff, 1) * 8))) return rc;
    if ((rc = c.fp_states.ensure((size_t)np * sizeof(FpState)))) return rc;
    if ((rc = c.fp_stage.ensure((size_t)np * FP_CAP * sizeof(gnx_cigar)))) return rc;
    for (int x = 0; x < 2; x++) {
        if ((rc = c.fp_wplans[x].ensure((size_t)np * sizeof(PairPlan)))) return rc;
        if ((rc = c.fp_active[x].ensure((size_t)np * 4))) return rc;
    }
    if (!c.ev[4]) for (int i = 4; i < 8; i++) HIPCHK(hipEventCreate(&c.ev[i]));
    int *d_err = reinterpret_cast<int *>(c.misc.p);
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    int *d_cnt = reinterpret_cast<int *>(reinterpret_cast<char *>(c.misc.p) + 32); // window-request counters (ping-pong)
    if (first) HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, stream));
    else HIPCHK(hipMemsetAsync(d_cnt, 0, 16, stream));
    HIPCHK(hipMemcpyAsync(c.plans.p, plans.data(), (size_t)np * sizeof(PairPlan), hipMemcpyHostToDevice, stream));
    const PairPlan *dpl = reinterpret_cast<const PairPlan *>(c.plans.p);
    int *d_hfwd = reinterpret_cast<int *>(c.hcol.p);
    int *d_whcol = d_hfwd + np;
    unsigned *d_rowi = reinterpret_cast<unsigned *>(c.fp_rowi.p);
    unsigned *d_tail = reinterpret_cast<unsigned *>(c.fp_tail.p);
    int2 *d_ckpt = reinterpret_cast<int2 *>(c.fp_ckpt.p);
    FpState *d_st = reinterpret_cast<FpState *>(c.fp_states.p);
    gnx_cigar *d_stage = reinterpret_cast<gnx_cigar *>(c.fp_stage.p);
    int64_t *d_nops = reinterpret_cast<int64_t *>(c.nops.p);
    PairPlan *d_wpl[2] = {reinterpret_cast<PairPlan *>(c.fp_wplans[0].p), reinterpret_cast<PairPlan *>(c.fp_wplans[1].p)};
    int *d_act[2] = {reinterpret_cast<int *>(c.fp_active[0].p), reinterpret_cast<int *>(c.fp_active[1].p)};
    const dim3 blockF(64), blockT(64);
    // window rounds continue while they pay: a fixed number with GNX_FP_MAXIT, else while more than 4 % of the pairs are waiting
    // (a round has a fixed latency of ~0.2 ms; the few pairs left over go to the tile re-fill, whose cost is per pair)
    const int max_it = getenv("GNX_FP_MAXIT") ? atoi(getenv("GNX_FP_MAXIT")) : -1;
    const int tiles_per = (int)((m_maxb + FP_TILE - 1) / FP_TILE);
    double refill_ms = 0;

    auto forward = [&](int p0, int cnt, hipStream_t st) -> int {
        const dim3 grid8((unsigned)((cnt + G8 - 1) / G8));
        if (rows_per_lane == 19) hipLaunchKernelGGL(fp_sweep_kernel<19>, grid8, blockF, 0, st, dpl + p0, cnt, d_a, d_as, d_b, d_bs, kp, d_hfwd, d_ckpt, d_rowi, d_tail, d_err);
        else hipLaunchKernelGGL(fp_sweep_kernel<20>, grid8, blockF, 0, st, dpl + p0, cnt, d_a, d_as, d_b, d_bs, kp, d_hfwd, d_ckpt, d_rowi, d_tail, d_err);
        HIPCHK(hipGetLastError());
        return GNX_OK;
    };
    // walk / window stages of the pairs [p0, p0+cnt) on stream `st`; their window slots are [p0, p0+cnt) as well
    auto post = [&](int p0, int cnt, hipStream_t st, int *cnt2, hipEvent_t e1, hipEvent_t e2) -> int {
        uint4 *wtr = reinterpret_cast<uint4 *>(c.trace.p) + (int64_t)p0 * FP_WWORDS * QA * G;
        int *whc = d_whcol + (int64_t)p0 * H;
        unsigned *wdc = reinterpret_cast<unsigned *>(c.dcol.p) + (int64_t)p0 * G;
        int cur = 0, n_act = 0, it = 0;
        float f = 0;
        hipLaunchKernelGGL(fp_walk_kernel<true>, dim3((unsigned)((cnt + 63) / 64)), blockT, 0, st, dpl, (const int *)nullptr, cnt, d_st, d_hfwd, d_rowi, d_tail,
                           (const PairPlan *)nullptr, wtr, whc, tp, d_stage, d_score, d_nops, d_act[0] + p0, cnt2, d_wpl[0] + p0, d_err, p0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&n_act, cnt2, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        while (n_act > 0 && (max_it >= 0 ? it < max_it : (it == 0 || (int64_t)n_act * 25 > cnt) && it < 16)) {
            const int nxt = cur ^ 1;
            HIPCHK(hipMemsetAsync(cnt2 + nxt, 0, 4, st));
            HIPCHK(hipEventRecord(e1, st));
            hipLaunchKernelGGL((fill_affine_kernel<false, false, false, true, true>), dim3((unsigned)((n_act + 3) / 4)), blockF, 0, st, d_wpl[cur] + p0, n_act, d_a, d_as, d_b, d_bs, kp,
                               wtr, whc, (int2 *)nullptr, wdc, d_ckpt, d_err);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(e2, st));
            hipLaunchKernelGGL(fp_walk_kernel<false>, dim3((unsigned)((n_act + 63) / 64)), blockT, 0, st, dpl, d_act[cur] + p0, n_act, d_st, d_hfwd, d_rowi, d_tail,
                               d_wpl[cur] + p0, wtr, whc, tp, d_stage, d_score, d_nops, d_act[nxt] + p0, cnt2 + nxt, d_wpl[nxt] + p0, d_err, 0);
            HIPCHK(hipGetLastError());
            int n_next = 0;
            HIPCHK(hipMemcpyAsync(&n_next, cnt2 + nxt, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipEventElapsedTime(&f, e1, e2));
            refill_ms += f;
            it++;
            if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] pairs [%d,%d) round %d: %d windows re-filled in %.3f ms, %d pairs continue\n", p0, p0 + cnt, it, n_act, f, n_next);
            n_act = n_next;
            cur = nxt;
        }
        if (n_act > 0) { // stragglers: all their remaining columns as independent tiles, one launch
            const int n_strag = n_act;
            const int64_t n_tiles = (int64_t)n_strag * tiles_per;
            const size_t tb = (size_t)n_tiles * FP_TWORDS * QA * G * 16;
            if ((int64_t)tb > c.ws_limit || n_tiles > 0x3fffffff) { if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] %lld straggler tiles exceed the workspace limit -> general path\n", (long long)n_tiles); return -1; }
            int rc2;
            if ((rc2 = c.rowbuf.ensure((size_t)n_tiles * sizeof(PairPlan)))) return rc2; // tile plans (rowbuf is unused on this path)
            if ((rc2 = c.fp_thcol.ensure((size_t)n_tiles * (H + G) * 4))) return rc2;
            if ((rc2 = c.fp_ttrace.ensure(tb))) return rc2;
            PairPlan *tpl = reinterpret_cast<PairPlan *>(c.rowbuf.p);
            int *thc = reinterpret_cast<int *>(c.fp_thcol.p);
            unsigned *tdc = reinterpret_cast<unsigned *>(thc + n_tiles * H);
            uint4 *ttr = reinterpret_cast<uint4 *>(c.fp_ttrace.p);
            hipLaunchKernelGGL(fp_straggler_plans_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, st, dpl, d_act[cur] + p0, n_strag, tiles_per, d_st, tpl);
            HIPCHK(hipEventRecord(e1, st));
            hipLaunchKernelGGL((fill_affine_kernel<false, false, false, true, true>), dim3((unsigned)((n_tiles + 3) / 4)), blockF, 0, st, tpl, (int)n_tiles, d_a, d_as, d_b, d_bs, kp,
                               ttr, thc, (int2 *)nullptr, tdc, d_ckpt, d_err);
            HIPCHK(hipEventRecord(e2, st));
            HIPCHK(hipMemsetAsync(cnt2, 0, 8, st));
            hipLaunchKernelGGL((fp_walk_kernel<false, true>), dim3((unsigned)((n_strag + 63) / 64)), blockT, 0, st, dpl, d_act[cur] + p0, n_strag, d_st, d_hfwd, d_rowi, d_tail,
                               tpl, ttr, thc, tp, d_stage, d_score, d_nops, d_act[cur ^ 1] + p0, cnt2, d_wpl[cur ^ 1] + p0, d_err, 0);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipEventElapsedTime(&f, e1, e2));
            refill_ms += f;
            if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] pairs [%d,%d): %d stragglers: %lld tiles of %d columns re-filled in %.3f ms\n", p0, p0 + cnt, n_strag, (long long)n_tiles, FP_TILE, f);
        }
        return GNX_OK;
    };

    HIPCHK(hipEventRecord(c.ev[0], stream));
    if ((rc = forward(0, np, stream))) return rc;
    HIPCHK(hipEventRecord(c.ev[1], stream));
    if ((rc = post(0, np, stream, d_cnt, c.ev[4], c.ev[5]))) return rc;
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, d_nops, np, d_ops_off, d_carry);
    hipLaunchKernelGGL(fp_compact_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, np, d_st, d_stage, d_nops, d_ops_off, d_ops, ops_capacity, d_err);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c.ev[3], stream));
    int h_misc[16];
    HIPCHK(hipMemcpyAsync(h_misc, c.misc.p, 64, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float tot = 0, fa = 0;
    HIPCHK(hipEventElapsedTime(&tot, c.ev[0], c.ev[3]));
    HIPCHK(hipEventElapsedTime(&fa, c.ev[0], c.ev[1]));
    const double forward_ms = (double)fa;
    if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] forward sweep (fp_sweep_kernel<%d>): %d pairs %.3f ms\n", rows_per_lane, np, fa);
    if (first) c.timing = gnx_timing{};
    c.timing.fill_ms += forward_ms + refill_ms; c.timing.traceback_ms += std::max(0.0, (double)tot - fa - refill_ms); c.timing.total_ms += tot;
    c.timing.cells += cells; c.timing.n_launches += 1; c.timing.trace_bytes += (int64_t)coff * 8 + (int64_t)roff * 4;
    c.timing.dominant_ms += forward_ms; c.timing.dominant_launches += 1; c.timing.fast_path = 1;
    int64_t total;
    memcpy(&total, reinterpret_cast<char *>(h_misc) + 16, 8);
    if (out_total) *out_total = total;
    const int ef = h_misc[0];
    if (ef & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (ef & 2) { set_err("unexpected traceback%s", ""); return GNX_ETRACE; }
    if (ef & 8) { if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] a CIGAR has more than %d runs -> general path\n", FP_CAP); return -1; } // redo on the general path
    if (ef & 4) { set_err("CIGAR buffer too small: need %s%lld elements", "", (long long)total); return GNX_ECAPACITY; }
    return GNX_OK;
}

// The device flow shared by all entry points.  All pointers are device pointers except h_*.
int run_device(const gnx_params *prm, int64_t n_pairs,
               const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
               const int64_t *h_alen, const int64_t *h_blen,
               int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off,
               int64_t *out_total, hipStream_t stream, const int *d_smat = nullptr, const int64_t *h_soff = nullptr,
               int gsw = 0, int2 *d_endpos = nullptr) {
    // gsw: 1 / 2 = LeftDynamicAln / RightDynamicAln of the graph aligner (constant-gap kernels with GSW = 1 / 2 and their own
    // traceback; d_endpos receives the (i, j) the reference returns); prm->mode must be GNX_CONST_GAP_HIGHMEM
    // d_smat / h_soff: explicit per-cell score matrices (SCORED kernels, N1 variants); the sequences are then unused
    Ctx &c = g_ctx;
    KParams kp; TbParams tp; bool affine, local, lowmem;
    int rc = check_params(prm, kp, tp, affine, local, lowmem);
    if (rc) return rc;
    if (d_smat && (!affine || local || lowmem)) { set_err("scored mode needs AffineGap_highMem semantics%s", ""); return GNX_EINVAL; }
    if (gsw && (affine || lowmem || !d_endpos || prm->gap_open > 0)) { set_err("gsw extension needs ConstGap_highMem parameters with gapPen <= 0%s", ""); return GNX_EINVAL; }
    if (n_pairs < 0 || n_pairs > 0x7ffffff0) { set_err("bad n_pairs%s", ""); return GNX_EINVAL; }
    c.timing = gnx_timing{};
    if (n_pairs == 0) {
        int64_t z = 0;
        HIPCHK(hipMemcpyAsync(d_ops_off, &z, 8, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (out_total) *out_total = 0;
        return GNX_OK;
    }
    const int64_t maxpen = max_abs_pen(prm, affine);
    // ---- fast path: every alpha fits one strip, long beta, global affine with gapOpen <= 0 ----
    {
        const char *fpenv = getenv("GNX_FASTPATH");
        bool fp = affine && !local && !d_smat && prm->gap_open <= 0 && !(fpenv && fpenv[0] == '0');
        // fp_sweep_kernel keeps an int16 profile of 4*(s - 2e); its padding rows need 4*|gapOpen| well inside int16
        if (prm->gap_open <= -8000) fp = false;
        for (int x = 0; x < 25; x++) { const int64_t v = 4 * (prm->scores[x] - 2 * prm->gap_extend); if (v > 32767 || v < -32000) fp = false; }
        int64_t n_hi = 0;
        for (int64_t p = 0; fp && p < n_pairs; p++) {
            const int64_t n = h_alen[p], m = h_blen[p];
            if (n < 1 || n > H || m < (fpenv && fpenv[0] == '2' ? 1 : 768) || m > 0x3fffffff) fp = false;
            else if ((n + m + 2) * std::max<int64_t>(maxpen, 1) >= ((int64_t)1 << 27)) fp = false;
            n_hi = std::max(n_hi, n);
        }
        const int rows_per_lane = n_hi <= 19 * G8 ? 19 : 20;
        if (fp) {
            // sub-batches whose fast-path working set (checkpoints, planes, window slots, staging) fits the workspace
            const size_t fixed = (size_t)FP_WWORDS * QA * G * 16 + FP_CAP * sizeof(gnx_cigar) + sizeof(FpState) + 3 * sizeof(PairPlan) + H * 4 + G * 4 + 64;
            std::vector<int64_t> cb{0};
            size_t acc_b = 0;
            const size_t budget = (size_t)(c.ws_limit - c.ws_limit / 8);
            for (int64_t p = 0; p < n_pairs; p++) {
                const size_t b = fixed + (size_t)((h_blen[p] - 1) / CKW) * h_alen[p] * 8 + (size_t)FP_PLANES * ((h_blen[p] + 30) / 16) * 4;
                if (b > budget) { fp = false; break; }
                if (acc_b + b > budget) { cb.push_back(p); acc_b = 0; }
                acc_b += b;
            }
            cb.push_back(n_pairs);
            rc = -1;
            for (size_t ch = 0; fp && ch + 1 < cb.size(); ch++) {
                const int64_t b = cb[ch], e = cb[ch + 1];
                rc = run_device_fp(kp, tp, e - b, d_a, d_as + b, d_b, d_bs + b, h_alen + b, h_blen + b, rows_per_lane, d_score + b, d_ops, ops_capacity,
                                   d_ops_off + b, out_total, stream, ch == 0);
                if (rc != GNX_OK) break;
            }
            if (fp && rc != -1) return rc;
        }
    }
    // ---- plan ----
    bool p16 = true; // 4*score fits a signed 16-bit profile entry
    for (int x = 0; x < 25; x++) if (prm->scores[x] > 8191 || prm->scores[x] < -8192) p16 = false;
    if (!getenv("GNX_FORCE_P16")) p16 = false; // int32 profile: plain 2-cycle VGPR add instead of a 4-cycle SDWA add
    bool hform = affine && prm->gap_open <= 0;
    if (getenv("GNX_NO_HFORM")) hform = false;
    const int Q = affine ? QA : QC;
    std::vector<PairPlan> plans((size_t)n_pairs);
    std::vector<int64_t> chunk_begin;
    int64_t cells = 0;
    {
        int64_t toff = 0, hoff = 0, roff = 0, doff = 0;
        chunk_begin.push_back(0);
        const int64_t trace_limit_u4 = std::max<int64_t>(c.ws_limit / 16, 1);
        for (int64_t p = 0; p < n_pairs; p++) {
            const int64_t n = h_alen[p], m = h_blen[p];
            if (n < 0 || m < 0 || n > 0x3fffffff || m > 0x3fffffff) { set_err("bad sequence length at pair %s%lld", "", (long long)p); return GNX_EINVAL; }
            if (lowmem && prm->checkersize_i != prm->checkersize_j && n > prm->checkersize_i) {
                // the reference indexes its saved columns with checkersize_j where checkersize_i is meant
                // (affineGap.go:252-254, constGap.go:207): undefined for non-square tiles once n > checkersize_i
                set_err("non-square checkerboards with n > checkersize_i are undefined in the reference (pair %s%lld)", "", (long long)p); return GNX_EINVAL;
            }
            if (lowmem && (n < 1 || m < 1)) { set_err("empty sequence at pair %s%lld: the reference never terminates on it", "", (long long)p); return GNX_EEMPTY; }
            if ((n + m + 2) * std::max<int64_t>(maxpen, 1) >= ((int64_t)1 << 27)) { set_err("pair %s%lld exceeds the int32 DP range", "", (long long)p); return GNX_ERANGE; }
            if (gsw == 2 && (n > 4095 || m > 4095 || (n + m + 2) * std::max<int64_t>(maxpen, 1) >= ((int64_t)1 << 19))) {
                set_err("pair %s%lld exceeds the range of the packed (score, column) maximum of RightDynamicAln", "", (long long)p); return GNX_ERANGE;
            }
            PairPlan &pl = plans[(size_t)p];
            pl.n = (int32_t)n; pl.m = (int32_t)m;
            pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; // src is chunk-relative, set below
            pl.strips = (m > 0) ? (int32_t)((n + H - 1) / H) : 0;
            pl.words = (int32_t)((m + 15 + 15) / 16);
            const int64_t tsz = (int64_t)pl.strips * pl.words * Q * G;
            if (tsz > trace_limit_u4) { set_err("pair %s%lld needs more direction-matrix workspace than the limit", "", (long long)p); return GNX_ENOMEM; }
            if (toff + tsz > trace_limit_u4) { // start a new chunk (keep chunks 4-aligned so waves stay whole)
                int64_t cb = p & ~(int64_t)3;
                if (cb <= chunk_begin.back()) cb = p;
                // re-plan the pairs moved into the new chunk
                toff = 0; hoff = 0; roff = 0; doff = 0;
                for (int64_t q2 = cb; q2 < p; q2++) {
                    PairPlan &pq = plans[(size_t)q2];
                    pq.trace_off = toff; pq.hcol_off = hoff; pq.rowbuf_off = roff; pq.dcol_off = doff;
                    toff += (int64_t)pq.strips * pq.words * Q * G; hoff += pq.n; roff += (pq.strips > 1) ? pq.m + 1 : 0; doff += (int64_t)pq.strips * G;
                }
                chunk_begin.push_back(cb);
            }
            pl.trace_off = toff; pl.hcol_off = hoff; pl.rowbuf_off = roff; pl.dcol_off = doff;
            if (h_soff) { pl.s_off = h_soff[p]; pl.s_pitch = (int64_t)std::max<int32_t>(pl.strips, 1) * H; }
            toff += tsz; hoff += n; roff += (pl.strips > 1) ? m + 1 : 0; doff += (int64_t)pl.strips * G;
            cells += n * m;
        }
        chunk_begin.push_back(n_pairs);
        for (size_t ch = 0; ch + 1 < chunk_begin.size(); ch++)
            for (int64_t p = chunk_begin[ch]; p < chunk_begin[ch + 1]; p++) plans[(size_t)p].src = (int32_t)(p - chunk_begin[ch]);
    }
    // workspace sizes = max over chunks
    int64_t max_t = 1, max_h = 1, max_r = 1, max_d = 1;
    for (size_t ch = 0; ch + 1 < chunk_begin.size(); ch++) {
        int64_t t = 0, h = 0, r = 0, d = 0;
        for (int64_t p = chunk_begin[ch]; p < chunk_begin[ch + 1]; p++) {
            const PairPlan &pl = plans[(size_t)p];
            t += (int64_t)pl.strips * pl.words * Q * G; h += pl.n; r += (pl.strips > 1) ? pl.m + 1 : 0; d += (int64_t)pl.strips * G;
        }
        max_t = std::max(max_t, t); max_h = std::max(max_h, h); max_r = std::max(max_r, r); max_d = std::max(max_d, d);
    }
    if ((rc = c.dcol.ensure((size_t)max_d * 4))) return rc;
    if ((rc = c.trace.ensure((size_t)max_t * 16))) return rc;
    if ((rc = c.hcol.ensure((size_t)max_h * 4))) return rc;
    if ((rc = c.rowbuf.ensure((size_t)max_r * 8))) return rc;
    if ((rc = c.plans.ensure((size_t)n_pairs * sizeof(PairPlan)))) return rc;
    if ((rc = c.nops.ensure((size_t)n_pairs * 8))) return rc;
    if ((rc = c.misc.ensure(64))) return rc;
    int *d_err = reinterpret_cast<int *>(c.misc.p);
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, stream));
    HIPCHK(hipMemcpyAsync(c.plans.p, plans.data(), (size_t)n_pairs * sizeof(PairPlan), hipMemcpyHostToDevice, stream));

    // ---- launches ----
    double fill_ms = 0, tb_ms = 0;
    int64_t trace_bytes = 0;
    HIPCHK(hipEventRecord(c.ev[0], stream));
    const size_t nchunks = chunk_begin.size() - 1;
    for (size_t ch = 0; ch < nchunks; ch++) {
        const int64_t b = chunk_begin[ch], e = chunk_begin[ch + 1];
        const int np = (int)(e - b);
        if (np <= 0) continue;
        const PairPlan *dpl = reinterpret_cast<const PairPlan *>(c.plans.p) + b;
        uint4 *dtrace = reinterpret_cast<uint4 *>(c.trace.p);
        int *dh = reinterpret_cast<int *>(c.hcol.p);
        int2 *drb = reinterpret_cast<int2 *>(c.rowbuf.p);
        unsigned *ddc = reinterpret_cast<unsigned *>(c.dcol.p);
        int64_t *dn = reinterpret_cast<int64_t *>(c.nops.p) + b;
        const dim3 gridF((unsigned)((np + 3) / 4)), blockF(64);
        const dim3 gridT((unsigned)((np + 63) / 64)), blockT(64);
        HIPCHK(hipEventRecord(c.ev[1], stream));
        bool multi = false;
        for (int64_t q2 = b; q2 < e; q2++) if (plans[(size_t)q2].strips > 1) { multi = true; break; }
        if (affine) {
#define GNX_LAUNCH_AFF(L_, M_, P_, H_) hipLaunchKernelGGL((fill_affine_kernel<L_, M_, P_, H_>), gridF, blockF, 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, dtrace, dh, drb, ddc, (const int2 *)nullptr, d_err)
#define GNX_LAUNCH_AFF2(L_, M_, P_) do { if (hform) GNX_LAUNCH_AFF(L_, M_, P_, true); else GNX_LAUNCH_AFF(L_, M_, P_, false); } while (0)
#define GNX_LAUNCH_SC(M_, H_) hipLaunchKernelGGL((fill_affine_kernel<false, M_, false, H_, false, true>), gridF, blockF, 0, stream, dpl, np, d_a, d_as, d_b, d_bs, kp, dtrace, dh, drb, ddc, (const int2 *)nullptr, d_err, d_smat)
            const int sel = d_smat ? 8 : ((local ? 4 : 0) | (multi ? 2 : 0) | (p16 ? 1 : 0));
            switch (sel) {
            case 8:
                if (multi) { if (hform) GNX_LAUNCH_SC(true, true); else GNX_LAUNCH_SC(true, false); }
                else { if (hform) GNX_LAUNCH_SC(false, true); else GNX_LAUNCH_SC(false, false); }
                break;
            case 0: GNX_LAUNCH_AFF2(false, false, false); break;
            case 1: GNX_LAUNCH_AFF2(false, false, true); break;
            case 2: GNX_LAUNCH_AFF2(false, true, false); break;
            case 3: GNX_LAUNCH_AFF2(false, true, true); break;
            case 4: GNX_LAUNCH_AFF2(true, false, false); break;
            case 5: GNX_LAUNCH_AFF2(true, false, true); break;
            case 6: GNX_LAUNCH_AFF2(true, true, false); break;
            default: GNX_LAUNCH_AFF2(true, true, true); break;
            }
#undef GNX_LAUNCH_SC
#undef GNX_LAUNCH_AFF2
#undef GNX_LAUNCH_AFF
        } else {
#define GNX_LAUNCH_CONST(M_, G_) hipLaunchKernelGGL((fill_const_kernel<M_, G_>), gridF, blockF, 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, dtrace, dh, drb, ddc, d_err)
            if (gsw == 1) { if (multi) GNX_LAUNCH_CONST(true, 1); else GNX_LAUNCH_CONST(false, 1); }
            else if (gsw == 2) { if (multi) GNX_LAUNCH_CONST(true, 2); else GNX_LAUNCH_CONST(false, 2); }
            else { if (multi) GNX_LAUNCH_CONST(true, 0); else GNX_LAUNCH_CONST(false, 0); }
#undef GNX_LAUNCH_CONST
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c.ev[2], stream));
#define GNX_GSW_TB(R_, W_) hipLaunchKernelGGL((gsw_traceback_kernel<R_, W_>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, d_a, d_as + b, d_b, d_bs + b, kp, d_score + b, d_endpos + b, dn, d_ops_off + b, d_ops, ops_capacity, d_err)
        if (gsw == 1) GNX_GSW_TB(false, false);
        else if (gsw == 2) GNX_GSW_TB(true, false);
        else if (affine) hipLaunchKernelGGL((traceback_kernel<true, false>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, (const int64_t *)nullptr, (gnx_cigar *)nullptr, (int64_t)0, d_err);
        else hipLaunchKernelGGL((traceback_kernel<false, false>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, (const int64_t *)nullptr, (gnx_cigar *)nullptr, (int64_t)0, d_err);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, dn, np, d_ops_off + b, d_carry);
        if (gsw == 1) GNX_GSW_TB(false, true);
        else if (gsw == 2) GNX_GSW_TB(true, true);
        else if (affine) hipLaunchKernelGGL((traceback_kernel<true, true>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, d_ops_off + b, d_ops, ops_capacity, d_err);
        else hipLaunchKernelGGL((traceback_kernel<false, true>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, d_ops_off + b, d_ops, ops_capacity, d_err);
#undef GNX_GSW_TB
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c.ev[3], stream));
        if (nchunks > 1 || true) {
            HIPCHK(hipEventSynchronize(c.ev[3]));
            float f1 = 0, f2 = 0;
            HIPCHK(hipEventElapsedTime(&f1, c.ev[1], c.ev[2]));
            HIPCHK(hipEventElapsedTime(&f2, c.ev[2], c.ev[3]));
            fill_ms += f1; tb_ms += f2;
        }
        for (int64_t p = b; p < e; p++) trace_bytes += (int64_t)plans[(size_t)p].strips * plans[(size_t)p].words * Q * G * 16;
    }
    HIPCHK(hipEventRecord(c.ev[2], stream));
    int h_misc[16];
    HIPCHK(hipMemcpyAsync(h_misc, c.misc.p, 64, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float tot = 0;
    HIPCHK(hipEventElapsedTime(&tot, c.ev[0], c.ev[2]));
    c.timing.fill_ms = fill_ms; c.timing.traceback_ms = tb_ms; c.timing.total_ms = tot;
    c.timing.cells = cells; c.timing.n_launches = (int64_t)nchunks; c.timing.trace_bytes = trace_bytes;
    c.timing.dominant_ms = fill_ms; c.timing.dominant_launches = (int64_t)nchunks; c.timing.fast_path = 0;
    int64_t total;
    memcpy(&total, reinterpret_cast<char *>(h_misc) + 16, 8);
    if (out_total) *out_total = total;
    const int ef = h_misc[0];
    if (ef & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (ef & 2) { set_err("unexpected traceback%s", ""); return GNX_ETRACE; }
    if (ef & 4) { set_err("CIGAR buffer too small: need %s%lld elements", "", (long long)total); return GNX_ECAPACITY; }
    return GNX_OK;
}

// N1 host flow: bases (pairwise sequences or alignment blocks) -> score matrices on the device -> SCORED fill + the
// ordinary highMem traceback -> run lengths times chunk size.  `sp` describes the pairs (offsets into `bases`).
int run_host_scored(const gnx_params *prm, int64_t chunk, bool groups, int64_t n_pairs, std::vector<ScorePair> &sp,
                    const uint8_t *bases, int64_t bases_len, int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    Ctx &c = g_ctx;
    if (!prm || !out_score || !out_ops || !out_ops_off || n_pairs < 0 || chunk < 1 || chunk > (1 << 20)) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    if (prm->mode != GNX_AFFINE_GAP_HIGHMEM) { set_err("the chunk / multiple-alignment variants have AffineGap_highMem semantics (mode %s%lld)", "", (long long)GNX_AFFINE_GAP_HIGHMEM); return GNX_EINVAL; }
    KParams kp0; TbParams tp0; bool aff, loc, low;
    int rc = check_params(prm, kp0, tp0, aff, loc, low);
    if (rc) return rc;
    gnx_params prm2 = *prm; // what the DP sees: gapExtend*chunkSize, |cell score| <= chunkSize*max|score|
    prm2.gap_extend = prm->gap_extend * chunk;
    for (int x = 0; x < 25; x++) prm2.scores[x] = prm->scores[x] * chunk;
    hipStream_t st = c.own_stream;
    std::vector<int64_t> hn((size_t)n_pairs), hm((size_t)n_pairs), hso((size_t)n_pairs);
    int64_t stot = 0, worst = 0, maxcells = 1;
    for (int64_t p = 0; p < n_pairs; p++) {
        ScorePair &q = sp[(size_t)p];
        const int64_t strips = std::max<int64_t>((q.nc + H - 1) / H, 1);
        q.s_pitch = strips * H; q.s_off = stot;
        stot += (int64_t)q.mc * q.s_pitch;
        hn[(size_t)p] = q.nc; hm[(size_t)p] = q.mc; hso[(size_t)p] = q.s_off;
        worst += q.nc + q.mc + 1;
        maxcells = std::max<int64_t>(maxcells, (int64_t)q.nc * q.mc);
    }
    if ((rc = c.in_a.ensure((size_t)bases_len + 16))) return rc;
    if ((rc = c.sc_pairs.ensure((size_t)std::max<int64_t>(n_pairs, 1) * sizeof(ScorePair)))) return rc;
    if ((rc = c.sc_mat.ensure((size_t)std::max<int64_t>(stot, 1) * 4))) return rc;
    if ((rc = c.sc_err.ensure(16))) return rc;
    const size_t np = (size_t)std::max<int64_t>(n_pairs, 1);
    if ((rc = c.out_score.ensure(np * 8))) return rc;
    if ((rc = c.out_off.ensure((np + 1) * 8))) return rc;
    if (bases_len) HIPCHK(hipMemcpyAsync(c.in_a.p, bases, (size_t)bases_len, hipMemcpyHostToDevice, st));
    if (n_pairs) HIPCHK(hipMemcpyAsync(c.sc_pairs.p, sp.data(), (size_t)n_pairs * sizeof(ScorePair), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(c.sc_err.p, 0, 16, st));
    for (int64_t b = 0; b < n_pairs; b += 32768) {
        const unsigned ny = (unsigned)std::min<int64_t>(32768, n_pairs - b);
        const unsigned nx = (unsigned)std::min<int64_t>((maxcells + 255) / 256, 4096);
        hipLaunchKernelGGL(score_matrix_kernel, dim3(nx, ny), dim3(256), 0, st, reinterpret_cast<const ScorePair *>(c.sc_pairs.p) + b,
                           reinterpret_cast<const uint8_t *>(c.in_a.p), kp0, (int)chunk, groups ? 1 : 0, reinterpret_cast<int *>(c.sc_mat.p), reinterpret_cast<int *>(c.sc_err.p));
    }
    HIPCHK(hipGetLastError());
    int sflag[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(sflag, c.sc_err.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (sflag[0] & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (sflag[0] & 16) { set_err("scoreColumnMatch over gap-only columns: the reference panics (integer divide by zero)%s", ""); return GNX_EINVAL; }
    int64_t cap = std::max<int64_t>(std::min<int64_t>(worst, std::max<int64_t>((int64_t)1 << 20, 64 * n_pairs)), 1);
    int64_t total = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = c.out_ops.ensure((size_t)cap * sizeof(gnx_cigar)))) return rc;
        rc = run_device(&prm2, n_pairs, nullptr, nullptr, nullptr, nullptr, hn.data(), hm.data(), (int64_t *)c.out_score.p, (gnx_cigar *)c.out_ops.p, cap,
                        (int64_t *)c.out_off.p, &total, st, reinterpret_cast<const int *>(c.sc_mat.p), hso.data());
        if (rc != GNX_ECAPACITY) break;
        cap = total;
    }
    if (rc) return rc;
    if (chunk > 1 && total > 0) hipLaunchKernelGGL(scale_runs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (gnx_cigar *)c.out_ops.p, total, chunk);
    gnx_cigar *ops = (gnx_cigar *)malloc((size_t)std::max<int64_t>(total, 1) * sizeof(gnx_cigar));
    int64_t *off = (int64_t *)malloc((size_t)(n_pairs + 1) * 8);
    if (!ops || !off) { free(ops); free(off); set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
    if (n_pairs) HIPCHK(hipMemcpyAsync(out_score, c.out_score.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(off, c.out_off.p, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, st));
    if (total) HIPCHK(hipMemcpyAsync(ops, c.out_ops.p, (size_t)total * sizeof(gnx_cigar), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *out_ops = ops; *out_ops_off = off;
    return GNX_OK;
}

int run_host_windows(const gnx_params *prm, int64_t n_pairs,
                     const uint8_t *a_buf, int64_t a_len, const int64_t *a_start, const int64_t *a_lens,
                     const uint8_t *b_buf, int64_t b_len, const int64_t *b_start, const int64_t *b_lens,
                     int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off,
                     int gsw = 0, int64_t *out_end_i = nullptr, int64_t *out_end_j = nullptr) {
    Ctx &c = g_ctx;
    if (!prm || n_pairs < 0 || !out_score || !out_ops || !out_ops_off || a_len < 0 || b_len < 0) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    if (gsw && (!out_end_i || !out_end_j)) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    if (n_pairs > 0 && (!a_start || !a_lens || !b_start || !b_lens)) { set_err("null window table%s", ""); return GNX_EINVAL; }
    for (int64_t p = 0; p < n_pairs; p++) {
        if (a_start[p] < 0 || a_lens[p] < 0 || a_start[p] + a_lens[p] > a_len || b_start[p] < 0 || b_lens[p] < 0 || b_start[p] + b_lens[p] > b_len) {
            set_err("window out of bounds at pair %s%lld", "", (long long)p); return GNX_EINVAL;
        }
    }
    int rc;
    hipStream_t st = c.own_stream;
    const size_t np = (size_t)std::max<int64_t>(n_pairs, 1);
    if ((rc = c.in_a.ensure((size_t)a_len + 16))) return rc;
    if ((rc = c.in_b.ensure((size_t)b_len + 16))) return rc;
    if ((rc = c.in_as.ensure(np * 8))) return rc;
    if ((rc = c.in_bs.ensure(np * 8))) return rc;
    if ((rc = c.out_score.ensure(np * 8))) return rc;
    if ((rc = c.out_off.ensure((np + 1) * 8))) return rc;
    if (gsw && (rc = c.out_end.ensure(np * 8))) return rc;
    if (a_len) HIPCHK(hipMemcpyAsync(c.in_a.p, a_buf, (size_t)a_len, hipMemcpyHostToDevice, st));
    if (b_len) HIPCHK(hipMemcpyAsync(c.in_b.p, b_buf, (size_t)b_len, hipMemcpyHostToDevice, st));
    if (n_pairs) {
        HIPCHK(hipMemcpyAsync(c.in_as.p, a_start, (size_t)n_pairs * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(c.in_bs.p, b_start, (size_t)n_pairs * 8, hipMemcpyHostToDevice, st));
    }
    // CIGAR capacity: exact worst case for small batches, a guess (+ one exact retry) for large ones
    int64_t worst = 0;
    for (int64_t p = 0; p < n_pairs; p++) worst += a_lens[p] + b_lens[p] + 1;
    int64_t cap = std::min<int64_t>(worst, std::max<int64_t>((int64_t)1 << 20, 64 * n_pairs));
    cap = std::max<int64_t>(cap, 1);
    int64_t total = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = c.out_ops.ensure((size_t)cap * sizeof(gnx_cigar)))) return rc;
        rc = run_device(prm, n_pairs, (const uint8_t *)c.in_a.p, (const int64_t *)c.in_as.p, (const uint8_t *)c.in_b.p, (const int64_t *)c.in_bs.p,
                        a_lens, b_lens, (int64_t *)c.out_score.p, (gnx_cigar *)c.out_ops.p, cap, (int64_t *)c.out_off.p, &total, st,
                        nullptr, nullptr, gsw, gsw ? (int2 *)c.out_end.p : nullptr);
        if (rc != GNX_ECAPACITY) break;
        cap = total;
    }
    if (rc) return rc;
    gnx_cigar *ops = (gnx_cigar *)malloc((size_t)std::max<int64_t>(total, 1) * sizeof(gnx_cigar));
    int64_t *off = (int64_t *)malloc((size_t)(n_pairs + 1) * 8);
    if (!ops || !off) { free(ops); free(off); set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
    if (n_pairs) HIPCHK(hipMemcpyAsync(out_score, c.out_score.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(off, c.out_off.p, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, st));
    if (total) HIPCHK(hipMemcpyAsync(ops, c.out_ops.p, (size_t)total * sizeof(gnx_cigar), hipMemcpyDeviceToHost, st));
    std::vector<int2> ends;
    if (gsw && n_pairs) { ends.resize((size_t)n_pairs); HIPCHK(hipMemcpyAsync(ends.data(), c.out_end.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st)); }
    HIPCHK(hipStreamSynchronize(st));
    for (int64_t p = 0; gsw && p < n_pairs; p++) { out_end_i[p] = ends[(size_t)p].x; out_end_j[p] = ends[(size_t)p].y; }
    *out_ops = ops; *out_ops_off = off;
    return GNX_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
extern "C" {

int gnx_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int gnx_init(int device, int64_t workspace_bytes) {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    if (g_ctx.inited) {
        if (device == g_ctx.device) { if (workspace_bytes > 0) g_ctx.ws_limit = workspace_bytes; return GNX_OK; }
        set_err("already bound to another device%s", ""); return GNX_EINVAL;
    }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { set_err("no HIP device available (this library has no CPU fallback)%s", ""); return GNX_EDEVICE; }
    if (device < 0 || device >= cnt) { set_err("device index out of range%s", ""); return GNX_EINVAL; }
    HIPCHK(hipSetDevice(device));
    g_ctx.device = device;
    HIPCHK(hipStreamCreate(&g_ctx.own_stream));
    for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&g_ctx.ev[i]));
    if (workspace_bytes > 0) g_ctx.ws_limit = workspace_bytes;
    else {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        g_ctx.ws_limit = (int64_t)std::min<size_t>(fr / 2, (size_t)64 << 30);
    }
    g_ctx.inited = true;
    return GNX_OK;
}

void gnx_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    if (!g_ctx.inited) return;
    (void)hipSetDevice(g_ctx.device);
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = {&g_ctx.fp_tail, &g_ctx.fp_thcol, &g_ctx.fp_ttrace, &g_ctx.fp_rowi, &g_ctx.fp_ckpt, &g_ctx.fp_states, &g_ctx.fp_stage, &g_ctx.fp_wplans[0], &g_ctx.fp_wplans[1], &g_ctx.fp_active[0], &g_ctx.fp_active[1],
                      &g_ctx.trace, &g_ctx.hcol, &g_ctx.rowbuf, &g_ctx.dcol, &g_ctx.plans, &g_ctx.nops, &g_ctx.misc, &g_ctx.in_a, &g_ctx.in_b,
                      &g_ctx.in_as, &g_ctx.in_al, &g_ctx.in_bs, &g_ctx.in_bl, &g_ctx.out_score, &g_ctx.out_off, &g_ctx.out_ops, &g_ctx.out_end,
                      &g_ctx.sc_pairs, &g_ctx.sc_mat, &g_ctx.sc_err};
    for (DevBuf *b : bufs) b->release();
    for (int i = 0; i < 4; i++) if (g_ctx.ev[i]) { (void)hipEventDestroy(g_ctx.ev[i]); g_ctx.ev[i] = nullptr; }
    if (g_ctx.own_stream) { (void)hipStreamDestroy(g_ctx.own_stream); g_ctx.own_stream = nullptr; }
    for (int i = 4; i < 8; i++) if (g_ctx.ev[i]) { (void)hipEventDestroy(g_ctx.ev[i]); g_ctx.ev[i] = nullptr; }
    g_ctx.inited = false;
    g_ctx.ws_limit = 0;
}

const char *gnx_last_error(void) { return g_err; }

void gnx_free(void *p) { free(p); }

int gnx_align_batch_windows(const gnx_params *p, int64_t n_pairs,
                            const uint8_t *alpha_buf, int64_t alpha_buf_len, const int64_t *alpha_start, const int64_t *alpha_len,
                            const uint8_t *beta_buf, int64_t beta_buf_len, const int64_t *beta_start, const int64_t *beta_len,
                            int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    g_err[0] = 0;
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(g_ctx.device));
    return run_host_windows(p, n_pairs, alpha_buf, alpha_buf_len, alpha_start, alpha_len, beta_buf, beta_buf_len, beta_start, beta_len,
                            out_score, out_ops, out_ops_off);
}

int gnx_align_batch(const gnx_params *p, int64_t n_pairs, const uint8_t *alpha_cat, const int64_t *alpha_off,
                    const uint8_t *beta_cat, const int64_t *beta_off, int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    if (n_pairs < 0 || !alpha_off || !beta_off) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    std::vector<int64_t> al((size_t)n_pairs), bl((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) { al[(size_t)q] = alpha_off[q + 1] - alpha_off[q]; bl[(size_t)q] = beta_off[q + 1] - beta_off[q]; }
    return gnx_align_batch_windows(p, n_pairs, alpha_cat, alpha_off[n_pairs], alpha_off, al.data(), beta_cat, beta_off[n_pairs], beta_off, bl.data(),
                                   out_score, out_ops, out_ops_off);
}

int gnx_gsw_extend_batch(int side, const int64_t *scores, int64_t gap_pen, int64_t n_pairs,
                         const uint8_t *alpha_cat, const int64_t *alpha_off, const uint8_t *beta_cat, const int64_t *beta_off,
                         int64_t *out_score, int64_t *out_end_i, int64_t *out_end_j, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    g_err[0] = 0;
    if ((side != GNX_GSW_LEFT && side != GNX_GSW_RIGHT) || !scores || n_pairs < 0 || !alpha_off || !beta_off) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(g_ctx.device));
    gnx_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.mode = GNX_CONST_GAP_HIGHMEM;
    for (int x = 0; x < 25; x++) prm.scores[x] = scores[x];
    prm.gap_open = gap_pen; prm.checkersize_i = 10000; prm.checkersize_j = 10000;
    std::vector<int64_t> al((size_t)n_pairs), bl((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) { al[(size_t)q] = alpha_off[q + 1] - alpha_off[q]; bl[(size_t)q] = beta_off[q + 1] - beta_off[q]; }
    return run_host_windows(&prm, n_pairs, alpha_cat, alpha_off[n_pairs], alpha_off, al.data(), beta_cat, beta_off[n_pairs], beta_off, bl.data(),
                            out_score, out_ops, out_ops_off, side == GNX_GSW_LEFT ? 1 : 2, out_end_i, out_end_j);
}

int gnx_align_pair(const gnx_params *p, const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m,
                   int64_t *out_score, gnx_cigar **out_ops, int64_t *out_n_ops) {
    if (!out_n_ops) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    const int64_t zero = 0;
    int64_t *off = nullptr;
    int rc = gnx_align_batch_windows(p, 1, alpha, n, &zero, &n, beta, m, &zero, &m, out_score, out_ops, &off);
    if (rc) return rc;
    *out_n_ops = off[1];
    free(off);
    return GNX_OK;
}

int gnx_align_batch_device(const gnx_params *p, int64_t n_pairs,
                           const uint8_t *d_alpha_buf, const int64_t *d_alpha_start, const int64_t *d_alpha_len,
                           const uint8_t *d_beta_buf, const int64_t *d_beta_start, const int64_t *d_beta_len,
                           const int64_t *h_alpha_len, const int64_t *h_beta_len,
                           int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off,
                           int64_t *out_total_ops, void *stream) {
    (void)d_alpha_len; (void)d_beta_len; // lengths are taken from the host copies (planning needs them anyway)
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    g_err[0] = 0;
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(g_ctx.device));
    if (!p || n_pairs < 0 || !d_score || !d_ops_off || ops_capacity < 0 || (n_pairs > 0 && (!h_alpha_len || !h_beta_len || !d_alpha_start || !d_beta_start))) {
        set_err("bad argument%s", ""); return GNX_EINVAL;
    }
    return run_device(p, n_pairs, d_alpha_buf, d_alpha_start, d_beta_buf, d_beta_start, h_alpha_len, h_beta_len,
                      d_score, d_ops, ops_capacity, d_ops_off, out_total_ops, (hipStream_t)stream);
}

int gnx_affine_gap_chunk_batch(const gnx_params *p, int64_t chunk_size, int64_t n_pairs,
                               const uint8_t *alpha_cat, const int64_t *alpha_off, const uint8_t *beta_cat, const int64_t *beta_off,
                               int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    g_err[0] = 0;
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(g_ctx.device));
    if (n_pairs < 0 || chunk_size < 1 || (n_pairs > 0 && (!alpha_off || !beta_off))) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    const int64_t la = n_pairs ? alpha_off[n_pairs] : 0, lb = n_pairs ? beta_off[n_pairs] : 0;
    std::vector<uint8_t> bases((size_t)(la + lb + 1));
    if (la) memcpy(bases.data(), alpha_cat, (size_t)la);
    if (lb) memcpy(bases.data() + la, beta_cat, (size_t)lb);
    std::vector<ScorePair> sp((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) {
        const int64_t n = alpha_off[q + 1] - alpha_off[q], m = beta_off[q + 1] - beta_off[q];
        if (n < 0 || m < 0 || n > 0x3fffffff || m > 0x3fffffff) { set_err("bad sequence length at pair %s%lld", "", (long long)q); return GNX_EINVAL; }
        if (n % chunk_size != 0 || m % chunk_size != 0) { // log.Fatalf in the reference (affineGap_highMem.go:229-234)
            set_err("pair %s%lld: sequence length is not a multiple of the chunk size", "", (long long)q); return GNX_EINVAL;
        }
        ScorePair &s = sp[(size_t)q];
        s.a_off = alpha_off[q]; s.b_off = la + beta_off[q]; s.a_nseq = 1; s.b_nseq = 1; s.a_len = (int32_t)n; s.b_len = (int32_t)m;
        s.nc = (int32_t)(n / chunk_size); s.mc = (int32_t)(m / chunk_size); s.s_off = 0; s.s_pitch = 0;
    }
    return run_host_scored(p, chunk_size, false, n_pairs, sp, bases.data(), la + lb, out_score, out_ops, out_ops_off);
}

int gnx_multiple_affine_gap_batch(const gnx_params *p, int64_t chunk_size, int64_t n_groups, const uint8_t *group_bases,
                                  const int64_t *group_off, const int32_t *group_nseq, const int64_t *group_len,
                                  int64_t n_pairs, const int32_t *pair_a, const int32_t *pair_b,
                                  int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    g_err[0] = 0;
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(g_ctx.device));
    if (n_pairs < 0 || n_groups < 0 || chunk_size < 1 || (n_groups > 0 && (!group_off || !group_nseq || !group_len)) || (n_pairs > 0 && (!pair_a || !pair_b))) {
        set_err("bad argument%s", ""); return GNX_EINVAL;
    }
    for (int64_t g = 0; g < n_groups; g++) {
        if (group_nseq[g] < 1 || group_len[g] < 0 || group_len[g] > 0x3fffffff || group_off[g + 1] - group_off[g] != (int64_t)group_nseq[g] * group_len[g]) {
            set_err("group %s%lld: bases do not match nseq x len", "", (long long)g); return GNX_EINVAL;
        }
    }
    std::vector<ScorePair> sp((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) {
        const int32_t a = pair_a[q], b = pair_b[q];
        if (a < 0 || b < 0 || a >= n_groups || b >= n_groups) { set_err("pair %s%lld: group index out of range", "", (long long)q); return GNX_EINVAL; }
        if (group_len[a] % chunk_size != 0 || group_len[b] % chunk_size != 0) { // log.Fatalf (affineGap_highMem.go:310-315)
            set_err("pair %s%lld: alignment length is not a multiple of the chunk size", "", (long long)q); return GNX_EINVAL;
        }
        ScorePair &s = sp[(size_t)q];
        s.a_off = group_off[a]; s.b_off = group_off[b]; s.a_nseq = group_nseq[a]; s.b_nseq = group_nseq[b];
        s.a_len = (int32_t)group_len[a]; s.b_len = (int32_t)group_len[b];
        s.nc = (int32_t)(group_len[a] / chunk_size); s.mc = (int32_t)(group_len[b] / chunk_size); s.s_off = 0; s.s_pitch = 0;
    }
    return run_host_scored(p, chunk_size, true, n_pairs, sp, group_bases, n_groups ? group_off[n_groups] : 0, out_score, out_ops, out_ops_off);
}

int gnx_get_timing(gnx_timing *out) {
    if (!out) return GNX_EINVAL;
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    *out = g_ctx.timing;
    return GNX_OK;
}

} // extern "C"
