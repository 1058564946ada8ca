// fp_sweep16.hip.h -- fast-path forward sweep on PACKED int16: two pairs per register, 16 pairs per wave64
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.5.
#pragma once
#include "fp_sweep.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// Same sweep as fp_sweep_kernel (8 lanes x RR rows per pair, rebased keys, right-aligned rows, checkpoints every CKW columns, I-planes
// of rows n..n-3) with every VGPR holding the cell of TWO pairs: pair 2g in the low, pair 2g+1 in the high 16 bits.  The two cells
// are independent (different DP matrices), so v_pk_add_i16 / v_pk_max_i16 do both at the price of one: 6 packed instructions per
// 2 plane-less cells instead of 2 x 5 (there is no packed max3), and the per-step overhead (DPP moves, LDS reads of the profile)
// is shared by 16 pairs.  Conditions (host): every pair of the batch has the same m and the same beta window (one chunk: config C2,
// cmd/faChunkAlign) -- then one base per column and ONE interleaved profile read serve both halves --, (smax - 2e) < 700, not XP.
//
// Range.  Rebased values V' = V - e(i+j) grow by up to c_i = max_b(s(a_i,b) - 2e) per row, so they do not fit 16 bits; but
// V'(i,j) <= R_i + o with R_i = c_1 + .. + c_i (proof: at least one gap open off the main diagonal, all else at most c_k), i.e.
// the values of a row live just below R_i.  Every lane therefore keeps its 19 / 20 rows relative to ITS frame F = R at the lane's
// middle row: values <= (RR/2) c_max above it; the transfer to the next lane (the two DPP moves) subtracts F_next - F.  Below,
// arithmetic SATURATES (v_pk_add_i16 clamp): a saturated cell is over-estimated, but in the row frame V'' = V' - R_i -- where every
// transition is <= 0 -- it sits at <= -8192 and so does everything derived from it, while the cells of the optimal path are >=
// V''(n,m).  Pairs with V''(n,m) > LOW16 are therefore exact wherever it matters: h(n,m), the planes and corner tags of rows n..n-3
// along the path, and the checkpoint entries a window re-fill needs for the cells the walk visits (a re-fill recomputes in int32
// from the checkpoint; over-estimated boundary cells lose every comparison against path cells).  The others (> ~35 mismatches or
// ~13 gap opens in 150 bases) are listed in `low_list` and swept again by fp_sweep_kernel, which overwrites their outputs.
// Plane rows (last four slots of a lane) are kept as 4*value + tag like everywhere else (range / 4: the -8192 above); the rows above
// them carry plain values, converted on entry (saturating x4) and exit (>> 2) of the four rows.
// ------------------------------------------------------------------------------------------------------
typedef short s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2v as_s2(int x) { return __builtin_bit_cast(s2v, x); }
__device__ __forceinline__ int as_i(s2v x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int pk_adds(int a, int b) { return as_i(__builtin_elementwise_add_sat(as_s2(a), as_s2(b))); }
__device__ __forceinline__ int pk_max(int a, int b) { return as_i(__builtin_elementwise_max(as_s2(a), as_s2(b))); }
__device__ __forceinline__ int pk_sar2(int a) { return as_i(as_s2(a) >> 2); }
__device__ __forceinline__ int pk2(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int lo16(int x) { return (int)(short)(x & 0xffff); }
__device__ __forceinline__ int hi16(int x) { return x >> 16; }
__device__ __forceinline__ int clamp16(int x) { return min(max(x, -32768), 32767); }

constexpr int S16_LW = 20;                  // dwords per lane per base: slot r of both pairs
constexpr int S16_BST = G8 * S16_LW;        // 160 dwords per base plane (multiple of 32)
constexpr int S16_PST = 5 * S16_BST + 16;   // dwords per group of two pairs
constexpr int LOW16 = -7500;                // V''(n,m) at or below this: the pair is swept again in int32

template <int RR>
__global__ __launch_bounds__(64) void fp_sweep16_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                        const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                        const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                        KParams kp, int *__restrict__ hcol, int2 *__restrict__ ckpt,
                                                        unsigned *__restrict__ rowi, unsigned *__restrict__ tail, int *__restrict__ err,
                                                        int *__restrict__ low_count, int *__restrict__ low_list) {
    // low_count[0] = number of entries of low_list; low_count[1] != 0: some pair has another beta window than pair 0 (the host then
    // sweeps the whole batch with fp_sweep_kernel instead)
    static_assert(RR <= S16_LW && RR > FP_PLANES + 1, "rows per lane");
    constexpr int NU = RR - FP_PLANES; // slots [0, NU) carry plain values, [NU, RR) 4*value + tag
    __shared__ int lds[32 + 8 * S16_PST];
    const int lane = threadIdx.x;
    const int g = lane >> 3;
    const int lp = (lane & 8) ? 15 - (lane & 15) : (lane & 7);
    const int E4 = kp.e4, o1 = kp.o4 >> 2; // (multiples of 4: exact)
    if (lane < 25) lds[lane] = (kp.sc4[lane] - 2 * E4) >> 2; // s - 2e
    int *prof = &lds[32 + g * S16_PST];
    const char *prof_lane = reinterpret_cast<const char *>(prof + lp * S16_LW);

    const int pbase = blockIdx.x * 16;
    const int pL = pbase + 2 * g, pH = pL + 1;
    const bool validL = pL < n_pairs, validH = pH < n_pairs;
    const PairPlan plL = plans[validL ? pL : n_pairs - 1];
    const PairPlan plH = validH ? plans[pH] : plL;
    const uint8_t *apL = a_buf + a_start[plL.src], *apH = a_buf + a_start[plH.src];
    const int64_t b0 = b_start[plans[0].src];
    if (__any(b_start[plL.src] != b0 || b_start[plH.src] != b0)) { if (lane == 0) atomicOr(low_count + 1, 1); return; }
    const uint8_t *bp = b_buf + b0; // one beta window for the whole batch
    const int m_eff = plL.m;                        // ... and one m
    const int PL = G8 * RR - plL.n, PH = G8 * RR - plH.n; // padding slots above row 1
    const int q0 = lp * RR;
    int bad = 0;

    // ---- per-slot bases, row maxima, lane frames ----
    int a5L[S16_LW], a5H[S16_LW];
#pragma unroll
    for (int r = 0; r < S16_LW; r++) {
        int aL = 5, aH = 5; // padding
        const int q = q0 + r;
        if (r < RR && q >= PL) { aL = apL[q - PL]; if (aL >= 5) { bad = 1; aL = 4; } }
        if (r < RR && q >= PH) { aH = apH[q - PH]; if (aH >= 5) { bad = 1; aH = 4; } }
        a5L[r] = aL * 5; a5H[r] = aH * 5;
    }
    __syncthreads();
    if (lane >= 32 && lane < 37) { // row maxima c(a) = max_b (s(a,b) - 2e); c(padding) = 0
        const int a = lane - 32;
        int c = lds[a * 5];
        for (int b = 1; b < 5; b++) c = max(c, lds[a * 5 + b]);
        lds[25 + a] = c;
    }
    if (lane == 37) lds[30] = 0;
    __syncthreads();
    int CL = 0, CH = 0, CmidL = 0, CmidH = 0; // sums of c over this lane's slots / up to its middle slot
#pragma unroll
    for (int r = 0; r < RR; r++) {
        const int cL = lds[25 + a5L[r] / 5], cH = lds[25 + a5H[r] / 5];
        CL += cL; CH += cH;
        if (r <= RR / 2) { CmidL += cL; CmidH += cH; }
    }
    int *scr = &lds[32 + g * S16_PST]; // scratch in the profile area: [lp][4]
    scr[lp * 4 + 0] = CL; scr[lp * 4 + 1] = CH; scr[lp * 4 + 2] = CmidL; scr[lp * 4 + 3] = CmidH;
    __syncthreads();
    int baseL = 0, baseH = 0;
    for (int x = 0; x < lp; x++) { baseL += scr[x * 4 + 0]; baseH += scr[x * 4 + 1]; }
    const int FL = baseL + CmidL, FH = baseH + CmidH; // lane frames
    int dFL = 0, dFH = 0;                               // F - F(previous lane)
    if (lp > 0) { dFL = FL - (baseL - scr[(lp - 1) * 4 + 0] + scr[(lp - 1) * 4 + 2]); dFH = FH - (baseH - scr[(lp - 1) * 4 + 1] + scr[(lp - 1) * 4 + 3]); }
    const int negDF = pk2(clamp16(-dFL), clamp16(-dFH));
    __syncthreads();
    // ---- profile: prof[b][lp][r] = {s(aH,b) - 2e, s(aL,b) - 2e}, x4 in the plane slots, padding -32768 ----
#pragma unroll
    for (int b = 0; b < 5; b++) {
#pragma unroll
        for (int r = 0; r < S16_LW; r++) {
            int vL = -32768, vH = -32768;
            if (r < RR) {
                if (a5L[r] < 25) vL = lds[a5L[r] + b] * (r >= NU ? 4 : 1);
                if (a5H[r] < 25) vH = lds[a5H[r] + b] * (r >= NU ? 4 : 1);
            }
            prof[b * S16_BST + lp * S16_LW + r] = pk2(vL, vH);
        }
    }
    __syncthreads();

    // ---- column 0 (see fp_sweep_kernel), in the lane frame ----
    int rt[RR], hold[RR];
    auto col0 = [&](int q, int P, int F, bool plane, int &h, int &rtv) {
        // real row i: h'(i,0) = D'(i,0) = o (tag D = 1), I'(i,1) = 2o (from D); padding: h' = I' = o (tag 2); slot above row 1: h(0,0) = 0 (tag 3)
        const int hv = (q >= P) ? o1 : (q == P - 1 ? 0 : o1), rv = (q >= P) ? 2 * o1 : o1;
        const int ht = (q >= P) ? 1 : (q == P - 1 ? 3 : 2), rtag = (q >= P) ? 1 : 2;
        h = plane ? (clamp16(4 * (hv - F)) | ht) : clamp16(hv - F);
        rtv = plane ? (clamp16(4 * (rv - F)) | rtag) : clamp16(rv - F);
    };
#pragma unroll
    for (int r = 0; r < RR; r++) {
        int hL, hH, rL, rH;
        col0(q0 + r, PL, FL, r >= NU, hL, rL);
        col0(q0 + r, PH, FH, r >= NU, hH, rH);
        hold[r] = pk2(hL, hH); rt[r] = pk2(rL, rH);
    }
    int diag0;
    {
        int hL, hH, x;
        col0(q0 - 1, PL, FL, false, hL, x); // the slot above this lane's first one (slot -1 of an unpadded pair is (0,0): q == P - 1)
        col0(q0 - 1, PH, FH, false, hH, x);
        diag0 = pk2(hL, hH);
    }
    // row-0 boundary of the first lane, constant in j >= 1: h'(0,j) = o, D'(1,j) = 2o (in its frame); kept by the DPP moves as `old`
    int up_h = pk2(clamp16(o1 - FL), clamp16(o1 - FH)), up_dn = pk2(clamp16(2 * o1 - FL), clamp16(2 * o1 - FH));
    int vO, vO4;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(vO), "=v"(vO4) : "s"(pk2(o1, o1)), "s"(pk2(4 * o1, 4 * o1)));
    unsigned accL[FP_PLANES] = {0u, 0u, 0u, 0u}, accH[FP_PLANES] = {0u, 0u, 0u, 0u};
    unsigned tailL = 0, tailH = 0;
    int dn_out = 0, h_out = 0, b_out = 0;
    auto base_of = [&](int c) {
        int b = 0;
        if (c >= 1 && c <= m_eff) { b = bp[c - 1]; if (b >= 5) { bad = 1; b = 4; } }
        return b * (S16_BST * 4);
    };
    int qb = base_of(lp), nb = 0;

    auto step = [&](const int t, auto chk, const bool ckflag) {
        constexpr bool CHECK = decltype(chk)::value;
        up_dn = dpp_prev8(up_dn, dn_out);
        up_h = dpp_prev8(up_h, h_out);
        const int pb = dpp_prev8(qb, b_out);
        qb = dpp_next8(qb);
        const int j = t - lp;
        b_out = pb;
        if (!CHECK || (j >= 1 && j <= m_eff)) {
            const int udn = pk_adds(up_dn, negDF), uh = pk_adds(up_h, negDF); // into this lane's frame
            const int2 *pw = reinterpret_cast<const int2 *>(prof_lane + pb);
            int w[S16_LW];
#pragma unroll
            for (int k = 0; k < S16_LW / 2; k++) { const int2 v = pw[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
            int hd = diag0, dnu = udn;
#pragma unroll
            for (int r = 0; r < NU; r++) { // plain values
                const int M = pk_adds(hd, w[r]);
                const int hnew = pk_max(pk_max(M, rt[r]), dnu);
                const int ho = pk_adds(hnew, vO);
                rt[r] = pk_max(ho, rt[r]);
                const int dnn = pk_max(ho, dnu);
                hd = hold[r];
                hold[r] = hnew;
                dnu = dnn;
            }
            { const int a = pk_adds(hd, hd), b = pk_adds(dnu, dnu); hd = pk_adds(a, a); dnu = pk_adds(b, b); } // x4, saturating
#pragma unroll
            for (int r = NU; r < RR; r++) { // 4*value + tag: the rows whose I-planes are kept when they are rows n..n-3
                accL[RR - 1 - r] = alignbit2((unsigned)rt[r], accL[RR - 1 - r]);
                accH[RR - 1 - r] = alignbit2((unsigned)rt[r] >> 16, accH[RR - 1 - r]);
                const int M3 = pk_adds(hd | 0x00030003, w[r]);
                const int I2 = (rt[r] & (int)0xfffcfffc) | 0x00020002;
                const int D1 = (dnu & (int)0xfffcfffc) | 0x00010001;
                const int hnew = pk_max(pk_max(M3, I2), D1);
                const int ho = pk_adds(hnew, vO4);
                rt[r] = pk_max(ho, I2);
                const int dnn = pk_max(ho, D1);
                hd = hold[r];
                hold[r] = hnew;
                dnu = dnn;
            }
            diag0 = uh;
            dn_out = pk_sar2(dnu);
            h_out = pk_sar2(hold[RR - 1]);
            if (CHECK && j + 3 >= m_eff) { // the last four columns (always in a CHECK half block): corner tags, see fp_sweep_kernel
#pragma unroll
                for (int d = 0; d < FP_PLANES; d++) {
                    tailL |= (unsigned)(hold[RR - 1 - d] & 3) << (8 * (m_eff - j) + 2 * d);
                    tailH |= (unsigned)((hold[RR - 1 - d] >> 16) & 3) << (8 * (m_eff - j) + 2 * d);
                }
            }
#ifndef FP16_NOCK
            if (ckflag) {
            asm volatile("" ::: "memory");
            if ((j & (CKW - 1)) == 0 && j < m_eff) { // column checkpoint {I(i,j+1), h(i,j)+e}: int32, un-rebased, tag bits junk
                int2 *ckL = ckpt + plL.ckpt_off + (int64_t)(j / CKW - 1) * plL.n;
                int2 *ckH = ckpt + plH.ckpt_off + (int64_t)(j / CKW - 1) * plH.n;
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    const int iL = q0 + r - PL + 1, iH = q0 + r - PH + 1;
                    const int sc = (r >= NU) ? 1 : 4;
                    if (iL >= 1 && validL) { const int off = 4 * FL + E4 * (iL + j + 1); ckL[iL - 1] = make_int2(sc * lo16(rt[r]) + off, sc * lo16(hold[r]) + off); }
                    if (iH >= 1 && validH) { const int off = 4 * FH + E4 * (iH + j + 1); ckH[iH - 1] = make_int2(sc * hi16(rt[r]) + off, sc * hi16(hold[r]) + off); }
                }
            }
            }
#endif
        }
    };

    const int Tend = ((m_eff + G8 - 1) / 16 + 1) * 16;
    auto flush = [&](int t0) { // the last lane owns rows n..n-3: after the odd half block, store the plane word of steps t0-8 .. t0+7
        if ((t0 & 8) && lp == G8 - 1) {
            const int w = t0 >> 4;
            const int miss = (t0 + 7) - (m_eff + G8 - 1);
            const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
            for (int d = 0; d < FP_PLANES; d++) {
                accL[d] >>= sh; accH[d] >>= sh;
                if (validL && w < plL.words && plL.n - d >= 1) rowi[plL.rowi_off + (int64_t)d * plL.words + w] = accL[d];
                if (validH && w < plH.words && plH.n - d >= 1) rowi[plH.rowi_off + (int64_t)d * plH.words + w] = accH[d];
            }
        }
    };
    auto edge_half_block = [&](int t0) {
        nb = base_of(t0 + 8 + lp);
#pragma unroll 1
        for (int u = 0; u < 8; u++) step(t0 + u, std::true_type{}, true);
        qb = nb;
        flush(t0);
    };
    int t0 = 0;
    for (; t0 < Tend && !(t0 >= 8 && t0 + 7 <= m_eff); t0 += 8) edge_half_block(t0);
    for (; t0 + 7 <= m_eff; t0 += 8) { // steady state: every lane of the wave is inside its matrix
        nb = base_of(t0 + 8 + lp);
        const bool ckflag = (t0 & (CKW - 1)) == 0;
#ifndef FP16_UNROLL
#define FP16_UNROLL 8
#endif
#pragma unroll FP16_UNROLL
        for (int u = 0; u < 8; u++) step(t0 + u, std::false_type{}, ckflag);
        qb = nb;
        flush(t0);
    }
    for (; t0 < Tend; t0 += 8) edge_half_block(t0);
    if (lp == G8 - 1 && m_eff >= 1) {
        // h(n, m) with its tag; V''(n,m) = h' - R_n decides whether the pair has to be swept again in int32
        const int kL = lo16(hold[RR - 1]), kH = hi16(hold[RR - 1]);
        if (validL) {
            hcol[plL.hcol_off] = kL + 4 * FL + E4 * (plL.n + m_eff); tail[plL.hcol_off] = tailL;
            if ((kL >> 2) - (baseL + CL - FL) <= LOW16) low_list[atomicAdd(low_count, 1)] = pL;
        }
        if (validH) {
            hcol[plH.hcol_off] = kH + 4 * FH + E4 * (plH.n + m_eff); tail[plH.hcol_off] = tailH;
            if ((kH >> 2) - (baseH + CH - FH) <= LOW16) low_list[atomicAdd(low_count, 1)] = pH;
        }
    }
    if (bad) atomicOr(err, 1);
}

// plans of the pairs the int16 sweep flagged, compacted for a launch of fp_sweep_kernel (their outputs land in the same slots)
__global__ __launch_bounds__(256) void fp_low_plans_kernel(const PairPlan *__restrict__ plans, const int *__restrict__ list, int n, PairPlan *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) out[x] = plans[list[x]];
}

} // namespace
