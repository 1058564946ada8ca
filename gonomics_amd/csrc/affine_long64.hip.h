// affine_long64.hip.h -- the snapshot path of affine_long.hip.h with the WHOLE WAVE on one pair: 64 lanes x 10 rows, strips of 640 rows
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.13.
#pragma once
#include "affine_long.hip.h"
#include "lat_fill.hip.h" // wave_shr1

namespace {
// ------------------------------------------------------------------------------------------------------
// al_sweep_kernel / al_walk_kernel map a pair to 16 lanes x 10 rows, four pairs per wave: right for batches, but ONE long pair -- the
// cmd/cigarToBed case, 1 Mb x 1 Mb and beyond -- leaves 48 of 64 lanes idle in every wave of its pipeline (sweep 1.09 s per 1e12 cells).
// Here the wavefront is the whole wave: lane l owns rows 10 l + 1 .. 10 l + 10 of a 640-row strip, the lane-to-lane moves are wave_shr:1
// DPP moves (lat_fill.hip.h), everything else is affine_long's scheme on moving bases (REBASE, const_long.hip.h):
//   al64_sweep_kernel  score only (add, max3, add, max, max per cell; a lane's last row keeps its tags), strips piped through the row buffer
//                      with progress words and claims, the bottom row {dn, h} of every strip (8 B per column and 640 rows), a snapshot of
//                      the wavefront every K steps (24 dwords per lane; K = KParams::ckc: CK64 = 128 for the walk kernels below, up to 512 when the walk
//                      farm of farm64.hip.h re-fills the tiles), bases[strip][block].  A column of the bottom row is written by lane 63 at step c + 63:
//                      its block is (c + 62) / K.
//   al64_walk_kernel   one wave per pair: re-fill the tile (strip, <= CK64 + 4 steps) with the recording recurrence into three 2-bit planes
//                      in LDS (69 KB: one workgroup per CU may declare up to 160 KB), walk inside it (quirks Q1 / Q2, run merging, MegaState for row panels: affine_long's code with 64 lanes).
// Always on moving bases (the pairs that come here are long).  Chosen by the host for launches of few pairs (GNX_W64=0 / 2: never / always).
// ------------------------------------------------------------------------------------------------------
constexpr int G64 = 64;                              // lanes per pair
constexpr int H64 = G64 * R;                         // rows per strip
constexpr int CK64 = CKA;                            // snapshot spacing in wavefront steps (the same as affine_long's: 0.075 B of snapshots per cell)
constexpr int AL64_WORDS = CK64 / 16 + 1;            // direction words per plane row of a tile
constexpr int AL64_DIRG = AL64_WORDS * 3 * R * G64;  // LDS dwords of the tile
constexpr int XB64 = G64 - 2;                        // a bottom-row column c is stored in the 16-step block of step c + 63: block index (c + XB64) / CK64
// Rows per lane of the 64-lane AFFINE sweep and of the walk farm's re-fills (round 6): RW = 6 / 8 / 10 / 16 -- strips of 384 / 512 / 640 / 1 024 rows.  One long pair is
// strips(RW) waves on 1 024 SIMDs, paced by the SIMD that holds most of them: the host picks the RW whose strip count fills whole rounds of the SIMDs (w64_pick_rows,
// gnx_align.hip).  The one-workgroup walks above (GNX_W64_FARM=0) and the constant-gap twins stay at R = 10.
__host__ __device__ constexpr int al64_snapw(int rw) { return (2 * rw + 2 + 3) & ~3; } // dwords per lane and snapshot: rt[RW], hold[RW], diag0, dn_out (RW = 10: AL_SNAPW)
static_assert(al64_snapw(R) == AL_SNAPW, "the walk kernels of this file read the sweep's snapshots at RW = R");
template <int RW>
__device__ __forceinline__ void al64_snap_store(uint4 *dst, const int (&rt)[RW], const int (&hold)[RW], int diag0, int dn_out) {
    constexpr int SW = al64_snapw(RW);
    unsigned v[SW];
#pragma unroll
    for (int r = 0; r < RW; r++) { v[r] = (unsigned)rt[r]; v[RW + r] = (unsigned)hold[r]; }
    v[2 * RW] = (unsigned)diag0; v[2 * RW + 1] = (unsigned)dn_out;
#pragma unroll
    for (int k = 2 * RW + 2; k < SW; k++) v[k] = 0u;
#pragma unroll
    for (int q = 0; q < SW / 4; q++) dst[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
template <int RW>
__device__ __forceinline__ void al64_snap_load(const uint4 *sp, int (&rt)[RW], int (&hold)[RW], int &diag0, int &dn_out) {
    constexpr int SW = al64_snapw(RW);
    unsigned v[SW];
    uint4 x[SW / 4];
#pragma unroll
    for (int q = 0; q < SW / 4; q++) x[q] = sp[q]; // (every load in flight before the first use)
#pragma unroll
    for (int q = 0; q < SW / 4; q++) { v[4 * q] = x[q].x; v[4 * q + 1] = x[q].y; v[4 * q + 2] = x[q].z; v[4 * q + 3] = x[q].w; }
#pragma unroll
    for (int r = 0; r < RW; r++) { rt[r] = (int)v[r]; hold[r] = (int)v[RW + r]; }
    diag0 = (int)v[2 * RW]; dn_out = (int)v[2 * RW + 1];
}

// Hand-over between the piped strips of the 64-lane sweeps WITHOUT a progress word (round 6).  Round 5: a strip stored its bottom row with write-through stores, waited
// for their acknowledgement (s_waitcnt vmcnt(0): a trip to memory, ~2 - 3 us) every 64 steps and advanced a progress word the strip below polled -- a wave alone on its
// SIMD (what the strips of ONE long pair are) sat out that wait in full: its step cost ~100 - 150 ns whatever its instruction count (ConstGap 150 kb x 180 kb: 24.1 ms at 10
// rows per lane, 24.3 ms at 4).  Now every datum says for itself whether it is there (the protocol of lat_fill_kernel): the host fills the row buffer and the bases with
// 0x80 bytes before the launch (W64_SENT / W64_BSENT: below every key, not a base), the producer just stores, the consumer re-loads an entry that still reads as the
// sentinel.  An entry is one aligned 8-byte (ConstGap: 4-byte) store and a base one 8-byte store: never torn; nobody relies on the ORDER of two stores.
constexpr int W64_SENT = (int)0x80808080;
constexpr long long W64_BSENT = (long long)0x8080808080808080ULL;
__device__ __forceinline__ long long w64_base_wait(const long long *p, int *err) {
    long long v = rbase_load(p, true);
    if (v == W64_BSENT) {
        const long long t_begin = wall_clock64();
        while ((v = rbase_load(p, true)) == W64_BSENT) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); v = 0; break; }
        }
    }
    return v;
}

template <typename F, int... Us>
__device__ __forceinline__ void al64_unrolled_block(std::integer_sequence<int, Us...>, F &&f) { (f(std::integral_constant<int, Us>{}), ...); }
// lane 0 <- lane U of its row of 16 (U = 0: a copy): the boundary / base queues of a block stay where they were loaded, the unrolled step U reads its entry
template <int U>
__device__ __forceinline__ int dpp_row_shl(int src) {
    if constexpr (U == 0) return src;
    else return __builtin_amdgcn_update_dpp(0, src, 0x100 + U, 0xf, 0xf, true); // (bound_ctrl: no `old` operand to copy; lane 0's source lane U always exists)
}
// lanes >= 1: src[l - 1] + inc, lane 0: oldv -- one v_add_u32_dpp (the profile offset of a column travels down the wave and gains a lane's stride per hop, so
// that it IS the LDS address).  s_nop: a DPP read of a VGPR needs two wait states after the VALU write of it, and the compiler does not look inside this statement.
__device__ __forceinline__ int wave_shr1_add(int oldv, int src, int inc) {
    int r = oldv;
    asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(src), "v"(inc));
    return r;
}

template <int RW, bool P16>
__global__ __launch_bounds__(64) void al64_sweep_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                        const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                        const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                        KParams kp, int2 *__restrict__ rowbuf, int *__restrict__ snap, int64_t *__restrict__ hfin,
                                                        int *__restrict__ err, const int2 *__restrict__ strip_map, int *__restrict__ strip_prog,
                                                        long long *__restrict__ bases) {
    static_assert(!P16 || RW % 2 == 0, "the int16 profile packs two rows per dword");
    constexpr int HW = G64 * RW;              // rows per strip
    constexpr int SW = al64_snapw(RW);        // dwords per lane and snapshot
    constexpr int LW = P16 ? RW / 2 : RW;     // profile dwords per lane and base
    constexpr int BST = G64 * LW;             // dwords per base plane (a multiple of 32)
    constexpr int TI = 2, TD = 1;
    __shared__ int lds[32 + 5 * BST];
    const int l = threadIdx.x;
    if (l < 25) lds[l] = kp.sc4[l] - 2 * kp.e4; // rebased diagonal move: 4*(s - 2e)
    int *prof = &lds[32];
    const char *prof_bytes = reinterpret_cast<const char *>(prof); // a column's profile offset = base plane + lane stride, accumulated hop by hop (wave_shr1_add)
    int vinc;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vinc) : "s"(LW * 4));
    const int s_own = strip_map[blockIdx.x].y;
    const int n_stolen = claim_items(strip_prog + gridDim.x, 1, s_own);
    if (n_stolen < 0) return;
    const int p = strip_map[blockIdx.x].x;
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    BetaBytes bp;
    bp.init(b_buf, kp, b_start[p], pl.m);
    const int m = pl.m;
    const int Tend = (m + (G64 - 1) + 15) & ~15;
    const int OE4 = kp.oe4, E4 = kp.e4, RB = kp.e4;
    const int CKR = kp.ckc, cksh = 31 - __clz(kp.ckc); // snapshot (and rebase) spacing of this call: CK64, or a wider power of two for the walk farm (farm64.hip.h)
    int vO4;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vO4) : "s"(kp.o4));
    int bad = 0;
    const int64_t rb_pitch = (int64_t)m + 1;
    for (int s = s_own - n_stolen; s <= s_own; s++) {
        const bool store_row = s + 1 < pl.strips;
        const int row0 = s * HW + l * RW;
        int rt[RW], hold[RW];
        {
            int a5[RW];
#pragma unroll
            for (int r = 0; r < RW; r++) {
                const int i0 = row0 + r;
                int a = 0;
                if (i0 < pl.n) { a = ap[i0]; if (a >= 5) { bad = 1; a = 4; } }
                a5[r] = a * 5;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = P16 ? ((lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16)) : lds[a5[k] + b];
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < RW; r++) { // column 0: M = I = -inf, D = D00 + i*ecol, rebased (a global alignment: the same constant in every row)
            const int i = row0 + r + 1;
            const int D1c = kp.d00_4 + i * kp.ecol4 + TD - RB * i;
            hold[r] = max3i(NEG4 + 3, NEG4 + TI, D1c);
            rt[r] = max3i(NEG4 + 3 + OE4, NEG4 + TI + E4, D1c + OE4) - RB;
        }
        int diag0 = (row0 == 0) ? max3i(3, kp.o4 + TI, kp.d00_4 + TD) : max3i(NEG4 + 3, NEG4 + TI, kp.d00_4 + row0 * kp.ecol4 + TD - RB * row0);
        int dn_out = 0, h_out = 0, b_out = l * (LW * 4), sq_dn = 0, sq_h = 0;
        int qdn = 0, qh = 0, qb = 0, ndn = 0, nh = 0, nb = 0, ndd = 0;
        // moving bases (REBASE, const_long.hip.h): mine; the strip above's for the two blocks the columns being loaded were written in
        long long Bown = 0;
        int dlo = 0, dhi = 0, qp = 0, edge = CKR, r0i = kp.o4 + TI;
        bool dhi_ok = false;
        long long *my_bases = bases + pl.rowi_off + (int64_t)s * pl.s_pitch;
        // the base of block q of the strip above (a block nobody needs -- its first column lies beyond m -- is never stored: 0)
        auto bprod = [&](int q) -> long long { return (s == 0 || (int64_t)q * CKR > (int64_t)m + XB64) ? 0LL : w64_base_wait(my_bases - pl.s_pitch + q, err); };
        // lanes 0 .. 15: column c of the row above the strip as it is in memory (odn, oh; s > 0: to be settled, then shifted by odd to my base), the base of column c
        auto boundary = [&](int c, int &odn, int &oh, int &ob, int &odd) {
            odn = 0; oh = 0; ob = 0; odd = 0;
            if (l < 16 && c >= 1 && c <= m) {
                if (s == 0) {
                    const int M3 = NEG4 + 3, I2 = r0i, D1 = NEG4 + TD; // row 0: I(0,c) = gapOpen + c*gapExtend, rebased, relative to my base
                    oh = max3i(M3, I2, D1);
                    odn = max3i(M3 + OE4, I2 + OE4, D1 + E4) - RB;
                } else {
                    const int2 v = rb_load(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], true);
                    odn = v.x; oh = v.y;
                    odd = (c + XB64 >= edge) ? dhi : dlo;
                }
                ob = bp.raw(c - 1);
            }
        };
        // ... until the strip above has stored them: an entry that still reads as the launch's fill is loaded again
        auto settle = [&](int c, int &odn, int &oh, const int odd) {
            if (s > 0) {
                const bool mine = l < 16 && c >= 1 && c <= m;
                if (__any(mine && (odn == W64_SENT || oh == W64_SENT))) {
                    const long long t_begin = wall_clock64();
                    while (true) {
                        if (mine && (odn == W64_SENT || oh == W64_SENT)) {
                            const int2 v = rb_load(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], true);
                            odn = v.x; oh = v.y;
                        }
                        if (!__any(mine && (odn == W64_SENT || oh == W64_SENT))) break;
                        __builtin_amdgcn_s_sleep(4);
                        if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); if (mine && (odn == W64_SENT || oh == W64_SENT)) { odn = 0; oh = 0; } break; }
                    }
                }
                odn += odd; oh += odd;
            }
        };
        auto base_off = [&](int raw, int c) { int b = (l < 16 && c >= 1 && c <= m) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); };
        boundary(l + 1, qdn, qh, qb, ndd);
        settle(l + 1, qdn, qh, ndd);
        qb = base_off(qb, l + 1);
        int wq[LW], pb_cur;
        auto fetch = [&](int pbv, int *w) {
            const int *pw = reinterpret_cast<const int *>(prof_bytes + pbv);
#pragma unroll
            for (int k = 0; k < LW; k++) w[k] = pw[k];
        };
        pb_cur = wave_shr1_add(qb, b_out, vinc);
        qb = dpp_shl1(qb, qb);
        fetch(pb_cur, wq);
        // uc: the step's position U in its block as a compile-time constant (the unrolled blocks: the queues stay in place, lane 0 takes entry U with one
        // row_shl:U move -- no shift of the queue, no copy for the move's `old` operand), or -1 (the ramps' rolled loop: the queues shift by one per step)
        auto step = [&](const int t, auto chk, auto uc, const bool take, const int nqv) {
            constexpr bool CHECK = decltype(chk)::value;
            constexpr int U = decltype(uc)::value;
            int up_dn, up_h, pb_next;
            if constexpr (U >= 0) {
                up_dn = wave_shr1(dpp_row_shl<U>(qdn), dn_out);
                up_h = wave_shr1(dpp_row_shl<U>(qh), h_out);
                if constexpr (U == 15) { pb_next = wave_shr1_add(nqv, pb_cur, vinc); qb = dpp_shl1(nqv, nqv); } // (the base queue runs one step ahead: the next block's takes over)
                else pb_next = wave_shr1_add(dpp_row_shl<U>(qb), pb_cur, vinc);
            } else {
                up_dn = wave_shr1(qdn, dn_out);
                up_h = wave_shr1(qh, h_out);
                qdn = dpp_shl1(qdn, qdn);
                qh = dpp_shl1(qh, qh);
                if (take) qb = nqv;
                pb_next = wave_shr1_add(qb, pb_cur, vinc);
                qb = dpp_shl1(qb, qb);
            }
            int wn[LW];
            fetch(pb_next, wn);
            asm volatile("" ::: "memory"); // the reads stay HERE, ahead of the arithmetic
            const int j = t - l;
            const int *w = wq;
            if (!CHECK || (j >= 1 && j <= m)) {
                int hd = diag0, dnu = up_dn;
#pragma unroll
                for (int r = 0; r < RW - 1; r++) {
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    const int hnew = max3i(hd + S4, rt[r], dnu);
                    const int ho = hnew + vO4;
                    rt[r] = max(ho, rt[r]);
                    const int dnn = max(ho, dnu);
                    hd = hold[r];
                    hold[r] = hnew;
                    dnu = dnn;
                }
                { // the lane's last row with argmax tags: the row buffer's entries carry them (affine_long.hip.h)
                    constexpr int r = RW - 1;
                    int S4;
                    if constexpr (P16) S4 = (r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff); else S4 = w[r];
                    const int M3 = (hd | 3) + S4;
                    const int I2 = (rt[r] & ~3) | TI;
                    const int D1 = (dnu & ~3) | TD;
                    const int hnew = max3i(M3, I2, D1);
                    const int ho = hnew + vO4;
                    rt[r] = max(ho, I2);
                    dnu = max(ho, D1);
                    hold[r] = hnew;
                }
                diag0 = up_h;
                dn_out = dnu;
                h_out = hold[RW - 1];
            }
            sq_dn = dpp_shl1(dn_out, sq_dn); // (row 3 of the wave: lane 63 inserts, lanes 48 .. 63 hold the last 16 columns of the bottom row)
            sq_h = dpp_shl1(h_out, sq_h);
#pragma unroll
            for (int k = 0; k < LW; k++) wq[k] = wn[k];
            pb_cur = pb_next;
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            if (t0 > 0 && (t0 & (CKR - 1)) == 0) { // move the base, then the snapshot
                const int rep = __builtin_amdgcn_readfirstlane(hold[0]);
                const bool rb_on = t0 <= m + (G64 - 1); // (while some lane still has columns)
                const int d = rb_on ? (rep & ~3) : 0;
#pragma unroll
                for (int r = 0; r < RW; r++) { rt[r] -= d; hold[r] -= d; }
                diag0 -= d; dn_out -= d; h_out -= d; qdn -= d; qh -= d;
                Bown += d; dlo -= d; dhi -= d;
                r0i = rbase_const((long long)kp.o4 + TI, Bown);
                if (rb_on && l == 0) rbase_store(my_bases + (t0 >> cksh), Bown, true);
                if (rb_on && snap != nullptr) { // snapshot: the state the wave resumes from at step t0
                    uint4 *dst = reinterpret_cast<uint4 *>(snap + pl.ckpt_off + (((int64_t)((t0 >> cksh) - 1) * pl.strips + s) * G64 + l) * SW);
                    al64_snap_store<RW>(dst, rt, hold, diag0, dn_out);
                }
            }
            if (s > 0) { // the columns loaded now are t0 + 17 .. t0 + 32: written by the strip above in its blocks (c + XB64) / CK64
                while (t0 + 17 + XB64 >= edge) { qp++; edge += CKR; dlo = dhi_ok ? dhi : rbase_delta(bprod(qp), Bown); dhi_ok = false; }
                if (!dhi_ok && t0 + 32 + XB64 >= edge) { dhi = rbase_delta(bprod(qp + 1), Bown); dhi_ok = true; }
            }
            boundary(t0 + 16 + l + 1, ndn, nh, nb, ndd);
            if (t0 >= G64 && t0 + 16 <= m) {
                al64_unrolled_block(std::make_integer_sequence<int, 16>{}, [&](auto uc) {
                    if constexpr (decltype(uc)::value == 15) nb = base_off(nb, t0 + 16 + l + 1); // (as late as possible: the boundary loads of the block have a block's time to arrive)
                    step(t0 + decltype(uc)::value + 1, std::false_type{}, uc, decltype(uc)::value == 15, nb);
                });
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, std::integral_constant<int, -1>{}, u == 15, nb); }
            }
            asm volatile("" :: "v"(ndn), "v"(nh));
            settle(t0 + 16 + l + 1, ndn, nh, ndd);
            qdn = ndn; qh = nh;
            if (store_row) {
                const int x = l - (G64 - 16), c = t0 + x + 1 - (G64 - 1); // lanes 48 .. 63: slot x holds what lane 63 handed down at step t0 + 1 + x
                if (x >= 0 && c >= 1 && c <= m) rb_store(&rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + c], sq_dn, sq_h, true);
            }
        }
        if (m >= 1) {
#pragma unroll
            for (int r = 0; r < RW; r++) if (row0 + r + 1 == pl.n) hfin[pl.hcol_off] = (Bown + (int64_t)hold[r] + (int64_t)RB * ((int64_t)pl.n + m)) >> 2; // plain score h(n, m)
        }
    }
    if (bad) atomicOr(err, 1);
}

template <bool P16>
__global__ __launch_bounds__(64) void al64_walk_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                       const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                       const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                       KParams kp, TbParams tp, const int2 *__restrict__ rowbuf, const int *__restrict__ snap,
                                                       const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                       const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err,
                                                       const long long *__restrict__ bases, MegaState *__restrict__ mst) {
    constexpr int LW = P16 ? R / 2 : R;
    constexpr int BST = G64 * LW;
    constexpr int TI = 2, TD = 1;
    __shared__ int lds[32 + 5 * BST + AL64_DIRG + H64];
    const int l = threadIdx.x;
    if (l < 25) lds[l] = kp.sc4[l] - 2 * kp.e4;
    int *prof = &lds[32];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    unsigned *dirg = reinterpret_cast<unsigned *>(&lds[32 + 5 * BST]);
    int *hcolT = &lds[32 + 5 * BST + AL64_DIRG]; // keys h(i, m) of the strip's rows whose lanes have passed column m
    const int p = blockIdx.x;
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    BetaBytes bp;
    bp.init(b_buf, kp, b_start[p], pl.m);
    const int m = pl.m;
    const int64_t rb_pitch = (int64_t)m + 1;
    const int po = pl.src;
    const int OE4 = kp.oe4, E4 = kp.e4, RB = kp.e4;
    int vO4;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vO4) : "s"(kp.o4));
    int bad = 0;
    // walker state (lane 0), as in al_walk_kernel
    int wi = pl.n, wj = m, wk = 0, wdone = 0, pend = 1; // (pend: 1 the walk's first state, 2 a quirk-Q1 restart whose entry cell the next tile holds)
    int q1n = 0, q1c = 0;
    int64_t li = (pl.n > 0) ? (int64_t)(pl.n - 1) % tp.ci : 0;
    int64_t cnt = 0, cur_run = 0;
    int cur_op = -1, last_op = -1;
    const int64_t sbase = scr_off[p];
    auto flush_run = [&]() {
        if (cur_op >= 0) {
            gnx_cigar c; c.run_length = cur_run; c.op = (uint8_t)cur_op;
            for (int z = 0; z < 7; z++) c._pad[z] = 0;
            scr[sbase + cnt] = c;
            cnt++;
        }
    };
    auto emit = [&](int op, int64_t run) {
        if (op == cur_op) cur_run += run;
        else { flush_run(); cur_op = op; cur_run = run; }
    };
    // row panels (MegaState, const_long.hip.h)
    int virt = 0;
    int64_t row_off = 0;
    bool pexit = false;
    if (mst) {
        virt = mst->virt; row_off = mst->row_off;
        if (mst->resume) { wi = mst->wi; wj = mst->wj; wk = mst->wk; pend = mst->pend; li = mst->li; cnt = mst->cnt; cur_run = mst->cur_run; cur_op = mst->cur_op; last_op = mst->last_op; }
        else li = (pl.n > 0) ? ((int64_t)pl.n + row_off - 1) % tp.ci : 0;
    }

    int s_prof = -1;
    while (true) {
        const int ci = __builtin_amdgcn_readfirstlane(wi), cj = __builtin_amdgcn_readfirstlane(wj);
        if (__builtin_amdgcn_readfirstlane(wdone)) break;
        if (virt > 0 && ci <= virt) { pexit = true; break; }
        const int s = (ci - 1) / H64;
        const int lw = (ci - 1 - s * H64) / R;
        const int te = cj + lw;               // step of the cell the walk is at
        const int c = (te >= 3) ? (te - 3) / CK64 : 0; // (the first two steps after a snapshot carry no usable plane fields, see al_walk_kernel)
        const int tbeg = c * CK64;
        const int tmin = c > 0 ? 2 : 0;
        const int tend = te + 1;              // one step further: quirk Q1
        const int nblk = (tend - tbeg + 15) >> 4;
        const int row0 = s * H64 + l * R;
        int rt[R], hold[R];
        unsigned acc[3 * R];
        __syncthreads(); // table visible; the previous round's walk is over
        if (s != s_prof) { // the strip's profile (most rounds stay in the strip of the round before: 640 rows against ~116 cells per tile)
            int a5[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i0 = row0 + r;
                int a = 0;
                if (i0 < pl.n) { a = ap[i0]; if (a >= 5) { bad = 1; a = 4; } }
                a5[r] = a * 5;
            }
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = P16 ? ((lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16)) : lds[a5[k] + b];
            }
            __syncthreads();
            s_prof = s;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = row0 + r + 1;
            const int D1c = kp.d00_4 + i * kp.ecol4 + TD - RB * i;
            hold[r] = max3i(NEG4 + 3, NEG4 + TI, D1c);
            rt[r] = max3i(NEG4 + 3 + OE4, NEG4 + TI + E4, D1c + OE4) - RB;
            acc[r] = 0; acc[R + r] = 0; acc[2 * R + r] = 0;
        }
        int diag0 = (row0 == 0) ? max3i(3, kp.o4 + TI, kp.d00_4 + TD) : max3i(NEG4 + 3, NEG4 + TI, kp.d00_4 + row0 * kp.ecol4 + TD - RB * row0);
        int dn_out = 0, h_out = 0, b_out = 0;
        if (c > 0) { // resume from the snapshot of step tbeg
            const uint4 *sp = reinterpret_cast<const uint4 *>(snap + pl.ckpt_off + (((int64_t)(c - 1) * pl.strips + s) * G64 + l) * AL_SNAPW);
            const uint4 x0 = sp[0], x1 = sp[1], x2 = sp[2], x3 = sp[3], x4 = sp[4], x5 = sp[5];
            rt[0] = (int)x0.x; rt[1] = (int)x0.y; rt[2] = (int)x0.z; rt[3] = (int)x0.w; rt[4] = (int)x1.x; rt[5] = (int)x1.y; rt[6] = (int)x1.z; rt[7] = (int)x1.w;
            rt[8] = (int)x2.x; rt[9] = (int)x2.y; hold[0] = (int)x2.z; hold[1] = (int)x2.w; hold[2] = (int)x3.x; hold[3] = (int)x3.y; hold[4] = (int)x3.z; hold[5] = (int)x3.w;
            hold[6] = (int)x4.x; hold[7] = (int)x4.y; hold[8] = (int)x4.z; hold[9] = (int)x4.w; diag0 = (int)x5.x; dn_out = (int)x5.y;
            h_out = hold[R - 1];
            const int jb = tbeg - l;
            if (jb >= 1 && jb <= m) { int b = bp.at(jb - 1); if (b >= 5) { bad = 1; b = 4; } b_out = b * (BST * 4); }
        }
        int qdn, qh, qb, ndn = 0, nh = 0, nb = 0;
        // the snapshot's keys are relative to the strip's base of block c; the row above, block by block, to the bases of the strip above
        long long Bt = 0;
        if (c > 0) Bt = bases[pl.rowi_off + (int64_t)s * pl.s_pitch + c];
        const int r0i = rbase_const((long long)kp.o4 + TI, Bt);
        auto boundary = [&](int cc, int &odn, int &oh, int &ob) {
            odn = 0; oh = 0; ob = 0;
            if (l < 16 && cc >= 1 && cc <= m) {
                if (s == 0) {
                    const int M3 = NEG4 + 3, I2 = r0i, D1 = NEG4 + TD;
                    oh = max3i(M3, I2, D1);
                    odn = max3i(M3 + OE4, I2 + OE4, D1 + E4) - RB;
                } else {
                    const int2 v = rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + cc];
                    const int q = (cc + XB64) / CK64;
                    const int dd = rbase_delta(bases[pl.rowi_off + (int64_t)(s - 1) * pl.s_pitch + q], Bt);
                    odn = v.x + dd; oh = v.y + dd;
                }
                ob = bp.raw(cc - 1);
            }
        };
        auto base_off = [&](int raw, int cc) { int b = (l < 16 && cc >= 1 && cc <= m) ? bp.value(raw, cc - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); };
        boundary(tbeg + l + 1, qdn, qh, qb);
        qb = base_off(qb, tbeg + l + 1);
        int wq[LW], pb_cur;
        auto fetch = [&](int pbv, int *w) {
            const int *pw = reinterpret_cast<const int *>(prof_lane + pbv);
#pragma unroll
            for (int k = 0; k < LW; k++) w[k] = pw[k];
        };
        pb_cur = wave_shr1(qb, b_out);
        qb = dpp_shl1(qb, qb);
        fetch(pb_cur, wq);
        auto step = [&](const int t, auto chk, const bool take, const int nqv) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_dn = wave_shr1(qdn, dn_out);
            const int up_h = wave_shr1(qh, h_out);
            qdn = dpp_shl1(qdn, qdn);
            qh = dpp_shl1(qh, qh);
            if (take) qb = nqv;
            const int pb_next = wave_shr1(qb, pb_cur);
            qb = dpp_shl1(qb, qb);
            int wn[LW];
            fetch(pb_next, wn);
            asm volatile("" ::: "memory");
            const int j = t - l;
            const int *w = wq;
            if (!CHECK || (j >= 1 && j <= m)) {
                int hd = diag0, dnu = up_dn;
#pragma unroll
                for (int r = 0; r < R; r++) { // the recording h-form of fill_affine_kernel (rebased keys)
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    acc[r] = alignbit2((unsigned)hd, acc[r]);
                    acc[R + r] = alignbit2((unsigned)rt[r], acc[R + r]);
                    acc[2 * R + r] = alignbit2((unsigned)dnu, acc[2 * R + r]);
                    const int M3 = (hd | 3) + S4;
                    const int I2 = (rt[r] & ~3) | TI;
                    const int D1 = (dnu & ~3) | TD;
                    const int hnew = max3i(M3, I2, D1);
                    const int ho = hnew + vO4;
                    rt[r] = max(ho, I2);
                    const int dnn = max(ho, D1);
                    hd = hold[r];
                    hold[r] = hnew;
                    dnu = dnn;
                }
                diag0 = up_h;
                dn_out = dnu;
                h_out = hold[R - 1];
            }
#pragma unroll
            for (int k = 0; k < LW; k++) wq[k] = wn[k];
            pb_cur = pb_next;
        };
        for (int b = 0; b < nblk; b++) {
            const int t0 = tbeg + 16 * b;
            boundary(t0 + 16 + l + 1, ndn, nh, nb);
            if (t0 >= G64 && t0 + 16 <= m) {
#pragma unroll
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::false_type{}, u == 15, nb); }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, u == 15, nb); }
            }
            qdn = ndn; qh = nh;
            const int miss = (t0 + 16 - l) - m;
            const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
#pragma unroll
                for (int r = 0; r < R; r++) dirg[((b * 3 + k) * R + r) * G64 + l] = acc[k * R + r] >> sh;
            }
            if (b == nblk - 1 && t0 + 16 - l >= m) { // lanes that have passed column m hold h(i, m) of their rows
#pragma unroll
                for (int r = 0; r < R; r++) hcolT[l * R + r] = hold[r];
            }
        }
        __syncthreads();
        if (l == 0) {
            int i = wi, j = wj, k = wk;
            if (pend) { const int kn = 3 - (hcolT[i - 1 - s * H64] & 3); if (pend == 2) { q1n++; q1c += (kn != k); } k = kn; pend = 0; }
            while (true) {
                if (i == 0 || j == 0) { wdone = 1; break; }
                const int i0 = i - 1 - s * H64;
                if (i0 < 0) break; // left the strip through its top edge
                const int l2 = i0 / R, r2 = i0 - l2 * R;
                const int t1 = j + l2 - 1 - tbeg;
                if (t1 < tmin) break; // left the (usable part of the) tile through its skewed left edge
                const int pos = t1 & 15;
                const unsigned w = dirg[(((t1 >> 4) * 3 + k) * R + r2) * G64 + l2];
                int tag = (int)((w >> (2 * pos)) & 3u);
                if (tag == 0) { atomicOr(err, 2); wdone = 1; break; }
                if (k == 1) { // horizontal run inside this word, see traceback_kernel
                    int avail = min(pos + 1, j);
                    if (t1 < 16) avail = min(avail, pos - tmin + 1);
                    unsigned x = w ^ 0xAAAAAAAAu;
                    if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
                    const int lowcut = pos + 1 - avail;
                    if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
                    int steps;
                    if (x == 0) steps = avail;
                    else {
                        const int pnz = (31 - __clz((int)x)) >> 1;
                        tag = (int)((w >> (2 * pnz)) & 3u);
                        if (tag == 0) { atomicOr(err, 2); wdone = 1; break; }
                        steps = pos - pnz + 1;
                        k = 3 - tag;
                    }
                    emit(1, steps); j -= steps; last_op = 1;
                    continue;
                }
                emit(k, 1);
                last_op = k;
                const bool up_exit = (li == 0);
                li = up_exit ? tp.ci - 1 : li - 1;
                i--;
                if (k == 0) j--;
                k = 3 - tag;
                const int kt = k; // (the traced state)
                if (up_exit && i > 0 && j > 0) { // quirk Q1 (affineGap.go:305): restart in the argmax state of the entry cell (i, j)
                    if (j < m) { // = the M-plane field of (i+1, j+1): the row the walk just left, at most one step past its cell
                        const int l3 = (i0) / R, r3 = i0 - l3 * R, t3 = (j + 1) + l3 - 1 - tbeg;
                        const unsigned w3 = dirg[(((t3 >> 4) * 3 + 0) * R + r3) * G64 + l3];
                        k = 3 - (int)((w3 >> (2 * (t3 & 15))) & 3u);
                    } else if (i - 1 - s * H64 >= 0 && m + (i - 1 - s * H64) / R - 1 - tbeg >= tmin) k = 3 - (hcolT[i - 1 - s * H64] & 3);
                    else pend = 2; // row i belongs to the strip above, or its lane passed column m before this tile began: the next tile has it
                    if (pend != 2) { q1n++; q1c += (k != kt); }
                }
            }
            wi = i; wj = j; wk = k;
        }
    }
    if (l == 0) q1_report(q1n, q1c);
    if (l == 0 && mst) {
        mst->wi = wi; mst->wj = wj; mst->wk = wk; mst->pend = pend; mst->li = li; mst->cnt = cnt; mst->cur_run = cur_run; mst->cur_op = cur_op; mst->last_op = last_op;
        mst->done = pexit ? 0 : 1;
    }
    if (l == 0 && !pexit) {
        // Step 4 (affineGap.go:135-139) -- quirk Q2 when the corner is not the origin
        const int64_t gi = (int64_t)wi + (wi > 0 ? row_off : 0);
        const bool up_exit = (last_op != 1) && (gi % tp.ci == 0);
        const bool left_exit = (last_op != 2) && ((int64_t)wj % tp.cj == 0);
        if (!up_exit && left_exit) emit(2, gi);
        else if (up_exit && !left_exit) emit(1, wj);
        flush_run();
        nops[po] = cnt;
        score_out[po] = hfin[pl.hcol_off];
    }
    if (bad) atomicOr(err, 1);
}

template <bool P16>
__global__ __launch_bounds__(128) void al64_walk2_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                        const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                        const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                        KParams kp, TbParams tp, const int2 *__restrict__ rowbuf, const int *__restrict__ snap,
                                                        const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                        const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err,
                                                        const long long *__restrict__ bases, MegaState *__restrict__ mst) {
    constexpr int LW = P16 ? R / 2 : R;
    constexpr int BST = G64 * LW;
    constexpr int TI = 2, TD = 1;
    __shared__ int lds[32 + 5 * BST + 2 * AL64_DIRG + 2 * H64 + 8];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63; // wave 0: the walk's tile; wave 1: its left neighbour
    if (threadIdx.x < 25) lds[threadIdx.x] = kp.sc4[threadIdx.x] - 2 * kp.e4;
    int *prof = &lds[32];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    unsigned *dirg0 = reinterpret_cast<unsigned *>(&lds[32 + 5 * BST]), *dirg = dirg0 + w * AL64_DIRG; // the tile this wave fills
    int *hcol0 = &lds[32 + 5 * BST + 2 * AL64_DIRG], *hcolT = hcol0 + w * H64; // keys h(i, m) of the strip's rows whose lanes have passed column m
    int *xch = &lds[32 + 5 * BST + 2 * AL64_DIRG + 2 * H64]; // {row, column, done} of the walk, from thread 0 to everybody
    const int p = blockIdx.x;
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    BetaBytes bp;
    bp.init(b_buf, kp, b_start[p], pl.m);
    const int m = pl.m;
    const int64_t rb_pitch = (int64_t)m + 1;
    const int po = pl.src;
    const int OE4 = kp.oe4, E4 = kp.e4, RB = kp.e4;
    int vO4;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vO4) : "s"(kp.o4));
    int bad = 0;
    // walker state (lane 0), as in al_walk_kernel
    int wi = pl.n, wj = m, wk = 0, wdone = 0, pend = 1; // (pend: 1 the walk's first state, 2 a quirk-Q1 restart whose entry cell the next tile holds)
    int q1n = 0, q1c = 0;
    int64_t li = (pl.n > 0) ? (int64_t)(pl.n - 1) % tp.ci : 0;
    int64_t cnt = 0, cur_run = 0;
    int cur_op = -1, last_op = -1;
    const int64_t sbase = scr_off[p];
    auto flush_run = [&]() {
        if (cur_op >= 0) {
            gnx_cigar c; c.run_length = cur_run; c.op = (uint8_t)cur_op;
            for (int z = 0; z < 7; z++) c._pad[z] = 0;
            scr[sbase + cnt] = c;
            cnt++;
        }
    };
    auto emit = [&](int op, int64_t run) {
        if (op == cur_op) cur_run += run;
        else { flush_run(); cur_op = op; cur_run = run; }
    };
    // row panels (MegaState, const_long.hip.h)
    int virt = 0;
    int64_t row_off = 0;
    bool pexit = false;
    if (mst) {
        virt = mst->virt; row_off = mst->row_off;
        if (mst->resume) { wi = mst->wi; wj = mst->wj; wk = mst->wk; pend = mst->pend; li = mst->li; cnt = mst->cnt; cur_run = mst->cur_run; cur_op = mst->cur_op; last_op = mst->last_op; }
        else li = (pl.n > 0) ? ((int64_t)pl.n + row_off - 1) % tp.ci : 0;
    }

    if (threadIdx.x == 0) { xch[0] = wi; xch[1] = wj; xch[2] = 0; }
    int s_prof = -1;
    while (true) {
        __syncthreads(); // table and walk state visible; the previous round's walk is over
        const int ci = xch[0], cj = xch[1];
        if (xch[2]) break;
        if (virt > 0 && ci <= virt) { pexit = true; break; }
        const int s = (ci - 1) / H64;
        const int lw = (ci - 1 - s * H64) / R;
        const int te = cj + lw;               // step of the cell the walk is at
        const int cA = (te >= 3) ? (te - 3) / CK64 : 0; // (the first two steps after a snapshot carry no usable plane fields, see al_walk_kernel)
        // Wave 1 re-fills the tile to the LEFT of the walk's at the same time: a walk that leaves its tile through the skewed left edge (the usual
        // exit: 640 rows against ~116 cells of path per tile) enters that one at step tbeg .. tbeg + 2 of its own tile, whatever its path was
        const bool act = w == 0 || cA >= 1;
        const int c = w == 0 ? cA : max(cA - 1, 0);
        const int tbeg = c * CK64;
        const int tmin = c > 0 ? 2 : 0;
        const int tend = w == 0 ? te + 1 : cA * CK64 + 3; // one step further: quirk Q1
        const int nblk = act ? (tend - tbeg + 15) >> 4 : 0;
        const int row0 = s * H64 + l * R;
        int rt[R], hold[R];
        unsigned acc[3 * R];
        if (s != s_prof && w == 0) { // the strip's profile (most rounds stay in the strip of the round before: 640 rows against ~116 cells per tile)
            int a5[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i0 = row0 + r;
                int a = 0;
                if (i0 < pl.n) { a = ap[i0]; if (a >= 5) { bad = 1; a = 4; } }
                a5[r] = a * 5;
            }
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = P16 ? ((lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16)) : lds[a5[k] + b];
            }
        }
        if (s != s_prof) { __syncthreads(); s_prof = s; }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = row0 + r + 1;
            const int D1c = kp.d00_4 + i * kp.ecol4 + TD - RB * i;
            hold[r] = max3i(NEG4 + 3, NEG4 + TI, D1c);
            rt[r] = max3i(NEG4 + 3 + OE4, NEG4 + TI + E4, D1c + OE4) - RB;
            acc[r] = 0; acc[R + r] = 0; acc[2 * R + r] = 0;
        }
        int diag0 = (row0 == 0) ? max3i(3, kp.o4 + TI, kp.d00_4 + TD) : max3i(NEG4 + 3, NEG4 + TI, kp.d00_4 + row0 * kp.ecol4 + TD - RB * row0);
        int dn_out = 0, h_out = 0, b_out = 0;
        if (act && c > 0) { // resume from the snapshot of step tbeg
            const uint4 *sp = reinterpret_cast<const uint4 *>(snap + pl.ckpt_off + (((int64_t)(c - 1) * pl.strips + s) * G64 + l) * AL_SNAPW);
            const uint4 x0 = sp[0], x1 = sp[1], x2 = sp[2], x3 = sp[3], x4 = sp[4], x5 = sp[5];
            rt[0] = (int)x0.x; rt[1] = (int)x0.y; rt[2] = (int)x0.z; rt[3] = (int)x0.w; rt[4] = (int)x1.x; rt[5] = (int)x1.y; rt[6] = (int)x1.z; rt[7] = (int)x1.w;
            rt[8] = (int)x2.x; rt[9] = (int)x2.y; hold[0] = (int)x2.z; hold[1] = (int)x2.w; hold[2] = (int)x3.x; hold[3] = (int)x3.y; hold[4] = (int)x3.z; hold[5] = (int)x3.w;
            hold[6] = (int)x4.x; hold[7] = (int)x4.y; hold[8] = (int)x4.z; hold[9] = (int)x4.w; diag0 = (int)x5.x; dn_out = (int)x5.y;
            h_out = hold[R - 1];
            const int jb = tbeg - l;
            if (jb >= 1 && jb <= m) { int b = bp.at(jb - 1); if (b >= 5) { bad = 1; b = 4; } b_out = b * (BST * 4); }
        }
        int qdn, qh, qb, ndn = 0, nh = 0, nb = 0;
        // the snapshot's keys are relative to the strip's base of block c; the row above, block by block, to the bases of the strip above
        long long Bt = 0;
        if (act && c > 0) Bt = bases[pl.rowi_off + (int64_t)s * pl.s_pitch + c];
        const int r0i = rbase_const((long long)kp.o4 + TI, Bt);
        auto boundary = [&](int cc, int &odn, int &oh, int &ob) {
            odn = 0; oh = 0; ob = 0;
            if (l < 16 && cc >= 1 && cc <= m) {
                if (s == 0) {
                    const int M3 = NEG4 + 3, I2 = r0i, D1 = NEG4 + TD;
                    oh = max3i(M3, I2, D1);
                    odn = max3i(M3 + OE4, I2 + OE4, D1 + E4) - RB;
                } else {
                    const int2 v = rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + cc];
                    const int q = (cc + XB64) / CK64;
                    const int dd = rbase_delta(bases[pl.rowi_off + (int64_t)(s - 1) * pl.s_pitch + q], Bt);
                    odn = v.x + dd; oh = v.y + dd;
                }
                ob = bp.raw(cc - 1);
            }
        };
        auto base_off = [&](int raw, int cc) { int b = (l < 16 && cc >= 1 && cc <= m) ? bp.value(raw, cc - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); };
        boundary(tbeg + l + 1, qdn, qh, qb);
        qb = base_off(qb, tbeg + l + 1);
        int wq[LW], pb_cur;
        auto fetch = [&](int pbv, int *w) {
            const int *pw = reinterpret_cast<const int *>(prof_lane + pbv);
#pragma unroll
            for (int k = 0; k < LW; k++) w[k] = pw[k];
        };
        pb_cur = wave_shr1(qb, b_out);
        qb = dpp_shl1(qb, qb);
        fetch(pb_cur, wq);
        auto step = [&](const int t, auto chk, const bool take, const int nqv) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_dn = wave_shr1(qdn, dn_out);
            const int up_h = wave_shr1(qh, h_out);
            qdn = dpp_shl1(qdn, qdn);
            qh = dpp_shl1(qh, qh);
            if (take) qb = nqv;
            const int pb_next = wave_shr1(qb, pb_cur);
            qb = dpp_shl1(qb, qb);
            int wn[LW];
            fetch(pb_next, wn);
            asm volatile("" ::: "memory");
            const int j = t - l;
            const int *w = wq;
            if (!CHECK || (j >= 1 && j <= m)) {
                int hd = diag0, dnu = up_dn;
#pragma unroll
                for (int r = 0; r < R; r++) { // the recording h-form of fill_affine_kernel (rebased keys)
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    acc[r] = alignbit2((unsigned)hd, acc[r]);
                    acc[R + r] = alignbit2((unsigned)rt[r], acc[R + r]);
                    acc[2 * R + r] = alignbit2((unsigned)dnu, acc[2 * R + r]);
                    const int M3 = (hd | 3) + S4;
                    const int I2 = (rt[r] & ~3) | TI;
                    const int D1 = (dnu & ~3) | TD;
                    const int hnew = max3i(M3, I2, D1);
                    const int ho = hnew + vO4;
                    rt[r] = max(ho, I2);
                    const int dnn = max(ho, D1);
                    hd = hold[r];
                    hold[r] = hnew;
                    dnu = dnn;
                }
                diag0 = up_h;
                dn_out = dnu;
                h_out = hold[R - 1];
            }
#pragma unroll
            for (int k = 0; k < LW; k++) wq[k] = wn[k];
            pb_cur = pb_next;
        };
        for (int b = 0; b < nblk; b++) {
            const int t0 = tbeg + 16 * b;
            boundary(t0 + 16 + l + 1, ndn, nh, nb);
            if (t0 >= G64 && t0 + 16 <= m) {
#pragma unroll
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::false_type{}, u == 15, nb); }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, u == 15, nb); }
            }
            qdn = ndn; qh = nh;
            const int miss = (t0 + 16 - l) - m;
            const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
#pragma unroll
                for (int r = 0; r < R; r++) dirg[((b * 3 + k) * R + r) * G64 + l] = acc[k * R + r] >> sh;
            }
            if (b == nblk - 1 && t0 + 16 - l >= m) { // lanes that have passed column m hold h(i, m) of their rows
#pragma unroll
                for (int r = 0; r < R; r++) hcolT[l * R + r] = hold[r];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // walk inside one tile; returns 0: left through the skewed left edge, 1: through the top edge of the strip, 2: the walk is over (or failed)
            auto walk_tile = [&](const unsigned *dgX, const int *hcX, const int tbX, const int tmX) -> int {
                int why = 2;
                int i = wi, j = wj, k = wk;
                if (pend) { const int kn = 3 - (hcX[i - 1 - s * H64] & 3); if (pend == 2) { q1n++; q1c += (kn != k); } k = kn; pend = 0; }
                while (true) {
                    if (i == 0 || j == 0) { wdone = 1; break; }
                    const int i0 = i - 1 - s * H64;
                    if (i0 < 0) { why = 1; break; } // left the strip through its top edge
                    const int l2 = i0 / R, r2 = i0 - l2 * R;
                    const int t1 = j + l2 - 1 - tbX;
                    if (t1 < tmX) { why = 0; break; } // left the (usable part of the) tile through its skewed left edge
                    const int pos = t1 & 15;
                    const unsigned w = dgX[(((t1 >> 4) * 3 + k) * R + r2) * G64 + l2];
                    int tag = (int)((w >> (2 * pos)) & 3u);
                    if (tag == 0) { atomicOr(err, 2); wdone = 1; break; }
                    if (k == 1) { // horizontal run inside this word, see traceback_kernel
                        int avail = min(pos + 1, j);
                        if (t1 < 16) avail = min(avail, pos - tmX + 1);
                        unsigned x = w ^ 0xAAAAAAAAu;
                        if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
                        const int lowcut = pos + 1 - avail;
                        if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
                        int steps;
                        if (x == 0) steps = avail;
                        else {
                            const int pnz = (31 - __clz((int)x)) >> 1;
                            tag = (int)((w >> (2 * pnz)) & 3u);
                            if (tag == 0) { atomicOr(err, 2); wdone = 1; break; }
                            steps = pos - pnz + 1;
                            k = 3 - tag;
                        }
                        emit(1, steps); j -= steps; last_op = 1;
                        continue;
                    }
                    emit(k, 1);
                    last_op = k;
                    const bool up_exit = (li == 0);
                    li = up_exit ? tp.ci - 1 : li - 1;
                    i--;
                    if (k == 0) j--;
                    k = 3 - tag;
                    const int kt = k; // (the traced state)
                    if (up_exit && i > 0 && j > 0) { // quirk Q1 (affineGap.go:305): restart in the argmax state of the entry cell (i, j)
                        if (j < m) { // = the M-plane field of (i+1, j+1): the row the walk just left, at most one step past its cell
                            const int l3 = (i0) / R, r3 = i0 - l3 * R, t3 = (j + 1) + l3 - 1 - tbX;
                            const unsigned w3 = dgX[(((t3 >> 4) * 3 + 0) * R + r3) * G64 + l3];
                            k = 3 - (int)((w3 >> (2 * (t3 & 15))) & 3u);
                        } else if (i - 1 - s * H64 >= 0 && m + (i - 1 - s * H64) / R - 1 - tbX >= tmX) k = 3 - (hcX[i - 1 - s * H64] & 3);
                        else pend = 2; // row i belongs to the strip above, or its lane passed column m before this tile began: the next tile has it
                        if (pend != 2) { q1n++; q1c += (k != kt); }
                    }
                }
                wi = i; wj = j; wk = k;
                return why;
            };
            const int tbA = cA * CK64;
            int why = walk_tile(dirg0, hcol0, tbA, cA > 0 ? 2 : 0);
            if (why == 0 && cA >= 1 && !pend && !wdone) why = walk_tile(dirg0 + AL64_DIRG, hcol0 + H64, tbA - CK64, cA > 1 ? 2 : 0);
            xch[0] = wi; xch[1] = wj; xch[2] = wdone;
        }
    }
    if (threadIdx.x == 0) q1_report(q1n, q1c);
    if (threadIdx.x == 0 && mst) {
        mst->wi = wi; mst->wj = wj; mst->wk = wk; mst->pend = pend; mst->li = li; mst->cnt = cnt; mst->cur_run = cur_run; mst->cur_op = cur_op; mst->last_op = last_op;
        mst->done = pexit ? 0 : 1;
    }
    if (threadIdx.x == 0 && !pexit) {
        // Step 4 (affineGap.go:135-139) -- quirk Q2 when the corner is not the origin
        const int64_t gi = (int64_t)wi + (wi > 0 ? row_off : 0);
        const bool up_exit = (last_op != 1) && (gi % tp.ci == 0);
        const bool left_exit = (last_op != 2) && ((int64_t)wj % tp.cj == 0);
        if (!up_exit && left_exit) emit(2, gi);
        else if (up_exit && !left_exit) emit(1, wj);
        flush_run();
        nops[po] = cnt;
        score_out[po] = hfin[pl.hcol_off];
    }
    if (bad) atomicOr(err, 1);
}


} // namespace
