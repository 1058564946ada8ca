// const_long64.hip.h -- the snapshot path of const_long.hip.h with the WHOLE WAVE on one pair: 64 lanes x 10 rows, strips of 640 rows
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.13.
#pragma once
#include "const_long.hip.h"
#include "affine_long64.hip.h" // G64, H64, XB64, wave_shr1

namespace {
// ------------------------------------------------------------------------------------------------------
// The constant-gap twin of affine_long64.hip.h (align/constGap.go:13-68 -- cmd/globalAlignment's one ConstGap call on two whole
// sequences, globalAlignment.go:84): cl_sweep_kernel / cl_walk_kernel<NP = 1> with 64 lanes per pair instead of 16.
//   cl64_sweep_kernel  score only (add, max3 per cell), the look-ahead step of the un-piped form (a launch of one pair has less than one
//                      wave per SIMD: nothing else hides the LDS round trip) on piped strips; bottom rows (4 B per column and 640 rows),
//                      a snapshot every CKC64 = 224 steps (12 dwords per lane), bases[strip][block]; block of column c: (c + 62) / CKC64.
//   cl64_walk_kernel   one wave per pair: the tile (strip, <= 224 steps) re-filled into one 2-bit plane in LDS (36 KB), walked by lane 0.
// Always on moving bases.
// ------------------------------------------------------------------------------------------------------
constexpr int CKC64 = CKC_SMALL;

// Rows per lane of the 64-lane CONSTANT-GAP sweep and of the farm's re-fills (round 6): RW = 4 / 10 -- strips of 256 / 640 rows.  cmd/globalAlignment's one ConstGap call on
// 150 kb x 180 kb is 235 strips of 640 rows: a quarter of a wave per SIMD; at RW = 4 it is 586 waves whose step is 8 + 9 instead of 20 + 9 instructions.  The host picks
// (w64_pick_rows, gnx_align.hip); the one-workgroup walks below stay at R = 10.
__host__ __device__ constexpr int cl64_snapw(int rw) { return (rw + 1 + 3) & ~3; } // dwords per lane and snapshot: val[RW], diag0 (RW = 10: SNAPW)
static_assert(cl64_snapw(R) == SNAPW, "the walk kernels of this file read the sweep's snapshots at RW = R");
template <int RW>
__device__ __forceinline__ void cl64_snap_store(uint4 *dst, const int (&val)[RW], int diag0) {
    constexpr int SW = cl64_snapw(RW);
    unsigned v[SW];
#pragma unroll
    for (int r = 0; r < RW; r++) v[r] = (unsigned)val[r];
    v[RW] = (unsigned)diag0;
#pragma unroll
    for (int q = RW + 1; q < SW; q++) v[q] = 0u;
#pragma unroll
    for (int q = 0; q < SW / 4; q++) dst[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
template <int RW>
__device__ __forceinline__ void cl64_snap_load(const uint4 *sp, int (&val)[RW], int &diag0) {
    constexpr int SW = cl64_snapw(RW);
    unsigned v[SW];
    uint4 x[SW / 4];
#pragma unroll
    for (int q = 0; q < SW / 4; q++) x[q] = sp[q];
#pragma unroll
    for (int q = 0; q < SW / 4; q++) { v[4 * q] = x[q].x; v[4 * q + 1] = x[q].y; v[4 * q + 2] = x[q].z; v[4 * q + 3] = x[q].w; }
#pragma unroll
    for (int r = 0; r < RW; r++) val[r] = (int)v[r];
    diag0 = (int)v[RW];
}

template <int RW, bool P16>
__global__ __launch_bounds__(64) void cl64_sweep_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                        const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                        const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                        KParams kp, int *__restrict__ rowbuf, int *__restrict__ snap, int64_t *__restrict__ hfin,
                                                        int *__restrict__ err, const int2 *__restrict__ strip_map, int *__restrict__ strip_prog,
                                                        long long *__restrict__ bases) {
    static_assert(!P16 || RW % 2 == 0, "the int16 profile packs two rows per dword");
    constexpr int HW = G64 * RW, SW = cl64_snapw(RW);
    constexpr int LW = P16 ? RW / 2 : RW;
    constexpr int BST = G64 * LW;
    __shared__ int lds[32 + 5 * BST];
    const int l = threadIdx.x;
    if (l < 25) lds[l] = kp.sc4[l] - 2 * kp.g4; // rebased diagonal move: 4*(s - 2g); every value carries tag 2
    int *prof = &lds[32];
    const char *prof_bytes = reinterpret_cast<const char *>(prof); // a column's profile offset = base plane + lane stride, accumulated hop by hop (wave_shr1_add, affine_long64.hip.h)
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
    int bad = 0;
    const int64_t rb_pitch = (int64_t)m + 1;
    for (int s = s_own - n_stolen; s <= s_own; s++) {
        const bool store_row = s + 1 < pl.strips;
        const int row0 = s * HW + l * RW;
        int val[RW];
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
        for (int r = 0; r < RW; r++) val[r] = 2; // column 0, rebased: 0 (tag 2)
        int diag0 = 2;
        int v_out = 0, b_out = l * (LW * 4), sq_v = 0;
        int qv = 0, qb = 0, nv = 0, nb = 0, ndd = 0;
        long long Bown = 0;
        int dlo = 0, dhi = 0, qp = 0, edge = CKC64, r0v = 2;
        bool dhi_ok = false;
        long long *my_bases = bases + pl.rowi_off + (int64_t)s * pl.s_pitch;
        // (the hand-over without a progress word: affine_long64.hip.h, W64_SENT)
        auto bprod = [&](int q) -> long long { return (s == 0 || (int64_t)q * CKC64 > (int64_t)m + XB64) ? 0LL : w64_base_wait(my_bases - pl.s_pitch + q, err); };
        auto boundary = [&](int c, int &ov, int &ob, int &odd) { // lanes 0 .. 15: column c of the row above the strip as it is in memory (s > 0: to be settled, then shifted by odd), the (raw) base of column c
            ov = 0; ob = 0; odd = 0;
            if (l < 16 && c >= 1 && c <= m) {
                if (s == 0) ov = r0v;
                else { ov = rb_load32(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], true); odd = (c + XB64 >= edge) ? dhi : dlo; }
                ob = bp.raw(c - 1);
            }
        };
        auto settle = [&](int c, int &ov, const int odd) {
            if (s > 0) {
                const bool mine = l < 16 && c >= 1 && c <= m;
                if (__any(mine && ov == W64_SENT)) {
                    const long long t_begin = wall_clock64();
                    while (true) {
                        if (mine && ov == W64_SENT) ov = rb_load32(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], true);
                        if (!__any(mine && ov == W64_SENT)) break;
                        __builtin_amdgcn_s_sleep(4);
                        if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); if (mine && ov == W64_SENT) ov = 0; break; }
                    }
                }
                ov += odd;
            }
        };
        auto base_off = [&](int raw, int c) { int b = (l < 16 && c >= 1 && c <= m) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); };
        boundary(l + 1, qv, qb, ndd);
        settle(l + 1, qv, ndd);
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
        // uc: the step's position in its block as a compile-time constant (unrolled blocks: the queues stay in place, lane 0 takes its entry with one row_shl move), or -1 (the ramps)
        auto step = [&](const int t, auto chk, auto uc, const bool take, const int nqv) {
            constexpr bool CHECK = decltype(chk)::value;
            constexpr int U = decltype(uc)::value;
            int up_v, pb_next;
            if constexpr (U >= 0) {
                up_v = wave_shr1(dpp_row_shl<U>(qv), v_out);
                if constexpr (U == 15) { pb_next = wave_shr1_add(nqv, pb_cur, vinc); qb = dpp_shl1(nqv, nqv); }
                else pb_next = wave_shr1_add(dpp_row_shl<U>(qb), pb_cur, vinc);
            } else {
                up_v = wave_shr1(qv, v_out);
                qv = dpp_shl1(qv, qv);
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
                int vd = diag0, vu = up_v;
#pragma unroll
                for (int r = 0; r < RW; r++) {
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    const int k = max3i(vd + S4, val[r], vu);
                    vd = val[r];
                    val[r] = k;
                    vu = k;
                }
                diag0 = up_v;
                v_out = vu;
            }
            sq_v = dpp_shl1(v_out, sq_v); // (row 3 of the wave: lane 63 inserts, lanes 48 .. 63 hold the last 16 columns of the bottom row)
#pragma unroll
            for (int k = 0; k < LW; k++) wq[k] = wn[k];
            pb_cur = pb_next;
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            if (t0 > 0 && t0 % CKC64 == 0) { // move the base, then the snapshot
                const int rep = __builtin_amdgcn_readfirstlane(val[0]);
                const bool rb_on = t0 <= m + (G64 - 1);
                const int d = rb_on ? (rep & ~3) : 0;
#pragma unroll
                for (int r = 0; r < RW; r++) val[r] -= d;
                diag0 -= d; v_out -= d; qv -= d;
                Bown += d; dlo -= d; dhi -= d;
                r0v = rbase_const(2, Bown);
                if (rb_on && l == 0) rbase_store(my_bases + t0 / CKC64, Bown, true);
                if (rb_on && snap != nullptr) {
                    uint4 *dst = reinterpret_cast<uint4 *>(snap + pl.ckpt_off + (((int64_t)(t0 / CKC64 - 1) * pl.strips + s) * G64 + l) * SW);
                    cl64_snap_store<RW>(dst, val, diag0);
                }
            }
            if (s > 0) { // the columns loaded now are t0 + 17 .. t0 + 32: written by the strip above in its blocks (c + XB64) / CKC64
                while (t0 + 17 + XB64 >= edge) { qp++; edge += CKC64; dlo = dhi_ok ? dhi : rbase_delta(bprod(qp), Bown); dhi_ok = false; }
                if (!dhi_ok && t0 + 32 + XB64 >= edge) { dhi = rbase_delta(bprod(qp + 1), Bown); dhi_ok = true; }
            }
            boundary(t0 + 16 + l + 1, nv, nb, ndd);
            if (t0 >= G64 && t0 + 16 <= m) {
                al64_unrolled_block(std::make_integer_sequence<int, 16>{}, [&](auto uc) {
                    if constexpr (decltype(uc)::value == 15) nb = base_off(nb, t0 + 16 + l + 1);
                    step(t0 + decltype(uc)::value + 1, std::false_type{}, uc, decltype(uc)::value == 15, nb);
                });
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, std::integral_constant<int, -1>{}, u == 15, nb); }
            }
            asm volatile("" :: "v"(nv));
            settle(t0 + 16 + l + 1, nv, ndd);
            qv = nv;
            if (store_row) {
                const int x = l - (G64 - 16), c = t0 + x + 1 - (G64 - 1); // lanes 48 .. 63: slot x holds what lane 63 handed down at step t0 + 1 + x
                if (x >= 0 && c >= 1 && c <= m) rb_store32(&rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + c], sq_v, true);
            }
        }
        if (m >= 1) {
#pragma unroll
            for (int r = 0; r < RW; r++) if (row0 + r + 1 == pl.n) hfin[pl.hcol_off] = (Bown >> 2) + (int64_t)(val[r] >> 2) + (int64_t)(kp.g4 >> 2) * ((int64_t)pl.n + m); // plain V(n, m)
        }
    }
    if (bad) atomicOr(err, 1);
}

template <bool P16>
__global__ __launch_bounds__(64) void cl64_walk_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                       const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                       const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                       KParams kp, TbParams tp, const int *__restrict__ rowbuf, const int *__restrict__ snap,
                                                       const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                       const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err,
                                                       const long long *__restrict__ bases, MegaState *__restrict__ mst) {
    constexpr int LW = P16 ? R / 2 : R;
    constexpr int BST = G64 * LW;
    constexpr int CK = CKC64;
    constexpr int DIRG = (CK / 16) * R * G64;
    __shared__ int lds[32 + 5 * BST + DIRG];
    const int l = threadIdx.x;
    if (l < 25) lds[l] = kp.sc4[l] - 2 * kp.g4 + 1; // pre-tagged diagonal candidate (tag 3), see fill_const_kernel
    int *prof = &lds[32];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    unsigned *dirg = reinterpret_cast<unsigned *>(&lds[32 + 5 * BST]);
    const int p = blockIdx.x;
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    BetaBytes bp;
    bp.init(b_buf, kp, b_start[p], pl.m);
    const int m = pl.m;
    const int64_t rb_pitch = (int64_t)m + 1;
    const int po = pl.src;
    int bad = 0;
    int wi = pl.n, wj = m, wdone = 0;
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
    int virt = 0;
    int64_t row_off = 0;
    bool pexit = false;
    if (mst) {
        virt = mst->virt; row_off = mst->row_off;
        if (mst->resume) { wi = mst->wi; wj = mst->wj; cnt = mst->cnt; cur_run = mst->cur_run; cur_op = mst->cur_op; last_op = mst->last_op; }
    }

    int s_prof = -1;
    while (true) {
        const int ci = __builtin_amdgcn_readfirstlane(wi), cj = __builtin_amdgcn_readfirstlane(wj);
        if (__builtin_amdgcn_readfirstlane(wdone)) break;
        if (virt > 0 && ci <= virt) { pexit = true; break; }
        const int s = (ci - 1) / H64;
        const int lw = (ci - 1 - s * H64) / R;
        const int tend = cj + lw;   // step of the cell the walk is at
        const int c = (tend - 1) / CK;
        const int tbeg = c * CK;
        const int nblk = (tend - tbeg + 15) >> 4;
        const int row0 = s * H64 + l * R;
        int val[R];
        unsigned acc[R];
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
        int diag0 = 2;
#pragma unroll
        for (int r = 0; r < R; r++) { val[r] = 2; acc[r] = 0; }
        int v_out = 0, b_out = 0;
        if (c > 0) { // resume from the snapshot of step tbeg
            const uint4 *sp = reinterpret_cast<const uint4 *>(snap + pl.ckpt_off + (((int64_t)(c - 1) * pl.strips + s) * G64 + l) * SNAPW);
            const uint4 x0 = sp[0], x1 = sp[1], x2 = sp[2];
            val[0] = (int)x0.x; val[1] = (int)x0.y; val[2] = (int)x0.z; val[3] = (int)x0.w;
            val[4] = (int)x1.x; val[5] = (int)x1.y; val[6] = (int)x1.z; val[7] = (int)x1.w;
            val[8] = (int)x2.x; val[9] = (int)x2.y; diag0 = (int)x2.z;
            v_out = val[R - 1];
            const int jb = tbeg - l; // the column this lane processed at step tbeg: its base goes to the next lane
            if (jb >= 1 && jb <= m) { int b = bp.at(jb - 1); if (b >= 5) { bad = 1; b = 4; } b_out = b * (BST * 4); }
        }
        int qv, qb, nv = 0, nb = 0;
        long long Bt = 0;
        if (c > 0) Bt = bases[pl.rowi_off + (int64_t)s * pl.s_pitch + c];
        const int r0v = rbase_const(2, Bt);
        auto boundary = [&](int cc, int &ov, int &ob) {
            ov = 0;
            int b = 0;
            if (l < 16 && cc >= 1 && cc <= m) {
                if (s == 0) ov = r0v;
                else {
                    const int q = (cc + XB64) / CK;
                    ov = rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + cc] + rbase_delta(bases[pl.rowi_off + (int64_t)(s - 1) * pl.s_pitch + q], Bt);
                }
                b = bp.at(cc - 1);
                if (b >= 5) { bad = 1; b = 4; }
            }
            ob = b * (BST * 4);
        };
        boundary(tbeg + l + 1, qv, qb);
        auto step = [&](const int t, auto chk) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_v = wave_shr1(qv, v_out);
            const int pb = wave_shr1(qb, b_out);
            qv = dpp_shl1(qv, qv);
            qb = dpp_shl1(qb, qb);
            const int j = t - l;
            b_out = pb;
            if (!CHECK || (j >= 1 && j <= m)) {
                const int *pw = reinterpret_cast<const int *>(prof_lane + pb);
                int w[LW];
#pragma unroll
                for (int k = 0; k < LW; k++) w[k] = pw[k];
                int vd = diag0, vu = up_v;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    const int k = max3i(vd + S4, val[r], vu - 1);
                    acc[r] = alignbit2((unsigned)k, acc[r]);
                    vd = val[r];
                    val[r] = (k & ~3) | 2;
                    vu = val[r];
                }
                diag0 = up_v;
                v_out = vu;
            }
        };
        for (int b = 0; b < nblk; b++) {
            const int t0 = tbeg + 16 * b;
            boundary(t0 + 16 + l + 1, nv, nb);
            if (t0 >= G64 && t0 + 16 <= m) {
#pragma unroll
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::false_type{});
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::true_type{});
            }
            qv = nv; qb = nb;
            const int miss = (t0 + 16 - l) - m; // steps this lane sat idle after its last column
            const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
            for (int r = 0; r < R; r++) dirg[(b * R + r) * G64 + l] = acc[r] >> sh;
        }
        __syncthreads();
        if (l == 0) {
            int i = wi, j = wj;
            while (true) {
                if (i == 0 || j == 0) { wdone = 1; break; }
                const int i0 = i - 1 - s * H64;
                if (i0 < 0) break; // left the strip through its top edge
                const int l2 = i0 / R, r2 = i0 - l2 * R;
                const int t1 = j + l2 - 1 - tbeg;
                if (t1 < 0) break; // left the tile through its (skewed) left edge
                const int pos = t1 & 15;
                const unsigned w = dirg[((t1 >> 4) * R + r2) * G64 + l2];
                int tag = (int)((w >> (2 * pos)) & 3u);
                if (tag == 0) { atomicOr(err, 2); wdone = 1; break; }
                const int op = 3 - tag;
                if (op == 1) { // horizontal run: count the fields "came from the left" below pos with one xor + clz
                    const int avail = min(pos + 1, j);
                    unsigned x = w ^ 0xAAAAAAAAu;
                    if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
                    const int lowcut = pos + 1 - avail;
                    if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
                    int steps;
                    if (x == 0) steps = avail;
                    else {
                        const int pnz = (31 - __clz((int)x)) >> 1;
                        if (((w >> (2 * pnz)) & 3u) == 0) { atomicOr(err, 2); wdone = 1; break; }
                        steps = pos - pnz;
                    }
                    emit(1, steps); j -= steps; last_op = 1;
                    continue;
                }
                emit(op, 1);
                last_op = op;
                i--;
                if (op == 0) j--;
            }
            wi = i; wj = j;
        }
    }
    if (l == 0 && mst) {
        mst->wi = wi; mst->wj = wj; mst->cnt = cnt; mst->cur_run = cur_run; mst->cur_op = cur_op; mst->last_op = last_op; mst->done = pexit ? 0 : 1;
    }
    if (l == 0 && !pexit) {
        // Step 4 (constGap.go:59-63), quirk Q2
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
__global__ __launch_bounds__(128) void cl64_walk2_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                        const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                        const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                        KParams kp, TbParams tp, const int *__restrict__ rowbuf, const int *__restrict__ snap,
                                                        const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                        const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err,
                                                        const long long *__restrict__ bases, MegaState *__restrict__ mst) {
    constexpr int LW = P16 ? R / 2 : R;
    constexpr int BST = G64 * LW;
    constexpr int CK = CKC64;
    constexpr int DIRG = (CK / 16) * R * G64;
    __shared__ int lds[32 + 5 * BST + 2 * DIRG + 8];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63; // wave 0: the walk's tile; wave 1: its left neighbour (see al64_walk2_kernel)
    if (threadIdx.x < 25) lds[threadIdx.x] = kp.sc4[threadIdx.x] - 2 * kp.g4 + 1; // pre-tagged diagonal candidate (tag 3), see fill_const_kernel
    int *prof = &lds[32];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    unsigned *dirg0 = reinterpret_cast<unsigned *>(&lds[32 + 5 * BST]), *dirg = dirg0 + w * DIRG; // the tile this wave fills
    int *xch = &lds[32 + 5 * BST + 2 * DIRG]; // {row, column, done} of the walk, from thread 0 to everybody
    const int p = blockIdx.x;
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    BetaBytes bp;
    bp.init(b_buf, kp, b_start[p], pl.m);
    const int m = pl.m;
    const int64_t rb_pitch = (int64_t)m + 1;
    const int po = pl.src;
    int bad = 0;
    int wi = pl.n, wj = m, wdone = 0;
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
    int virt = 0;
    int64_t row_off = 0;
    bool pexit = false;
    if (mst) {
        virt = mst->virt; row_off = mst->row_off;
        if (mst->resume) { wi = mst->wi; wj = mst->wj; cnt = mst->cnt; cur_run = mst->cur_run; cur_op = mst->cur_op; last_op = mst->last_op; }
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
        const int te = cj + lw;     // step of the cell the walk is at
        const int cA = (te - 1) / CK;
        // wave 1: the whole tile to the left -- a walk that leaves its tile through the skewed left edge enters that one at its last step or the one before
        const bool act = w == 0 || cA >= 1;
        const int c = w == 0 ? cA : max(cA - 1, 0);
        const int tbeg = c * CK;
        const int tend = w == 0 ? te : cA * CK;
        const int nblk = act ? (tend - tbeg + 15) >> 4 : 0;
        const int row0 = s * H64 + l * R;
        int val[R];
        unsigned acc[R];
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
        int diag0 = 2;
#pragma unroll
        for (int r = 0; r < R; r++) { val[r] = 2; acc[r] = 0; }
        int v_out = 0, b_out = 0;
        if (act && c > 0) { // resume from the snapshot of step tbeg
            const uint4 *sp = reinterpret_cast<const uint4 *>(snap + pl.ckpt_off + (((int64_t)(c - 1) * pl.strips + s) * G64 + l) * SNAPW);
            const uint4 x0 = sp[0], x1 = sp[1], x2 = sp[2];
            val[0] = (int)x0.x; val[1] = (int)x0.y; val[2] = (int)x0.z; val[3] = (int)x0.w;
            val[4] = (int)x1.x; val[5] = (int)x1.y; val[6] = (int)x1.z; val[7] = (int)x1.w;
            val[8] = (int)x2.x; val[9] = (int)x2.y; diag0 = (int)x2.z;
            v_out = val[R - 1];
            const int jb = tbeg - l; // the column this lane processed at step tbeg: its base goes to the next lane
            if (jb >= 1 && jb <= m) { int b = bp.at(jb - 1); if (b >= 5) { bad = 1; b = 4; } b_out = b * (BST * 4); }
        }
        int qv, qb, nv = 0, nb = 0;
        long long Bt = 0;
        if (act && c > 0) Bt = bases[pl.rowi_off + (int64_t)s * pl.s_pitch + c];
        const int r0v = rbase_const(2, Bt);
        auto boundary = [&](int cc, int &ov, int &ob) {
            ov = 0;
            int b = 0;
            if (l < 16 && cc >= 1 && cc <= m) {
                if (s == 0) ov = r0v;
                else {
                    const int q = (cc + XB64) / CK;
                    ov = rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + cc] + rbase_delta(bases[pl.rowi_off + (int64_t)(s - 1) * pl.s_pitch + q], Bt);
                }
                b = bp.at(cc - 1);
                if (b >= 5) { bad = 1; b = 4; }
            }
            ob = b * (BST * 4);
        };
        boundary(tbeg + l + 1, qv, qb);
        auto step = [&](const int t, auto chk) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_v = wave_shr1(qv, v_out);
            const int pb = wave_shr1(qb, b_out);
            qv = dpp_shl1(qv, qv);
            qb = dpp_shl1(qb, qb);
            const int j = t - l;
            b_out = pb;
            if (!CHECK || (j >= 1 && j <= m)) {
                const int *pw = reinterpret_cast<const int *>(prof_lane + pb);
                int w[LW];
#pragma unroll
                for (int k = 0; k < LW; k++) w[k] = pw[k];
                int vd = diag0, vu = up_v;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    const int k = max3i(vd + S4, val[r], vu - 1);
                    acc[r] = alignbit2((unsigned)k, acc[r]);
                    vd = val[r];
                    val[r] = (k & ~3) | 2;
                    vu = val[r];
                }
                diag0 = up_v;
                v_out = vu;
            }
        };
        for (int b = 0; b < nblk; b++) {
            const int t0 = tbeg + 16 * b;
            boundary(t0 + 16 + l + 1, nv, nb);
            if (t0 >= G64 && t0 + 16 <= m) {
#pragma unroll
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::false_type{});
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::true_type{});
            }
            qv = nv; qb = nb;
            const int miss = (t0 + 16 - l) - m; // steps this lane sat idle after its last column
            const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
            for (int r = 0; r < R; r++) dirg[(b * R + r) * G64 + l] = acc[r] >> sh;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // walk inside one tile; returns 0: left through the skewed left edge, 1: through the top edge of the strip, 2: the walk is over (or failed)
            auto walk_tile = [&](const unsigned *dgX, const int tbX) -> int {
                int why = 2;
                int i = wi, j = wj;
                while (true) {
                    if (i == 0 || j == 0) { wdone = 1; break; }
                    const int i0 = i - 1 - s * H64;
                    if (i0 < 0) { why = 1; break; } // left the strip through its top edge
                    const int l2 = i0 / R, r2 = i0 - l2 * R;
                    const int t1 = j + l2 - 1 - tbX;
                    if (t1 < 0) { why = 0; break; } // left the tile through its (skewed) left edge
                    const int pos = t1 & 15;
                    const unsigned w = dgX[((t1 >> 4) * R + r2) * G64 + l2];
                    int tag = (int)((w >> (2 * pos)) & 3u);
                    if (tag == 0) { atomicOr(err, 2); wdone = 1; break; }
                    const int op = 3 - tag;
                    if (op == 1) { // horizontal run: count the fields "came from the left" below pos with one xor + clz
                        const int avail = min(pos + 1, j);
                        unsigned x = w ^ 0xAAAAAAAAu;
                        if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
                        const int lowcut = pos + 1 - avail;
                        if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
                        int steps;
                        if (x == 0) steps = avail;
                        else {
                            const int pnz = (31 - __clz((int)x)) >> 1;
                            if (((w >> (2 * pnz)) & 3u) == 0) { atomicOr(err, 2); wdone = 1; break; }
                            steps = pos - pnz;
                        }
                        emit(1, steps); j -= steps; last_op = 1;
                        continue;
                    }
                    emit(op, 1);
                    last_op = op;
                    i--;
                    if (op == 0) j--;
                }
                wi = i; wj = j;
                return why;
            };
            int why = walk_tile(dirg0, cA * CK);
            if (why == 0 && cA >= 1 && !wdone) why = walk_tile(dirg0 + DIRG, (cA - 1) * CK);
            xch[0] = wi; xch[1] = wj; xch[2] = wdone;
        }
    }
    if (threadIdx.x == 0 && mst) {
        mst->wi = wi; mst->wj = wj; mst->cnt = cnt; mst->cur_run = cur_run; mst->cur_op = cur_op; mst->last_op = last_op; mst->done = pexit ? 0 : 1;
    }
    if (threadIdx.x == 0 && !pexit) {
        // Step 4 (constGap.go:59-63), quirk Q2
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
