// farm64.hip.h -- the walk of ONE long pair on many CUs: tiles re-filled ahead of the walk, a round at a time
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).
#pragma once
#include "affine_long64.hip.h"
#include "const_long64.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// al64_walk_kernel / cl64_walk_kernel alternate, on one wave (two with the speculative left tile), "re-fill the tile the walk is in" (640 rows x
// <= 132 steps of the recording recurrence: ~50 us of a lone wave) and "walk ~116 cells": 0.54 s per 1e6 cells of path, 60 % of a 1 Mb x 1 Mb
// call (align/affineGap.go:59-68 is what cmd/cigarToBed/cigarToBed.go:86 calls; one pair, nothing else on the device).  But a re-fill does not
// depend on the path at all -- snapshot + boundary row in, direction planes out -- only WHICH tiles are needed does, and at the scale of a tile
// a global alignment is a line.  So the walk becomes ROUNDS on the call's stream, no host round trip inside a batch of them:
//   *_farm_fill_body     one wave re-fills tile {strip, block} of the round's set completely (every step of the block, all planes) into
//                        planes[set][q] in global memory (69 KB, L2-resident) -- the fill code of al64_walk_kernel / cl64_walk_kernel.
//   farm_walk_body       one workgroup per pair: while the tile of the walk's cell is one of the set, copy the <= 24 lanes of it above the cell
//                        into LDS (4 waves, every load in flight at once), walk it on wave 0 with the walk state in SGPRs (quirks Q1 / Q2,
//                        MegaState for row panels: the automaton of al64_walk_kernel); all 64 lanes read the fields of the next 64 cells of the
//                        diagonal at once, the walk takes the leading run of "from the diagonal" fields in one step and the word of the cell the
//                        run ends at serves the scalar step that follows.  Then it asks for the next set: the tiles of the straight line through
//                        the cell it stopped at whose direction is what the walk did lately (farm_predict).
//   overlapped rounds    (default) ONE launch a round, *_farm_round_kernel: workgroup 0 walks set `par` while workgroups 1 .. k re-fill the other
//                        set on their own CUs; the set the walk asks for continues the line behind the set being re-filled.
//   plain rounds         (GNX_W64_FARM_PIPE=0) two launches a round: re-fill set 0, walk set 0.
// A tile that was not asked for ends the walk's round; the walk's own tile is the first of the next set (overlapped: of one of the next two),
// so the rounds always move, and a wrong guess costs the rest of one round -- never a result: the planes are the ones a round of al64_walk_kernel
// computes (same snapshot, same recurrence; the steps beyond the walk's are never read), the walk is the same automaton.
// GNX_W64_FARM=0: off (the one-workgroup walks); =k: k tiles per round (default: 24 for launches of one or two AffineGap pairs since round 6 -- at 512-step tiles a round is bound by its re-fills, so more of them side by side pay --, 16 otherwise; at most 32).  GNX_W64_CK=128 / 256 / 512: the affine snapshot spacing (below).
// Measured (profiles/r5_long_pairs.jsonl, r5_experiments.md section 10): AffineGap 1 Mb x 1 Mb walk 538 -> 41 ms (call 0.86 -> 0.35 s, workspace 87 -> 31 GB),
// 340 kb x 340 kb 0.30 -> 0.089 s, 2 Mb x 2 Mb 3.5 -> 1.1 s (no row panels any more), ConstGap 150 kb x 180 kb 0.092 -> 0.036 s, 300 kb x 2 Mb (450 000 runs: the
// scalar steps of the walk) 0.57 -> 0.43 s.
// ------------------------------------------------------------------------------------------------------
// the fill of a tile is ONE wave: its LDS traffic is ordered by a fence, not a workgroup barrier (the overlapped rounds run it inside 256-thread workgroups)
#define FARM_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
constexpr int FARM_MAX = 32; // tiles per round and pair, at most
constexpr int FARM_NLC = 24; // lanes (of 10 rows) of a tile the walker holds in LDS (a diagonal leaves a 128-step tile after ~15)
constexpr int FARM_WW = 18;  // ... and words of 16 steps (affine; a diagonal through 24 lanes crosses 240 + 24 steps = 17 words)

static_assert(sizeof(gnx_cigar) == 16, "farm_walk_body stores a run as one 16-byte word");
struct FarmCtl {
    int32_t fin, rounds, hits, pad0;
    int32_t acc_i, acc_j;      // rows / columns the walk moved lately (each round: halved, plus the round's)
    int32_t n[2];              // tiles of the two sets (the plain rounds use set 0 only)
    int2 tile[2][FARM_MAX];    // {strip, block}
};

// The affine sweep's snapshot spacing is a run-time power of two for the farm (KParams::ckc): its re-fills run beside the walk, so a tile of 512 steps costs
// the walk nothing per cell of path -- and the snapshots of a pair shrink from 0.075 to 0.019 B per cell (1 Mb x 1 Mb: 75 -> 19 GB, and what a process's
// first call pays to allocate them; 2 Mb x 2 Mb fits the device without row panels).  The constant-gap tiles keep CKC64 = 224 steps (12 dwords per lane and snapshot).
constexpr int FARM_CK_MAX = 512; // (the walk's LDS window: 24 lanes x 33 words x 3 planes x 10 rows = 95 KB)
constexpr int farm_nlc(int rw) { return 240 / rw; } // ... = 24 lanes of 10 rows: the window's 240 rows whatever the lane holds (RW = 6 / 8 / 16: 40 / 30 / 15 lanes)
template <bool AFF, int RW = R>
struct FarmGeo {
    static constexpr int HW = G64 * RW;                                       // rows per strip
    static constexpr int NPL = AFF ? 3 : 1;                                   // planes
    static constexpr int WORDS_MAX = AFF ? FARM_CK_MAX / 16 + 1 : CKC64 / 16; // direction words per plane row, at most
    static constexpr int ROWS_MAX = WORDS_MAX * NPL * RW;
    int ck, sh;                                                               // snapshot spacing in steps; its log2 (affine)
    __host__ __device__ __forceinline__ explicit FarmGeo(int ckr) : ck(AFF ? ckr : CKC64), sh(0) { while ((1 << sh) < ck) sh++; }
    __host__ __device__ __forceinline__ int words() const { return AFF ? (ck >> 4) + 1 : CKC64 / 16; }
    __host__ __device__ __forceinline__ int rows() const { return words() * NPL * RW; }                 // plane rows of G64 dwords
    __host__ __device__ __forceinline__ int tile_dw() const { return rows() * G64 + HW; }              // + the keys h(i, m) of the rows that have passed column m (affine)
    __device__ __forceinline__ int block_of(int te) const { return AFF ? (te >= 3 ? (te - 3) >> sh : 0) : (te - 1) / CKC64; }
    __device__ __forceinline__ int tbeg_of(int c) const { return AFF ? c << sh : c * CKC64; }
    static __device__ __forceinline__ int tmin_of(int c) { return AFF ? (c > 0 ? 2 : 0) : 0; }
};

// the tiles of the straight line through cell (i, j) whose direction is what the walk did lately ((da, db) rows / columns; the diagonal while
// nothing is known), the cell's own tile first.  A line, not the diagonal: a global alignment of sequences of unequal length (300 kb x 2 Mb: 450 000
// runs) is a staircase that looks like a line of its mean slope at the scale of a tile.  Float arithmetic decides where the line leaves a tile;
// what comes out is only a guess at the tiles worth re-filling -- a wrong one costs its re-fill, never a result.
template <bool AFF, int RW, typename Skip>
__device__ __forceinline__ int farm_predict(const FarmGeo<AFF, RW> geo, int2 *tile, int i, int j, const int virt, const int nt, const bool store, const int da, const int db, Skip skip) {
    float fa = 1.0f, fb = 1.0f;
    if (da > 0 || db > 0) { const float mx = (float)max(da, db); fa = (float)da / mx; fb = (float)db / mx; }
    constexpr int HW = G64 * RW;
    const float den = fb + fa * (1.0f / RW); // steps of the wavefront the line crosses per unit
    int n = 0, ps = -1, pc = -1;
    bool taking = false;
    for (int guard = 0; guard < 6 * nt && n < nt && i > 0 && j > 0 && !(virt > 0 && i <= virt); guard++) {
        const int s = (i - 1) / HW, i0 = i - 1 - s * HW, lw = i0 / RW, te = j + lw;
        const int c = geo.block_of(te), tbeg = geo.tbeg_of(c), tmin = FarmGeo<AFF, RW>::tmin_of(c);
        if (s == ps && c == pc) { // (rounding left the line inside the tile it was to leave)
            if (fb >= fa) j -= 2; else i -= 2;
            continue;
        }
        ps = s; pc = c;
        // overlapped rounds: the line's leading tiles that are in the set being re-filled right now are what the next walk will have; from the first one
        // it will not have on, EVERY tile is taken -- a set is a contiguous piece of the line (the next walk stops at that tile, and the one after
        // it finds nothing of today's other set left), so one wrong guess costs the rest of one round and the sets are in step again
        if (taking || !skip(s, c)) {
            taking = true;
            if (store) tile[n] = make_int2(s, c);
            n++;
        }
        // units until the line is above the strip (xt), left of column 1 (xc), or at a step of the tile before tmin (xl)
        const float xl = (float)(te - 1 - tbeg - tmin + 1) / den;
        const float xt = fa > 0.0f ? (float)(i0 + 1) / fa : 3.0e9f;
        const float xc = fb > 0.0f ? (float)j / fb : 3.0e9f;
        const float x = fminf(xl, fminf(xt, xc));
        int di = (int)(fa * x + 0.999f), dj = (int)(fb * x + 0.999f);
        if (di + dj == 0) { di = fa >= fb; dj = fb > fa; }
        i -= min(di, i0 + 1); j -= dj;
    }
    return n;
}

template <bool AFF, int RW>
__global__ __launch_bounds__(64) void farm_init_kernel(const PairPlan *__restrict__ plans, int n_pairs, TbParams tp, MegaState *__restrict__ mst, FarmCtl *__restrict__ ctl, int nt, int ckr) {
    const FarmGeo<AFF, RW> geo(ckr);
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n_pairs) return;
    MegaState *st = mst + p;
    const PairPlan pl = plans[p];
    if (!st->resume) {
        st->wi = pl.n; st->wj = pl.m; st->wk = 0; st->pend = 1;
        st->li = (pl.n > 0) ? ((int64_t)pl.n + st->row_off - 1) % tp.ci : 0;
        st->cnt = 0; st->cur_run = 0; st->cur_op = -1; st->last_op = -1;
    }
    st->done = 0;
    FarmCtl *cp = ctl + p;
    cp->fin = 0; cp->rounds = 0; cp->hits = 0; cp->acc_i = 0; cp->acc_j = 0;
    const int n0 = farm_predict<AFF, RW>(geo, cp->tile[0], st->wi, st->wj, st->virt, nt, true, 0, 0, [](int, int) { return false; });
    cp->n[0] = n0;
    // (overlapped rounds: the second set continues the line behind the first)
    cp->n[1] = farm_predict<AFF, RW>(geo, cp->tile[1], st->wi, st->wj, st->virt, nt, true, 0, 0, [&](int s, int c) { for (int x = 0; x < n0; x++) if (cp->tile[0][x].x == s && cp->tile[0][x].y == c) return true; return false; });
}

// ---- affine: the fill of al64_walk_kernel, tile {s, c} completely, planes to global memory ----
template <int RW, bool P16>
__device__ __forceinline__ void al64_farm_fill_body(const PairPlan *__restrict__ plans,
                                                    const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                    const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                    const KParams &kp, const int2 *__restrict__ rowbuf, const int *__restrict__ snap,
                                                    int *__restrict__ err, const long long *__restrict__ bases,
                                                    const FarmCtl *__restrict__ ctl, unsigned *__restrict__ planes, const int p, const int q, const int par) {
    const FarmGeo<true, RW> geo(kp.ckc);
    constexpr int HW = G64 * RW, SW = al64_snapw(RW);
    constexpr int LW = P16 ? RW / 2 : RW;
    constexpr int BST = G64 * LW;
    constexpr int TI = 2, TD = 1;
    __shared__ int lds[32 + 5 * BST];
    const FarmCtl *cp = ctl + p;
    if (cp->fin || q >= cp->n[par]) return;
    const int s = cp->tile[par][q].x, c = cp->tile[par][q].y;
    unsigned *dirg = planes + (((int64_t)p * 2 + par) * FARM_MAX + q) * geo.tile_dw();
    int *hcolT = reinterpret_cast<int *>(dirg + geo.rows() * G64);
    const int l = threadIdx.x;
    if (l < 25) lds[l] = kp.sc4[l] - 2 * kp.e4;
    int *prof = &lds[32];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    BetaBytes bp;
    bp.init(b_buf, kp, b_start[p], pl.m);
    const int m = pl.m;
    const int64_t rb_pitch = (int64_t)m + 1;
    const int OE4 = kp.oe4, E4 = kp.e4, RB = kp.e4;
    int vO4;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vO4) : "s"(kp.o4));
    int bad = 0;
    const int tbeg = c << geo.sh;
    const int nblk = min(geo.words(), (m + G64 - tbeg + 15) >> 4); // (no walk stands beyond step m + 63)
    const int row0 = s * HW + l * RW;
    int rt[RW], hold[RW];
    unsigned acc[3 * RW];
    FARM_WAVE_SYNC();
    {
        int a5[RW];
#pragma unroll
        for (int r = 0; r < RW; r++) {
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
        FARM_WAVE_SYNC();
    }
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int i = row0 + r + 1;
        const int D1c = kp.d00_4 + i * kp.ecol4 + TD - RB * i;
        hold[r] = max3i(NEG4 + 3, NEG4 + TI, D1c);
        rt[r] = max3i(NEG4 + 3 + OE4, NEG4 + TI + E4, D1c + OE4) - RB;
        acc[r] = 0; acc[RW + r] = 0; acc[2 * RW + r] = 0;
    }
    int diag0 = (row0 == 0) ? max3i(3, kp.o4 + TI, kp.d00_4 + TD) : max3i(NEG4 + 3, NEG4 + TI, kp.d00_4 + row0 * kp.ecol4 + TD - RB * row0);
    int dn_out = 0, h_out = 0, b_out = 0;
    if (c > 0) { // resume from the snapshot of step tbeg
        al64_snap_load<RW>(reinterpret_cast<const uint4 *>(snap + pl.ckpt_off + (((int64_t)(c - 1) * pl.strips + s) * G64 + l) * SW), rt, hold, diag0, dn_out);
        h_out = hold[RW - 1];
        const int jb = tbeg - l;
        if (jb >= 1 && jb <= m) { int b = bp.at(jb - 1); if (b >= 5) { bad = 1; b = 4; } b_out = b * (BST * 4); }
    }
    int qdn, qh, qb, ndn = 0, nh = 0, nb = 0;
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
                const int qq = (cc + XB64) >> geo.sh;
                const int dd = rbase_delta(bases[pl.rowi_off + (int64_t)(s - 1) * pl.s_pitch + qq], Bt);
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
            for (int r = 0; r < RW; r++) { // the recording h-form of fill_affine_kernel (rebased keys)
                const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                acc[r] = alignbit2((unsigned)hd, acc[r]);
                acc[RW + r] = alignbit2((unsigned)rt[r], acc[RW + r]);
                acc[2 * RW + r] = alignbit2((unsigned)dnu, acc[2 * RW + r]);
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
            h_out = hold[RW - 1];
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
            for (int r = 0; r < RW; r++) dirg[((b * 3 + k) * RW + r) * G64 + l] = acc[k * RW + r] >> sh;
        }
        if (b == nblk - 1 && t0 + 16 - l >= m) { // lanes that have passed column m hold h(i, m) of their rows
#pragma unroll
            for (int r = 0; r < RW; r++) hcolT[l * RW + r] = hold[r];
        }
    }
    if (bad) atomicOr(err, 1);
}

// ---- constant gap: the fill of cl64_walk_kernel, tile {s, c} completely, its one plane to global memory ----
template <int RW, bool P16>
__device__ __forceinline__ void cl64_farm_fill_body(const PairPlan *__restrict__ plans,
                                                    const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                    const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                    const KParams &kp, const int *__restrict__ rowbuf, const int *__restrict__ snap,
                                                    int *__restrict__ err, const long long *__restrict__ bases,
                                                    const FarmCtl *__restrict__ ctl, unsigned *__restrict__ planes, const int p, const int q, const int par) {
    const FarmGeo<false, RW> geo(0);
    constexpr int HW = G64 * RW, SW = cl64_snapw(RW);
    constexpr int LW = P16 ? RW / 2 : RW;
    constexpr int BST = G64 * LW;
    constexpr int CK = CKC64;
    __shared__ int lds[32 + 5 * BST];
    const FarmCtl *cp = ctl + p;
    if (cp->fin || q >= cp->n[par]) return;
    const int s = cp->tile[par][q].x, c = cp->tile[par][q].y;
    unsigned *dirg = planes + (((int64_t)p * 2 + par) * FARM_MAX + q) * geo.tile_dw();
    const int l = threadIdx.x;
    if (l < 25) lds[l] = kp.sc4[l] - 2 * kp.g4 + 1; // pre-tagged diagonal candidate (tag 3), see fill_const_kernel
    int *prof = &lds[32];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    BetaBytes bp;
    bp.init(b_buf, kp, b_start[p], pl.m);
    const int m = pl.m;
    const int64_t rb_pitch = (int64_t)m + 1;
    int bad = 0;
    const int tbeg = c * CK;
    const int nblk = min(CK / 16, (m + (G64 - 1) - tbeg + 15) >> 4); // (no walk stands beyond step m + 63)
    const int row0 = s * HW + l * RW;
    int val[RW];
    unsigned acc[RW];
    FARM_WAVE_SYNC();
    {
        int a5[RW];
#pragma unroll
        for (int r = 0; r < RW; r++) {
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
        FARM_WAVE_SYNC();
    }
    int diag0 = 2;
#pragma unroll
    for (int r = 0; r < RW; r++) { val[r] = 2; acc[r] = 0; }
    int v_out = 0, b_out = 0;
    if (c > 0) { // resume from the snapshot of step tbeg
        cl64_snap_load<RW>(reinterpret_cast<const uint4 *>(snap + pl.ckpt_off + (((int64_t)(c - 1) * pl.strips + s) * G64 + l) * SW), val, diag0);
        v_out = val[RW - 1];
        const int jb = tbeg - l;
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
                const int qq = (cc + XB64) / CK;
                ov = rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + cc] + rbase_delta(bases[pl.rowi_off + (int64_t)(s - 1) * pl.s_pitch + qq], Bt);
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
            for (int r = 0; r < RW; r++) {
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
        for (int r = 0; r < RW; r++) dirg[(b * RW + r) * G64 + l] = acc[r] >> sh;
    }
    if (bad) atomicOr(err, 1);
}

__device__ __forceinline__ int64_t farm_rfl64(int64_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffLL));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ---- the walk of one round: wave 0 walks (state uniform: SGPRs), all four waves copy the tile's window into LDS ----
// par: the set this round walks (and, walked, replaces by a new one).  PIPE: the other set is being re-filled by this launch's other workgroups --
// the new set continues the line behind it.
template <bool AFF, int RW, bool PIPE>
__device__ __forceinline__ void farm_walk_body(const PairPlan *__restrict__ plans, const TbParams &tp,
                                               const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                               const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err,
                                               MegaState *__restrict__ mst_all, FarmCtl *__restrict__ ctl_all,
                                               const unsigned *__restrict__ planes, const int nt, const int ckr, const int p, const int par) {
    const FarmGeo<AFF, RW> geo(ckr);
    constexpr int HW = G64 * RW;
    constexpr int NPL = FarmGeo<AFF, RW>::NPL, NLC = farm_nlc(RW);
    constexpr int WW = AFF ? FARM_WW : CKC64 / 16; // direction words (16 steps each) of a tile the window holds: the walk's and the WW - 1 before it
    __shared__ unsigned win[WW * NPL * RW * NLC];
    __shared__ int xch[8];
    FarmCtl *ctl = ctl_all + p;
    MegaState *mst = mst_all + p;
    const int tid = threadIdx.x, l = tid & 63;
    const bool w0 = tid < 64;
    if (__builtin_amdgcn_readfirstlane(ctl->fin)) return;
    const PairPlan pl = plans[p];
    const int m = pl.m, po = pl.src;
#define GNX_RFL(x_) __builtin_amdgcn_readfirstlane(x_)
    int wi = GNX_RFL(mst->wi), wj = GNX_RFL(mst->wj), wk = GNX_RFL(mst->wk), pend = GNX_RFL(mst->pend), wdone = 0;
    int cur_op = GNX_RFL(mst->cur_op), last_op = GNX_RFL(mst->last_op);
    int64_t li = farm_rfl64(mst->li), cnt = farm_rfl64(mst->cnt), cur_run = farm_rfl64(mst->cur_run);
    const int virt = GNX_RFL(mst->virt);
    const int64_t row_off = farm_rfl64(mst->row_off);
    const int64_t sbase = scr_off[p];
    const int n_list = GNX_RFL(ctl->n[par]);
    const int wi_in = wi, wj_in = wj;
    int2 mytile = make_int2(-1, -1), optile = make_int2(-1, -1); // lane l: tile l of the round's set / of the set being re-filled
    if (l < n_list) mytile = ctl->tile[par][l];
    if (PIPE && l < GNX_RFL(ctl->n[par ^ 1])) optile = ctl->tile[par ^ 1][l];
    auto flush_run = [&]() {
        if (cur_op >= 0) {
            if (tid == 0) { // {int64 run_length; uint8 op; 7 zero bytes}: one 16-byte store
                *reinterpret_cast<longlong2 *>(&scr[sbase + cnt]) = make_longlong2((long long)cur_run, (long long)(cur_op & 0xff));
            }
            cnt++;
        }
    };
    auto emit = [&](int op, int64_t run) {
        if (op == cur_op) cur_run += run;
        else { flush_run(); cur_op = op; cur_run = run; }
    };
    bool pexit = false;
    int hits = 0, q1n = 0, q1c = 0;
    while (true) {
        int slot = -1, s = 0, c = 0, lw = 0, lo = 0;
        if (w0) {
            if (wi == 0 || wj == 0) wdone = 1;
            if (!wdone && virt > 0 && wi <= virt) pexit = true;
            if (!wdone && !pexit) {
                s = (wi - 1) / HW;
                lw = (wi - 1 - s * HW) / RW;
                c = geo.block_of(wj + lw);
                const unsigned long long bal = __ballot(mytile.x == s && mytile.y == c);
                slot = bal ? (int)__builtin_ctzll(bal) : -1;
                lo = max(0, lw - (NLC - 1));
            }
            if (tid == 0) { xch[0] = slot; xch[1] = lo; xch[2] = lw; xch[3] = c; xch[4] = wj; }
        }
        __syncthreads();
        // (every thread of a wave reads the same word: made uniform, so that what the walk derives from them stays in SGPRs)
        slot = GNX_RFL(xch[0]); lo = GNX_RFL(xch[1]);
        const int hi = GNX_RFL(xch[2]), cT = GNX_RFL(xch[3]), wjT = GNX_RFL(xch[4]);
        if (slot < 0) break;
        const unsigned *src = planes + (((int64_t)p * 2 + par) * FARM_MAX + slot) * geo.tile_dw();
        // the window: lanes lo .. hi, words wlo .. whi of the tile (the walk only moves to earlier steps; the step after its cell: quirk Q1)
        const int tbX = geo.tbeg_of(cT), tmX = FarmGeo<AFF, RW>::tmin_of(cT);
        const int whi = min(geo.words() - 1, (wjT + hi - tbX) >> 4), wlo = max(0, whi - (WW - 1)); // (the walk's cell is at step t1 = wj + hi - 1 - tbX of the tile)
        { // the loads of the copy in flight, NIT per thread at a time, before their LDS stores: a window costs a trip or two to L2, not one per element.
          // Straight-line: an index beyond the window is clamped to its last element (loaded and stored again, the same value) and the lanes
          // lo + ll <= 63 beyond hi are copied with the rest -- a branch per element makes the compiler wait for every load where it stands
            constexpr int NIT = AFF ? 26 : 14;
            const int total = (whi - wlo + 1) * NPL * RW * NLC;
            const unsigned *srcw = src + wlo * (NPL * RW) * G64 + lo;
            for (int base = 0; base < total; base += NIT * 256) {
                unsigned tmp[NIT];
#pragma unroll
                for (int u = 0; u < NIT; u++) {
                    const int idx = min(base + tid + u * 256, total - 1);
                    const int row = idx / NLC, ll = idx - row * NLC;
                    tmp[u] = srcw[row * G64 + ll];
                }
#pragma unroll
                for (int u = 0; u < NIT; u++) win[min(base + tid + u * 256, total - 1)] = tmp[u];
            }
        }
        __syncthreads();
        if (w0) {
            hits++;
            int i = wi, j = wj, k = wk;
            if constexpr (AFF) {
                const int *hcX = reinterpret_cast<const int *>(src + geo.rows() * G64);
                if (pend) { const int kn = 3 - (GNX_RFL(hcX[i - 1 - s * HW]) & 3); if (pend == 2) { q1n++; q1c += (kn != k); } k = kn; pend = 0; }
                while (true) {
                    if (i == 0 || j == 0) { wdone = 1; break; }
                    int i0 = i - 1 - s * HW;
                    if (i0 < 0) break; // left the strip through its top edge
                    int l2 = i0 / RW, r2 = i0 - l2 * RW;
                    if (l2 < lo) break; // above the window: the next copy follows
                    int t1 = j + l2 - 1 - tbX;
                    if (t1 < tmX) break; // left the (usable part of the) tile through its skewed left edge
                    if ((t1 >> 4) < wlo) break; // left of the window's words (a long horizontal run): the next copy follows
                    unsigned w;
                    if (k == 0) { // state M: all 64 lanes read the M fields of the cells (i - x, j - x); the leading run that says "from M" is one step
                        const int ix = i0 - l;                                  // (no checkerboard edge inside it: x < li); the word of the cell the run
                        const int ixc = max(ix, 0);                             // ends at serves the scalar step that follows
                        const int l2x = ixc / RW, r2x = ixc - l2x * RW;
                        const int t1x = (j - l) + l2x - 1 - tbX;
                        const bool in = ix >= 0 && j - l >= 1 && t1x >= tmX && l2x >= lo && (t1x >> 4) >= wlo;
                        const unsigned wv = in ? win[((((t1x >> 4) - wlo) * 3 + 0) * RW + r2x) * NLC + (l2x - lo)] : 0u;
                        const bool ok = in && (int64_t)l < li && ((wv >> (2 * (t1x & 15))) & 3u) == 3u;
                        const unsigned long long nbal = ~__ballot(ok);
                        const int nb = nbal ? (int)__builtin_ctzll(nbal) : 64;
                        if (nb > 0) {
                            emit(0, nb); last_op = 0; li -= nb; i -= nb; j -= nb;
                            if (nb == 64 || !((__ballot(in) >> nb) & 1ull)) continue;
                            i0 = i - 1 - s * HW; l2 = i0 / RW; r2 = i0 - l2 * RW; t1 = j + l2 - 1 - tbX;
                        }
                        w = (unsigned)__builtin_amdgcn_readlane((int)wv, nb);
                    } else w = (unsigned)GNX_RFL((int)win[((((t1 >> 4) - wlo) * 3 + k) * RW + r2) * NLC + (l2 - lo)]);
                    const int pos = t1 & 15;
                    int tag = (int)((w >> (2 * pos)) & 3u);
                    if (tag == 0) { if (tid == 0) atomicOr(err, 2); wdone = 1; break; }
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
                            if (tag == 0) { if (tid == 0) atomicOr(err, 2); wdone = 1; break; }
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
                            const int t3 = (j + 1) + l2 - 1 - tbX;
                            const unsigned w3 = (unsigned)GNX_RFL((int)win[((((t3 >> 4) - wlo) * 3 + 0) * RW + r2) * NLC + (l2 - lo)]);
                            k = 3 - (int)((w3 >> (2 * (t3 & 15))) & 3u);
                        } else if (i - 1 - s * HW >= 0 && m + (i - 1 - s * HW) / RW - 1 - tbX >= tmX) k = 3 - (GNX_RFL(hcX[i - 1 - s * HW]) & 3);
                        else pend = 2; // row i belongs to the strip above, or its lane passed column m before this tile began: the next tile has it
                        if (pend != 2) { q1n++; q1c += (k != kt); }
                    }
                }
            } else {
                while (true) {
                    if (i == 0 || j == 0) { wdone = 1; break; }
                    int i0 = i - 1 - s * HW;
                    if (i0 < 0) break; // left the strip through its top edge
                    int l2 = i0 / RW, r2 = i0 - l2 * RW;
                    if (l2 < lo) break; // above the window
                    int t1 = j + l2 - 1 - tbX;
                    if (t1 < 0) break; // left the tile through its (skewed) left edge
                    unsigned w;
                    { // all 64 lanes read the fields of the cells (i - x, j - x): the leading run of diagonal fields is one step, the word of the cell it ends at serves the scalar step
                        const int ix = i0 - l;
                        const int ixc = max(ix, 0);
                        const int l2x = ixc / RW, r2x = ixc - l2x * RW;
                        const int t1x = (j - l) + l2x - 1 - tbX;
                        const bool in = ix >= 0 && j - l >= 1 && t1x >= 0 && l2x >= lo;
                        const unsigned wv = in ? win[((t1x >> 4) * RW + r2x) * NLC + (l2x - lo)] : 0u;
                        const bool ok = in && ((wv >> (2 * (t1x & 15))) & 3u) == 3u;
                        const unsigned long long nbal = ~__ballot(ok);
                        const int nb = nbal ? (int)__builtin_ctzll(nbal) : 64;
                        if (nb > 0) {
                            emit(0, nb); last_op = 0; i -= nb; j -= nb;
                            if (nb == 64 || !((__ballot(in) >> nb) & 1ull)) continue;
                            i0 = i - 1 - s * HW; l2 = i0 / RW; r2 = i0 - l2 * RW; t1 = j + l2 - 1 - tbX;
                        }
                        w = (unsigned)__builtin_amdgcn_readlane((int)wv, nb);
                    }
                    const int pos = t1 & 15;
                    const int tag = (int)((w >> (2 * pos)) & 3u);
                    if (tag == 0) { if (tid == 0) atomicOr(err, 2); wdone = 1; break; }
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
                            if (((w >> (2 * pnz)) & 3u) == 0) { if (tid == 0) atomicOr(err, 2); wdone = 1; break; }
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
            }
            wi = i; wj = j; wk = k;
        }
    }
    if (!w0) return;
    if (tid == 0) {
        mst->wi = wi; mst->wj = wj; mst->wk = wk; mst->pend = pend; mst->li = li; mst->cnt = cnt; mst->cur_run = cur_run; mst->cur_op = cur_op; mst->last_op = last_op;
        ctl->rounds += 1; ctl->hits += hits;
        q1_report(q1n, q1c);
    }
    if (wdone || pexit) {
        if (tid == 0) { mst->done = pexit ? 0 : 1; ctl->fin = 1; ctl->n[0] = 0; ctl->n[1] = 0; }
        if (!pexit) {
            // Step 4 (affineGap.go:135-139 / constGap.go:59-63) -- quirk Q2 when the corner is not the origin
            const int64_t gi = (int64_t)wi + (wi > 0 ? row_off : 0);
            const bool up_exit = (last_op != 1) && (gi % tp.ci == 0);
            const bool left_exit = (last_op != 2) && ((int64_t)wj % tp.cj == 0);
            if (!up_exit && left_exit) emit(2, gi);
            else if (up_exit && !left_exit) emit(1, wj);
            flush_run();
            if (tid == 0) { nops[po] = cnt; score_out[po] = hfin[pl.hcol_off]; }
        }
    } else {
        const int da = GNX_RFL(ctl->acc_i) / 2 + (wi_in - wi), db = GNX_RFL(ctl->acc_j) / 2 + (wj_in - wj);
        const int n = farm_predict<AFF, RW>(geo, ctl->tile[par], wi, wj, virt, nt, tid == 0, da, db,
                                        [&](int s, int c) { return PIPE && __ballot(optile.x == s && optile.y == c) != 0ull; });
        if (tid == 0) { ctl->n[par] = n; ctl->acc_i = da; ctl->acc_j = db; }
    }
#undef GNX_RFL
}

// plain rounds: launch {fill set 0, walk set 0} a round
template <int RW, bool P16>
__global__ __launch_bounds__(64) void al64_farm_fill_kernel(const PairPlan *__restrict__ plans, const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                            const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start, KParams kp,
                                                            const int2 *__restrict__ rowbuf, const int *__restrict__ snap, int *__restrict__ err,
                                                            const long long *__restrict__ bases, const FarmCtl *__restrict__ ctl, unsigned *__restrict__ planes, int par) {
    al64_farm_fill_body<RW, P16>(plans, a_buf, a_start, b_buf, b_start, kp, rowbuf, snap, err, bases, ctl, planes, blockIdx.y, blockIdx.x, par);
}
template <int RW, bool P16>
__global__ __launch_bounds__(64) void cl64_farm_fill_kernel(const PairPlan *__restrict__ plans, const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                            const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start, KParams kp,
                                                            const int *__restrict__ rowbuf, const int *__restrict__ snap, int *__restrict__ err,
                                                            const long long *__restrict__ bases, const FarmCtl *__restrict__ ctl, unsigned *__restrict__ planes, int par) {
    cl64_farm_fill_body<RW, P16>(plans, a_buf, a_start, b_buf, b_start, kp, rowbuf, snap, err, bases, ctl, planes, blockIdx.y, blockIdx.x, par);
}
template <bool AFF, int RW>
__global__ __launch_bounds__(256) void farm_walk_kernel(const PairPlan *__restrict__ plans, TbParams tp,
                                                        const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                        const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err,
                                                        MegaState *__restrict__ mst_all, FarmCtl *__restrict__ ctl_all,
                                                        const unsigned *__restrict__ planes, int nt, int ckr) {
    farm_walk_body<AFF, RW, false>(plans, tp, hfin, score_out, nops, scr_off, scr, err, mst_all, ctl_all, planes, nt, ckr, blockIdx.x, 0);
}

// overlapped rounds: ONE launch a round -- workgroup 0 of a pair walks set `par` (re-filled by the launch before), workgroups 1 .. nt re-fill
// the other set (asked for by the walk of the launch before) meanwhile, on their own CUs; the walk then asks for the set after that one.
// The walk's own tile is in one of the two sets at the latest two rounds after a wrong guess (the first tile of a line that starts at the
// walk's cell is either being re-filled or asked for), so the rounds still move.
template <int RW, bool P16>
__global__ __launch_bounds__(256) void al64_farm_round_kernel(const PairPlan *__restrict__ plans, const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                              const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start, KParams kp, TbParams tp,
                                                              const int2 *__restrict__ rowbuf, const int *__restrict__ snap, const int64_t *__restrict__ hfin,
                                                              int64_t *__restrict__ score_out, int64_t *__restrict__ nops, const int64_t *__restrict__ scr_off,
                                                              gnx_cigar *__restrict__ scr, int *__restrict__ err, const long long *__restrict__ bases,
                                                              MegaState *__restrict__ mst_all, FarmCtl *__restrict__ ctl_all, unsigned *__restrict__ planes, int nt, int par) {
    if (blockIdx.x == 0) farm_walk_body<true, RW, true>(plans, tp, hfin, score_out, nops, scr_off, scr, err, mst_all, ctl_all, planes, nt, kp.ckc, blockIdx.y, par);
    else if (threadIdx.x < 64) al64_farm_fill_body<RW, P16>(plans, a_buf, a_start, b_buf, b_start, kp, rowbuf, snap, err, bases, ctl_all, planes, blockIdx.y, (int)blockIdx.x - 1, par ^ 1);
}
template <int RW, bool P16>
__global__ __launch_bounds__(256) void cl64_farm_round_kernel(const PairPlan *__restrict__ plans, const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                              const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start, KParams kp, TbParams tp,
                                                              const int *__restrict__ rowbuf, const int *__restrict__ snap, const int64_t *__restrict__ hfin,
                                                              int64_t *__restrict__ score_out, int64_t *__restrict__ nops, const int64_t *__restrict__ scr_off,
                                                              gnx_cigar *__restrict__ scr, int *__restrict__ err, const long long *__restrict__ bases,
                                                              MegaState *__restrict__ mst_all, FarmCtl *__restrict__ ctl_all, unsigned *__restrict__ planes, int nt, int par) {
    if (blockIdx.x == 0) farm_walk_body<false, RW, true>(plans, tp, hfin, score_out, nops, scr_off, scr, err, mst_all, ctl_all, planes, nt, 0, blockIdx.y, par);
    else if (threadIdx.x < 64) cl64_farm_fill_body<RW, P16>(plans, a_buf, a_start, b_buf, b_start, kp, rowbuf, snap, err, bases, ctl_all, planes, blockIdx.y, (int)blockIdx.x - 1, par ^ 1);
}

} // namespace
