// const_long_wg.hip.h -- the piped constant-gap sweep of config C5 with SEVERAL STRIPS PER WORKGROUP: waves hand rows over through LDS
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.6.
#pragma once
#include "const_long.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// cl_sweep_kernel (const_long.hip.h) runs every 160-row strip of a group of 4 pairs as a workgroup of its own: a 20 kb read is a chain of
// 125 workgroups that hand their bottom rows over through HBM (write-through stores, a progress word, polls).  Round 3's counters: 55 % of
// the wave-cycles of that launch are waits, an ablation without them is 25 % faster, and 143 of the 463 VALU instructions of a block
// of 16 steps are hand-over plumbing (boundary queue, base queue, bottom-row collection: DPP moves and the copies they need).
// Here a workgroup is NW waves = NW CONSECUTIVE strips of the group (an "item"), one wave per strip:
//   * wave w passes the bottom row of its strip to wave w + 1 through a ring in LDS, one value per step: the last lane of a pair
//     writes v(row 160 (s+1), column t - 15) at step t, the first lane of the consumer reads v(.., column t') one step before it needs
//     it.  No DPP queue, no rotate, no collection: the step is  1 DPP move + 10 x (add, max3) + the LDS address add.
//   * the bases of the columns travel through a ring in LDS as well (one ds_read_u16 per step instead of two DPP moves), as in fp_sweep.
//   * flags in LDS (blocks produced / blocks consumed per ring) pace the waves: the consumer of a ring stays two blocks (32 steps) behind
//     its producer -- the skew of the 16 lanes plus one block -- and the producer at most NSLOT blocks ahead of what has been read.
//     A poll costs an LDS round trip (~100 cycles) instead of an uncached memory round trip (~2 us) plus s_sleep.
//   * only the item boundary goes through memory, exactly as before (write-through row stores, progress word every rb_pub steps, claims
//     for forward progress -- an item now is NW strips): 25 memory hand-overs per 20 kb read instead of 125.
//   * every strip still stores its bottom row (plain stores for the inner strips): cl_walk_kernel re-fills tiles of any strip from them.
// Outputs are bit-identical to cl_sweep_kernel's (row buffer, snapshots, final values): the walk does not know which sweep ran.
//
// Ring geometry.  A block is 16 steps t0 + 1 .. t0 + 16 (t0 = 16 k).  Position p of a hand-over ring holds the column p - 14: the producer's
// last lane writes position t - 1 at step t (its column is t - 15), so block k of the producer fills exactly the 16 aligned positions
// [t0, t0 + 16) -- immediate offsets, no wrap inside a block.  The consumer's first lane needs column t' at step t' = position t' + 14; it
// reads one step ahead, so during its block k it reads positions [t0 + 16, t0 + 32) = the producer's block k + 1.  Hence: the consumer
// starts block k when the producer has finished block k + 1; the producer starts block k when the consumer has finished block
// k - NSLOT (whose reads were of the slot block k goes into).
// NW = 4: one wave of every workgroup per SIMD, so that workgroups pack without fragmenting the SIMDs' wave slots (NW = 5 -- 125 strips =
// 25 items -- measured 2.1 waves per SIMD resident instead of 5: the fifth wave of a workgroup needs a free slot on one particular SIMD).
// LDS per workgroup: score table 128 B + 4 profiles x 6400 B + 5 rings x 1024 B + 4 base rings x 256 B + flags = 31 928 B = 25 granules
// of 1280 B: 5 workgroups = 20 waves per CU, what the one-strip kernel's registers allowed (5 per SIMD at <= 96 registers).
// ------------------------------------------------------------------------------------------------------
// Round 6 experiment: the ITEM boundary -- the one hand-over through memory, every NW strips -- without a progress word.  Shipped form: the item's last wave stored its bottom row
// with write-through stores, waited for their acknowledgement (s_waitcnt vmcnt(0)) and advanced a progress word every rb_pub steps; the next item's first wave polled
// that word with s_sleep(32) between polls.  The four waves of an item are rate-coupled through their LDS rings, so the publishing wave's memory round trips paced
// all of them -- that was the hypothesis (SQ_WAIT_ANY 0.31 of the wave-cycles at 4.4 waves per SIMD, profiles/r5_hbm_traffic.json).  In the experiment the boundary rows start as CLW_SENT (clw_fill_sentinel_kernel,
// before the launch), the producer just stores, the consumer re-loads an entry that still reads as the fill (lat_fill_kernel's protocol): a 4-byte store is never torn,
// nobody relies on the order of two stores.  MEASURED AND NOT SHIPPED (-DGNX_CLW_SENT=1 builds it): 1 024 pairs of C5 sweep in 154.1 ms with it (150 ms + 4 ms of fill) against 153.6 ms with the
// progress words -- with 4.4 waves per SIMD the publishing wave's round trips were already hidden; the waits of this kernel are the LDS flag polls between the waves of an item
// (profiles/r6_experiments.md section 7).
#ifndef GNX_CLW_SENT
#define GNX_CLW_SENT 0
#endif
constexpr int CLW_SENT = (int)0x80808080;  // "not written yet": below every key of the static range (> -2^29), not the "-inf" NEG4 either
// the bottom rows of the strips that END an item (strip NW k + NW - 1 of every pair, when another strip follows): grid (items, pairs)
__global__ __launch_bounds__(256) void clw_fill_sentinel_kernel(const PairPlan *__restrict__ plans, int n_pairs, int nw, int *__restrict__ rowbuf) {
    const int p = blockIdx.y;
    if (p >= n_pairs) return;
    const PairPlan pl = plans[p];
    const int s = (int)blockIdx.x * nw + nw - 1;
    if (s + 1 >= pl.strips) return; // (the last strip of a pair hands nothing down)
    int4 *row = reinterpret_cast<int4 *>(rowbuf + pl.rowbuf_off + (int64_t)s * ((int64_t)pl.m + 1));
    // (whole int4s where the row is aligned; the ragged ends as ints)
    int *ri = rowbuf + pl.rowbuf_off + (int64_t)s * ((int64_t)pl.m + 1);
    const int n = pl.m + 1;
    const int head = (int)((4 - ((reinterpret_cast<uintptr_t>(ri) >> 2) & 3)) & 3);
    for (int x = threadIdx.x; x < min(head, n); x += 256) ri[x] = CLW_SENT;
    const int nq = max(n - head, 0) >> 2;
    int4 *rq = reinterpret_cast<int4 *>(ri + head);
    for (int x = threadIdx.x; x < nq; x += 256) rq[x] = make_int4(CLW_SENT, CLW_SENT, CLW_SENT, CLW_SENT);
    for (int x = head + 4 * nq + threadIdx.x; x < n; x += 256) ri[x] = CLW_SENT;
    (void)row;
}

constexpr int CLW_RC = 64;                 // columns (positions) per hand-over ring and pair
constexpr int CLW_NSLOT = CLW_RC / 16;     // blocks per ring

template <int NW>
struct ClwLds {
    using PC = ProfCfg<true>;
    static constexpr int PROF = PC::TOTAL - 0;                 // dwords of one wave's profile (4 pairs)
    static constexpr int OFF_PROF = 32;
    static constexpr int OFF_RING = OFF_PROF + NW * PROF;      // NW + 1 rings of 4 x CLW_RC dwords: ring r = input of wave r = output of wave r - 1
    static constexpr int OFF_BRING = OFF_RING + (NW + 1) * 4 * CLW_RC; // per wave, per pair: 64 bytes (the bases 0 .. 4 of 32 columns, each stored twice 32 apart)
    static constexpr int OFF_FLAG = OFF_BRING + NW * 4 * 16;   // prod[r], r = 0 .. NW; cons[r]; claim word; pad
    static constexpr int TOTAL = OFF_FLAG + 2 * (NW + 1) + 4;
};

// spin on an LDS word until it reaches `need` (another wave of this workgroup advances it); a 2 s bound turns a bug into error flag 16
__device__ __forceinline__ void clw_wait_ge(int *flag, int need, int *err) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) return;
    const long long t_begin = wall_clock64();
    int it = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
        __builtin_amdgcn_s_sleep(2);
        if ((++it & 1023) == 0 && wall_clock64() - t_begin > 200000000LL) { atomicOr(err, 16); break; }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(5, 5))) void cl_sweep_wg_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                              const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                              const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                              KParams kp, int *__restrict__ rowbuf, int *__restrict__ snap, int64_t *__restrict__ hfin,
                                                              int *__restrict__ err, const int2 *__restrict__ item_map, int *__restrict__ item_prog) {
    using PC = ProfCfg<true>;
    using LY = ClwLds<NW>;
    constexpr int LW = PC::LW, BST = PC::BST;
    constexpr int RC = CLW_RC, NSLOT = CLW_NSLOT;
    __shared__ int lds[LY::TOTAL];
    const int tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, l = lane & 15;
    if (tid < 25) lds[tid] = kp.sc4[tid] - 2 * kp.g4; // rebased diagonal move: 4*(s - 2g); every value carries tag 2
    // claims (claim_items, gnx_common.hip.h): an item = NW strips of a group; wave 0 claims, the workgroup follows
    int *claim_word = &lds[LY::OFF_FLAG + 2 * (NW + 1)];
    const int gq = item_map[blockIdx.x].x, it_own = item_map[blockIdx.x].y;
    if (w == 0) {
        const int n = claim_items(item_prog + gridDim.x, 1, it_own);
        if (lane == 0) *claim_word = n;
    }
    __syncthreads();
    const int n_stolen = *claim_word;
    if (n_stolen < 0) return;

    const int pbase = gq * 4;
    int S_max = 0, m_max = 0;
    for (int q = 0; q < 4; q++) {
        if (pbase + q < n_pairs) { S_max = max(S_max, plans[pbase + q].strips); m_max = max(m_max, plans[pbase + q].m); }
    }
    const int p = pbase + g;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const uint8_t *ap = a_buf + (valid ? a_start[p] : 0);
    BetaBytes bp;
    bp.init(b_buf, kp, valid ? b_start[p] : 0, valid ? pl.m : 0);
    const int Tend = (m_max + 15 + 15) & ~15;
    const int K = Tend >> 4; // blocks
    const int64_t rb_pitch = (int64_t)pl.m + 1;
    int bad = 0;

    int *prof = &lds[LY::OFF_PROF + w * LY::PROF + PC::pair_off(g)];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    int *rin = &lds[LY::OFF_RING + w * 4 * RC + g * RC];        // what the strip above hands to this wave
    int *rout = &lds[LY::OFF_RING + (w + 1) * 4 * RC + g * RC]; // what this wave hands down
    // Bank conflicts: an LDS access serves 32 lanes = two pairs per cycle, and both pairs touch the same position of their rings at the
    // same step -- regions 64 (hand-over) / 32 (bases) dwords apart are the same banks.  So the odd pair of a duo keeps its ring ROTATED:
    // position p lives at (p + 16) mod RC, column c of the base ring at (c + 16) mod 32 -- 16 resp. 8 banks away from its neighbour.
    const int rot = (g & 1) * 16;
    unsigned char *bring = reinterpret_cast<unsigned char *>(&lds[LY::OFF_BRING + w * 4 * 16 + g * 16]);
    int *prod_in = &lds[LY::OFF_FLAG + w], *prod_out = &lds[LY::OFF_FLAG + w + 1];
    int *cons_in = &lds[LY::OFF_FLAG + (NW + 1) + w], *cons_out = &lds[LY::OFF_FLAG + (NW + 1) + w + 1];

    for (int it = it_own - n_stolen; it <= it_own; it++) {
        const int bid = (int)blockIdx.x - it_own + it; // block index of item `it` of this group = its slot in item_prog
        __syncthreads(); // every wave is done with the item before (and with the score table / the claim word)
        if (tid < 2 * (NW + 1)) lds[LY::OFF_FLAG + tid] = 0;
        __syncthreads();
        const int s = it * NW + w;
        if (s >= S_max) continue; // (an item's last waves may have no strip; they still take part in the barriers above)
        const bool gact = valid && s < pl.strips;
        const int m_eff = gact ? pl.m : 0;
        int m_min = 0x7fffffff;
        for (int q = 0; q < 4; q++) m_min = min(m_min, (pbase + q < n_pairs && s < plans[pbase + q].strips) ? plans[pbase + q].m : 0);
        const bool store_row = gact && (s + 1 < pl.strips);
        const bool has_cons = (w + 1 < NW) && (s + 1 < S_max); // the strip below runs in this workgroup: hand over through LDS
        const bool to_mem = !has_cons;                         // the strip below is another workgroup's: write-through stores + progress word
        const bool from_mem = (w == 0) && s > 0;               // the strip above was another workgroup's
        const int row0 = s * H + l * R;
        int val[R];
        {
            int a5[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i0 = row0 + r;
                int a = 0;
                if (gact && i0 < pl.n) { a = ap[i0]; if (a >= 5) { bad = 1; a = 4; } }
                a5[r] = a * 5;
            }
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = (lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) val[r] = 2; // column 0, rebased: 0 (tag 2)
        int diag0 = 2, v_out = 0;
        auto base_raw = [&](int c) { return (c >= 1 && c <= m_eff) ? bp.raw(c - 1) : 0; };
        auto base_off = [&](int raw, int c) { int b = (c >= 1 && c <= m_eff) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b; }; // the base 0 .. 4 (its profile plane is b * BST dwords into the profile: one v_mad in fetch)
        int rb_seen = 0;
        auto wait_rows = [&](int cmax) { // columns <= cmax of the row above are in memory (the item before publishes them)
            if (from_mem && rb_seen < cmax) {
                const long long t_begin = wall_clock64();
                while ((rb_seen = rb_progress(&item_prog[bid - 1])) < cmax) {
                    __builtin_amdgcn_s_sleep(32);
                    if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); break; }
                }
            }
        };
        auto row_above = [&](int c) { return (c >= 1 && c <= m_eff) ? rb_load32(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], true) : 0; };
        // (GNX_CLW_SENT) ... until the item above has stored it: an entry that still reads as the launch's fill is loaded again
        auto settle_above = [&](int c, int &v) {
            if (!GNX_CLW_SENT || !from_mem) return;
            const bool mine = c >= 1 && c <= m_eff;
            if (__any(mine && v == CLW_SENT)) {
                const long long t_begin = wall_clock64();
                while (true) {
                    if (mine && v == CLW_SENT) v = rb_load32(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], true);
                    if (!__any(mine && v == CLW_SENT)) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); if (mine && v == CLW_SENT) v = 0; break; }
                }
            }
        };

        // ---- prologue: the input ring's blocks 0 and 1 (columns <= 17), the base ring's columns 1 .. 16 ----
        if (w == 0) {
            if (s == 0) { // row 0, rebased: the constant 2 in every position, once
#pragma unroll
                for (int x = 0; x < NSLOT; x++) rin[l + 16 * x] = 2; // (every position: the rotation does not matter)
            } else {
                if (!GNX_CLW_SENT) wait_rows(18);
                int v0 = row_above(l - 14), v1 = row_above(l + 2);
                settle_above(l - 14, v0); settle_above(l + 2, v1);
                rin[(l + rot) & (RC - 1)] = v0;          // block 0: only position 15 (column 1) is ever read
                rin[(16 + l + rot) & (RC - 1)] = v1;     // block 1: columns 2 .. 17
            }
        } else {
            clw_wait_ge(prod_in, 2, err);
        }
        {
            const int o = base_off(base_raw(l + 1), l + 1);
            const int x = (l + 1 + rot) & 31;
            bring[x] = (unsigned char)o; bring[x + 32] = (unsigned char)o;
            const int z = (l + 17 + rot) & 31; // columns -15 .. 0: never used by a live cell
            bring[z] = 0; bring[z + 32] = 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int up_lds = rin[(15 + rot) & (RC - 1)];     // what the first lane needs at step 1: column 1
        int pb_cur = (int)bring[(1 - l + rot) & 31]; // LDS offset of the base of this lane's column at step 1
        int nraw = 0, nv = 0;

        auto fetch = [&](int pbv, int *wv) {
            const int *pw = reinterpret_cast<const int *>(prof_lane + pbv * (BST * 4));
#pragma unroll
            for (int k = 0; k < LW; k++) wv[k] = pw[k];
        };
        // one step: t = t0 + 1 + u.  rin_n = input ring positions of the NEXT block, rout_b = output ring positions of this block,
        // br = this lane's base-ring entries of the block (entry u + 1 = the step after this one)
        auto step = [&](const int t, auto chk, const int u, const int *rin_n, int *rout_b, const unsigned char *br) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_v = dpp_shr1(up_lds, v_out); // first lane of a pair: the ring's value (no source: keeps `old`); others: the previous lane's bottom row
            int wv[LW];
            fetch(pb_cur, wv);
            const int up_next = rin_n[u];       // column t + 1 of the row above (used by the first lane at the next step)
            const int pb_next = (int)br[u];     // base of this lane's column at the next step
            const int j = t - l;
            if (!CHECK || (j >= 1 && j <= m_eff)) {
                int vd = diag0, vu = up_v;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int S4 = (r & 1) ? (wv[r >> 1] >> 16) : (int)(short)(wv[r >> 1] & 0xffff);
                    const int k = max3i(vd + S4, val[r], vu);
                    vd = val[r];
                    val[r] = k;
                    vu = k;
                }
                diag0 = up_v;
                v_out = vu;
            }
            if (l == G - 1) rout_b[u] = v_out;  // the pair's last lane: bottom row of the strip, column t - 15 (position t - 1)
            up_lds = up_next;
            pb_cur = pb_next;
        };

        for (int k = 0; k < K; k++) {
            const int t0 = k << 4;
            if (t0 > 0 && t0 % kp.ckc == 0 && gact && t0 <= m_eff + 15) { // snapshot: the state the wave resumes from at step t0
                uint4 *dst = reinterpret_cast<uint4 *>(snap + pl.ckpt_off + (((int64_t)(t0 / kp.ckc - 1) * pl.strips + s) * G + l) * SNAPW);
                dst[0] = make_uint4((unsigned)val[0], (unsigned)val[1], (unsigned)val[2], (unsigned)val[3]);
                dst[1] = make_uint4((unsigned)val[4], (unsigned)val[5], (unsigned)val[6], (unsigned)val[7]);
                dst[2] = make_uint4((unsigned)val[8], (unsigned)val[9], (unsigned)diag0, 0u);
            }
            // pacing (see the header): the input ring's block k + 1 must be complete, the output ring's slot of block k free
            if (w > 0 && k + 1 < K) clw_wait_ge(prod_in, k + 2, err);
            if (has_cons && k > NSLOT) clw_wait_ge(cons_out, k - NSLOT, err); // (the slot's old content, ring block k - NSLOT, was read during the consumer's block k - NSLOT - 1)
            if (from_mem) { if (!GNX_CLW_SENT) wait_rows(t0 + 33); nv = row_above(t0 + 18 + l); } // input ring block k + 2 (columns t0 + 18 .. t0 + 33), written at the end of this block
            nraw = base_raw(t0 + 17 + l);                                        // bases of the next block's new columns
            asm volatile("" ::: "memory");
            const int *rin_n = rin + ((t0 + 16 + rot) & (RC - 1));
            int *rout_b = rout + ((t0 + rot) & (RC - 1));
            const unsigned char *br = bring + ((t0 + 2 - l + rot) & 31);
            if (t0 >= 16 && t0 + 16 <= m_min) {
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    if (u == 15) { const int o = base_off(nraw, t0 + 17 + l); const int x = (t0 + 17 + l + rot) & 31; bring[x] = (unsigned char)o; bring[x + 32] = (unsigned char)o; }
                    step(t0 + u + 1, std::false_type{}, u, rin_n, rout_b, br);
                }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) {
                    if (u == 15) { const int o = base_off(nraw, t0 + 17 + l); const int x = (t0 + 17 + l + rot) & 31; bring[x] = (unsigned char)o; bring[x + 32] = (unsigned char)o; }
                    step(t0 + u + 1, std::true_type{}, u, rin_n, rout_b, br);
                }
            }
            asm volatile("" ::: "memory");
            if (from_mem) { settle_above(t0 + 18 + l, nv); rin[(t0 + 32 + l + rot) & (RC - 1)] = nv; }
            // the block's 16 bottom-row values, one per lane: column t0 + l - 14 (position t0 + l)
            const int dv = rout_b[l];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's ring writes are done (and dv is here)
            if (lane == 0) {
                if (has_cons) __hip_atomic_store(prod_out, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (w > 0) __hip_atomic_store(cons_in, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (store_row) {
                const int c = t0 + l - 14;
                if (c >= 1 && c <= m_eff) rb_store32(&rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + c], dv, to_mem);
            }
            if (!GNX_CLW_SENT && to_mem && ((t0 + 16) & (kp.rb_pub - 1)) == 0) rb_publish(&item_prog[bid], t0 + 1, lane);
        }
        if (has_cons && lane == 0) __hip_atomic_store(prod_out, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (gact && m_eff >= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) if (row0 + r + 1 == pl.n) hfin[pl.hcol_off] = (int64_t)((val[r] >> 2) + (kp.g4 >> 2) * (pl.n + m_eff)); // plain V(n, m) (this kernel only runs pairs inside the static int32 range)
        }
        if (!GNX_CLW_SENT && to_mem) rb_publish(&item_prog[bid], 0x7fffffff, lane);
    }
    if (bad) atomicOr(err, 1);
}

} // namespace
