// const_long.hip.h -- constant-gap alignments WITHOUT a stored direction matrix: score-only sweep with snapshots + fused re-fill / walk
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.6.
#pragma once
#include "fill_affine.hip.h" // ProfCfg
#include "traceback.hip.h"   // TbParams

namespace {
// ------------------------------------------------------------------------------------------------------
// Long constant-gap pairs (config C5: 20 kb reads x 100 kb windows, align/constGap.go:13-68 with 10 000 x 10 000 checkerboards,
// and ConstGap_highMem): a stored 2-bit direction matrix is 0.5 GB per pair.  Like the reference -- which keeps one row / column per
// checkerboard and re-computes the checkerboards its path crosses (constGap.go:129-222) -- this path stores only what is needed to
// re-compute, but at the granularity of the GPU mapping instead of 10 000 x 10 000 tiles:
//
//   cl_sweep_kernel   forward pass, SCORE ONLY.  Same wavefront mapping as fill_const_kernel (16 lanes x 10 rows per pair, 160-row
//                     strips, pipelined strips for small launches) on rebased values V' = V - g*(i+j): per cell  add, max3  -- no
//                     tags, no direction bits (2 VALU instructions instead of 5).  It keeps
//                       * the bottom row of every strip (the row buffer the strips hand over anyway), 4 B per column, and
//                       * a SNAPSHOT of the wavefront every CKC steps: the 10 row values + the diagonal value of every lane, i.e. the
//                         complete state from which the wave can resume at step c*CKC (wave-uniform, coalesced 768 B per pair).
//   cl_walk_kernel    traceback, one wave per 4 pairs, everything on the device: for the strip s and snapshot interval c the walk is
//                     in, RE-FILL the <= CKC steps from the snapshot with the recording recurrence of fill_const_kernel into a
//                     direction-bit tile in LDS (18 KB per pair), walk inside the tile with LDS latency instead of HBM latency, move
//                     on to the next tile.  The tiles a path crosses are ~1 % of the matrix.
//
// Values and argmax tags of a re-filled tile are those of the full fill (same recurrence from exact state), so the walk sees the
// bits fill_const_kernel would have stored; the walk itself (run merging, Step 4 with quirk Q2) is traceback_kernel<false>'s.
// Memory per 20 kb x 100 kb pair: row buffer 50 MB + snapshots 19 MB + run staging 2 MB instead of 500 MB.
// ------------------------------------------------------------------------------------------------------
// Snapshot spacing in wavefront steps (multiple of 16), chosen per call (KParams::ckc; the walk kernel is compiled for both).  A small tile
// means less to re-fill before the walk can look at its cell (on average half a tile) but more tiles per path, each with the fixed cost
// of a round (profile, snapshot, boundary): while the walk is a latency chain -- fewer workgroups than the GPU holds two per SIMD -- 224
// wins (ConstGap 250 .. 3200 x 10 000: walk stage -8 .. -27 %; 1024 pairs of C5: 41.4 -> 38.1 ms); with 2048 pairs of C5 resident the
// walk is bound by its instruction count and 448 wins (42 against 50 ms).
constexpr int CKC = 448, CKC_SMALL = 224;
constexpr int SNAPW = 12;                         // dwords per lane per snapshot: val[R], diag0, pad

__device__ __forceinline__ void rb_store32(int *p, int v, bool piped) {
    if (piped && !GNX_RB_FENCE) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ int rb_load32(const int *p, bool piped) {
    if (piped && !GNX_RB_FENCE) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// ------------------------------------------------------------------------------------------------------
// REBASE: alignments of ANY length on int32 keys (round 5; align/align.go:8 -- the reference is int64 end to end, and its low-memory
// checkerboard, affineGap.go:59-68 / constGap.go:13-68, exists for sequences far beyond the int32 range of 4*score).
// The statically rebased values V' = V - g*(i+j) still grow along the diagonal, by up to (s_max - 2g) per step: min(n, m) bases long
// they leave int32.  But every max of the recurrence compares candidates of ONE cell, and the cells a wave holds at any moment -- 160
// rows x 16 columns of an anti-diagonal band -- differ by a bounded amount (neighbouring cells of a global alignment by at most one
// substitution score + two gaps).  So a strip keeps its keys relative to a BASE of its own that it moves along: every K steps (K = the
// snapshot spacing, the rebase comes right before the snapshot) it subtracts d = the key of its first row's current cell (a multiple
// of 4: tags stay) from everything it holds and adds d to its base, an int64 it also leaves in memory: bases[strip][t0 / K].
//   * the bottom row a strip hands down carries, per 16-column block, the base that was in force when the block was written (block of
//     column c: (c + 14) / K); the strip below converts what it loads with the difference of the two bases -- an int32, the rows are neighbours;
//   * a snapshot is relative to bases[strip][its index]; the walk's re-fill converts the boundary row the same way and never rebases
//     inside its <= K steps; directions depend on differences inside a cell only, so the tile's bits are the plain recurrence's;
//   * the score leaves as base + key, in int64.
// The REBASE instantiations are chosen by the host for pairs beyond the static range (and by GNX_REBASE=1, for the tests: then every
// pair of the snapshot path with more than K columns rebases); everybody else runs the code without it, instruction for instruction.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rbase_store(long long *p, long long v, bool piped) {
    if (piped) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ long long rbase_load(const long long *p, bool piped) {
    if (piped) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
// a key that is a constant of the recurrence (row 0) relative to a base: far below everything the strip holds once the base has moved
// -- kept above the sentinel so that it can neither wrap nor win
__device__ __forceinline__ int rbase_const(long long key, long long base) {
    const long long v = key - base;
    return (int)(v < (long long)(NEG4 + 8) ? (long long)(NEG4 + 8) + (key & 3) : v);
}
__device__ __forceinline__ int rbase_delta(long long theirs, long long mine) {
    const long long v = theirs - mine;
    return (int)(v > (1LL << 30) ? (1LL << 30) : (v < -(1LL << 30) ? -(1LL << 30) : v));
}

// PairPlan fields used here: n, m, strips, rowbuf_off (ints), ckpt_off (ints: snapshots [c-1][strip][lane][SNAPW]), hcol_off (slot of
// the final value), src (output slot).
// PIPED = the strips of a pair run as separate, pipelined workgroups (strip_map); else one wave walks the strips of its 4 pairs in turn.
// The two forms differ in what bounds them.  Un-piped (big batches), the wave is alone with its arithmetic: the profile entries of a
// step are read from LDS ONE STEP AHEAD (the base a lane needs at step t + 1 is the one its left neighbour has at step t, so the DPP
// move and the reads for t + 1 go out before the arithmetic of step t), and a base loaded for the next block is turned into its LDS
// offset only where the queue is needed -- checked on the spot it costs a full memory round trip per block (the compiler's
// s_waitcnt vmcnt(0) sits right behind the global_load_ubyte): 320 x 10 000 x 32 768 pairs 10.2 -> 9.0 ms, 1.16e13 cells/s.
// Piped, a wave spends two thirds of its time parked behind the hand-over of the strip above it (uncached loads, spin waits) and
// what counts is how many waves a SIMD holds: the ten registers of the look-ahead cost the fifth wave, and every variant that made
// the single wave stall less (look-ahead, deferred conversion, lagged publishes, loads two blocks ahead with hand-kept counts) made the
// launch SLOWER -- 451 -> 492 -> 522 ms for 2048 pairs of config C5 (profiles/r3_experiments.md).  So the piped form keeps the
// plain step.
#ifndef GNX_CL_PRIO
#define GNX_CL_PRIO 1
#endif
template <bool P16, bool PIPED, bool REBASE = false>
__device__ __forceinline__ void cl_sweep_body(int *__restrict__ lds, const PairPlan *__restrict__ plans, int n_pairs,
                                              const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                              const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                              const KParams &kp, int *__restrict__ rowbuf, int *__restrict__ snap, int64_t *__restrict__ hfin,
                                              int *__restrict__ err, const int2 *__restrict__ strip_map, int *__restrict__ strip_prog,
                                              long long *__restrict__ bases = nullptr) {
    // REBASE: pl.rowi_off = the pair's slice of `bases` (int64 per strip and K-step block), pl.s_pitch = blocks per strip
    using PC = ProfCfg<P16>;
    constexpr int LW = PC::LW, BST = PC::BST, PTOT = PC::TOTAL;
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    if (lane < 25) lds[lane] = kp.sc4[lane] - 2 * kp.g4; // rebased diagonal move: 4*(s - 2g); every value carries tag 2
    int *prof = &lds[32 + PC::pair_off(g)];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    // (PIPED instantiation: `piped` stays a run-time value on purpose -- with the un-piped branches folded away the compiler allocates 77
    // registers instead of 92 and schedules the block loop so that the launch takes 3.5 % longer: 242 against 234 ms, 1024 pairs of C5)
    const bool piped = PIPED && strip_map != nullptr;
    constexpr bool PF = !PIPED; // look-ahead form, see above
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
    const uint8_t *ap = a_buf + (valid ? a_start[p] : 0);
    BetaBytes bp;
    bp.init(b_buf, kp, valid ? b_start[p] : 0, valid ? pl.m : 0);
    const int Tend = (m_max + 15 + 15) & ~15;
    int bad = 0;

    const int s_own = piped ? strip_map[blockIdx.x].y : 0;
    const int s_lo = piped ? s_own - n_stolen : 0, s_hi = piped ? s_own + 1 : S_max;
    const int64_t rb_pitch = (int64_t)pl.m + 1;
    for (int s = s_lo; s < s_hi; s++) {
        const int bid = (int)blockIdx.x - s_own + s; // piped: block index of strip s of this group = its slot in strip_prog
        const bool gact = valid && s < pl.strips;
        const int m_eff = gact ? pl.m : 0;
        int m_min = 0x7fffffff;
        for (int q = 0; q < 4; q++) m_min = min(m_min, (pbase + q < n_pairs && s < plans[pbase + q].strips) ? plans[pbase + q].m : 0);
        const bool store_row = gact && (s + 1 < pl.strips);
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
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = P16 ? ((lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16)) : lds[a5[k] + b];
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < R; r++) val[r] = 2; // column 0, rebased: 0 (tag 2)
        int diag0 = 2;
        int v_out = 0, b_out = 0, sq_v = 0;
        int qv, qb, nv = 0, nb = 0;
        // REBASE: this strip's base; the strip above's bases of the two K-step blocks the columns being loaded were written in, as
        // differences to mine (dlo: block qp, dhi: block qp + 1 = columns c with c + 14 >= edge); row 0's key relative to my base
        long long Bown = 0;
        int dlo = 0, dhi = 0, qp = 0, edge = kp.ckc, r0v = 2;
        bool dhi_ok = false;
        long long *my_bases = REBASE ? bases + pl.rowi_off + (int64_t)s * pl.s_pitch : nullptr;
        auto bprod = [&](int q) -> long long { return s == 0 ? 0LL : rbase_load(my_bases - pl.s_pitch + q, piped); /* (block 0 too: 0 inside a pair, the frame shift of a row panel's stand-in strip, run_device_mega) */ };
        auto boundary = [&](int c, int &ov, int &ob) {
            if (s == 0) ov = REBASE ? r0v : 2; // row 0, rebased
            else if (c >= 1 && c <= m_eff) {
                ov = rb_load32(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], piped);
                if (REBASE) ov += (c + 14 >= edge) ? dhi : dlo;
            }
            else ov = 0;
            int b = 0;
            if (c >= 1 && c <= m_eff) b = bp.raw(c - 1);
            if (PF) { ob = b; return; } // RAW base: base_off() turns it into the LDS offset where the queue is needed (no wait on the load here)
            if (c >= 1 && c <= m_eff) b = bp.value(b, c - 1);
            if (b >= 5) { bad = 1; b = 4; }
            ob = b * (BST * 4);
        };
        auto base_off = [&](int raw, int c) { int b = (c >= 1 && c <= m_eff) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); }; // LDS byte offset of the base's profile plane
        int rb_seen = 0, pf_seen = 0;
        auto wait_rows = [&](int cmax) {
            if (piped && s > 0 && rb_seen < cmax) {
                const long long t_begin = wall_clock64();
                while ((rb_seen = rb_progress(&strip_prog[bid - 1])) < cmax) {
                    __builtin_amdgcn_s_sleep(32);
                    if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); break; }
                }
            }
        };
        if (!piped && s > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        wait_rows(G);
        boundary(l + 1, qv, qb);
        if (PF) qb = base_off(qb, l + 1);

        int wq[LW], pb_cur = 0; // PF: the profile entries and the base of the CURRENT step (fetched during the step before)
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
        // take / nqv (PF): at the last step of a block the base queue of the next one takes over
        auto step = [&](const int t, auto chk, const bool take, const int nqv) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_v = dpp_shr1(qv, v_out);
            qv = dpp_shl1(qv, qv);
            int w[LW], wn[LW], pb_next = 0;
            if (PF) {
                if (take) qb = nqv;
                pb_next = dpp_shr1(qb, pb_cur);
                qb = dpp_shl1(qb, qb);
                fetch(pb_next, wn);
                asm volatile("" ::: "memory"); // the reads stay HERE, ahead of the arithmetic (the scheduler would sink them next to their first use)
#pragma unroll
                for (int k = 0; k < LW; k++) w[k] = wq[k];
            } else {
                const int pb = dpp_shr1(qb, b_out);
                qb = dpp_shl1(qb, qb);
                b_out = pb;
                fetch(pb, w);
            }
            const int j = t - l;
            if (!CHECK || (j >= 1 && j <= m_eff)) {
                int vd = diag0, vu = up_v;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    const int k = max3i(vd + S4, val[r], vu);
                    vd = val[r];
                    val[r] = k;
                    vu = k;
                }
                diag0 = up_v;
                v_out = vu;
            }
            sq_v = dpp_shl1(v_out, sq_v);
            if (PF) {
#pragma unroll
                for (int k = 0; k < LW; k++) wq[k] = wn[k];
                pb_cur = pb_next;
            }
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            if (REBASE && t0 > 0 && t0 % kp.ckc == 0) { // move the base (see REBASE above): everything this strip holds, relative to the key of its first row's current cell
                const int rep = __shfl(val[0], lane & 48, 64);
                const bool rb_on = gact && t0 <= m_eff + 15; // (a wave runs on for its longest pair: a pair whose last lane has passed column m is done, and has no base slots beyond)
                const int d = rb_on ? (rep & ~3) : 0;
#pragma unroll
                for (int r = 0; r < R; r++) val[r] -= d;
                diag0 -= d; v_out -= d; qv -= d;
                Bown += d; dlo -= d; dhi -= d;
                r0v = rbase_const(2, Bown);
                if (rb_on && l == 0) rbase_store(my_bases + t0 / kp.ckc, Bown, piped);
            }
            if (t0 > 0 && t0 % kp.ckc == 0 && gact && t0 <= m_eff + 15 && (!REBASE || snap != nullptr)) { // snapshot: the state the wave resumes from at step t0 (REBASE, no buffer: a forward pass of row panels, which keeps none)
                uint4 *dst = reinterpret_cast<uint4 *>(snap + pl.ckpt_off + (((int64_t)(t0 / kp.ckc - 1) * pl.strips + s) * G + l) * SNAPW);
                dst[0] = make_uint4((unsigned)val[0], (unsigned)val[1], (unsigned)val[2], (unsigned)val[3]);
                dst[1] = make_uint4((unsigned)val[4], (unsigned)val[5], (unsigned)val[6], (unsigned)val[7]);
                dst[2] = make_uint4((unsigned)val[8], (unsigned)val[9], (unsigned)diag0, 0u);
            }
            // the progress word of the strip above is polled ONE BLOCK AHEAD (the load issued in the previous block lands while its 16
            // steps run); only a strip that has caught up with its producer falls into the blocking spin of wait_rows
            if (piped && s > 0) { rb_seen = max(rb_seen, pf_seen); if (rb_seen < t0 + 5 * G) pf_seen = rb_progress(&strip_prog[bid - 1]); }
#if GNX_CL_PRIO
            // Issue priority by slack (round 3): a strip that has columns in hand -- or the top strip -- is what the strips below it wait
            // for; one that runs right behind its producer will stop at the next block anyway.  Raising the former over the latter in the
            // SIMD's arbitration: 2048 pairs of C5 400 -> 382 ms (levels 2 / 0, 3 / 0 and 3 / 1 / 0 measure the same).
            if (piped) {
                const int slack = s == 0 ? (1 << 30) : __builtin_amdgcn_readfirstlane(rb_seen) - t0;
                if (slack >= 5 * G) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
            }
#endif
            wait_rows(t0 + 2 * G);
            if (REBASE && s > 0 && gact) { // the columns loaded now are t0 + 17 .. t0 + 32: written by the strip above in its blocks (c + 14) / K
                while (t0 + 31 >= edge) { qp++; edge += kp.ckc; dlo = dhi_ok ? dhi : rbase_delta(bprod(qp), Bown); dhi_ok = false; }
                if (!dhi_ok && t0 + 46 >= edge) { dhi = rbase_delta(bprod(qp + 1), Bown); dhi_ok = true; }
            }
            boundary(t0 + 16 + l + 1, nv, nb);
            if (t0 >= 16 && t0 + 16 <= m_min) {
#pragma unroll
                for (int u = 0; u < 16; u++) { if (PF && u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::false_type{}, u == 15, nb); }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (PF && u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, u == 15, nb); }
            }
            if (PF) asm volatile("" :: "v"(nv)); // the row values loaded at the top of this block are consumed before the store below goes out (else: a vmcnt(0) behind it)
            qv = nv;
            if (!PF) qb = nb;
            if (store_row) {
                const int c = t0 + l - 14;
                if (c >= 1 && c <= m_eff) rb_store32(&rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + c], sq_v, piped);
            }
            if (piped && ((t0 + 16) & (kp.rb_pub - 1)) == 0) rb_publish(&strip_prog[bid], t0 + 1, lane);
        }
        if (gact && m_eff >= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) if (row0 + r + 1 == pl.n) hfin[pl.hcol_off] = (Bown >> 2) + (int64_t)(val[r] >> 2) + (int64_t)(kp.g4 >> 2) * ((int64_t)pl.n + m_eff); // plain V(n, m)
        }
        if (piped) rb_publish(&strip_prog[bid], 0x7fffffff, lane);
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    if (bad) atomicOr(err, 1);
}
// the piped form wants as many waves per SIMD as fit (it waits); the un-piped form, compute-bound, runs best with the registers of four
// waves per SIMD for its look-ahead (103 registers: 9.0 ms for 320 x 10 000 x 32 768; squeezed into the 96 of five waves: 9.8 ms)
template <bool P16, bool REBASE = false>
__global__ __launch_bounds__(64) void cl_sweep_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                      const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                      const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                      KParams kp, int *__restrict__ rowbuf, int *__restrict__ snap, int64_t *__restrict__ hfin,
                                                      int *__restrict__ err, const int2 *__restrict__ strip_map, int *__restrict__ strip_prog,
                                                      long long *__restrict__ bases) {
    __shared__ int lds[32 + ProfCfg<P16>::TOTAL];
    cl_sweep_body<P16, true, REBASE>(lds, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, rowbuf, snap, hfin, err, strip_map, strip_prog, bases);
}
template <bool P16, bool REBASE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((P16 && !REBASE) ? 4 : 3, (P16 && !REBASE) ? 4 : 3))) void cl_sweep_flat_kernel( // (the int32 profile form needs 3 waves' worth of registers; GNX_CL_P16=0 only)
    const PairPlan *__restrict__ plans, int n_pairs, const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
    const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start, KParams kp, int *__restrict__ rowbuf, int *__restrict__ snap,
    int64_t *__restrict__ hfin, int *__restrict__ err, long long *__restrict__ bases) {
    __shared__ int lds[32 + ProfCfg<P16>::TOTAL];
    cl_sweep_body<P16, false, REBASE>(lds, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, rowbuf, snap, hfin, err, nullptr, nullptr, bases);
}

// Row PANELS (run_device_mega): a pair whose bottom rows + snapshots exceed the workspace (5 Mb x 5 Mb: 1.9 TB) is swept panel by panel --
// a panel = as many 160-row strips as fit, its top boundary = the bottom row of the panel above, kept with its bases -- and walked back panel
// by panel: the walk of a REBASE kernel given a MegaState stops where it steps from the panel's first real strip into the row above it
// (`virt` = 160: the panel's strip 0 is a stand-in whose bottom row the host has put into the row buffer) and resumes in the panel above
// with everything it carries.  Rows are panel-local; row_off turns them into the pair's (checkerboard bookkeeping of quirks Q1 / Q2).
struct MegaState {
    int32_t resume, done, wi, wj, wk, pend, cur_op, last_op;
    int64_t li, cnt, cur_run, row_off;
    int32_t virt, pad;
};

__global__ __launch_bounds__(64) void add_i64_kernel(long long *__restrict__ b, int64_t n, long long delta) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) b[x] += delta;
}

// One wave per NP pairs (NP = 4, 2 or 1: lanes 16 * NP .. 63 idle).  The walk of a pair is one long chain of dependent steps (re-fill a
// tile, walk it with one lane, next tile), so a launch is as fast as its slowest wave -- and a wave of four pairs re-fills, every round,
// as many blocks as the pair that needs most, and walks until the last of the four has left its tile.  With one pair per wave a
// workgroup needs a quarter of the LDS, so the same number of pairs is resident (8 workgroups per CU instead of 2) on all four SIMDs
// instead of two, and nobody waits for a neighbour.  Per round every pair (16 lanes) re-fills the tile its walk is in -- strip s = (i-1)/160, steps
// (c*CKC, j + lane(i)] of that strip's wavefront -- into LDS, then lane 0 of the pair walks inside the tile until it leaves it.
// Runs are staged in traceback order at scr[scr_off[p] ..] (n + m + 2 entries per pair); reverse_runs_kernel puts them in place.
template <bool P16, int NP, int CK, bool REBASE = false>
__global__ __launch_bounds__(64) void cl_walk_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                     const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                     const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                     KParams kp, TbParams tp, const int *__restrict__ rowbuf, const int *__restrict__ snap,
                                                     const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                     const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err,
                                                     const long long *__restrict__ bases, MegaState *__restrict__ mst = nullptr) {
    using PC = ProfCfg<P16>;
    static_assert(NP == 1 || NP == 2 || NP == 4, "pairs per workgroup");
    // profile of NP pairs: the layout of ProfCfg (one or two duos), or, for a single pair, planes of its own 16 * LW dwords (a plane
    // stride of 80 or 160 dwords keeps the 16 lanes of one pair on 16 distinct banks whatever bases they look up)
    constexpr int LW = PC::LW, BST = NP == 1 ? G * LW : PC::BST, PTOT = NP == 1 ? 5 * BST : (NP == 2 ? PC::TOTAL / 2 : PC::TOTAL);
    static_assert(CK % 16 == 0 && CK <= CKC, "snapshot spacing");
    constexpr int DIRG = (CK / 16) * R * G + 16; // LDS dwords of one pair's tile (+16: neighbouring pairs start in different banks)
    __shared__ int lds[32 + PTOT + NP * DIRG];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    if (lane < 25) lds[lane] = kp.sc4[lane] - 2 * kp.g4 + 1; // pre-tagged diagonal candidate (tag 3), see fill_const_kernel
    const int gl = g < NP ? g : 0; // (idle lane groups compute on pair 0's addresses and store nothing)
    int *prof = &lds[32 + (NP == 1 ? 0 : PC::pair_off(gl))];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    unsigned *dirg = reinterpret_cast<unsigned *>(&lds[32 + PTOT + gl * DIRG]);
    const int p = blockIdx.x * NP + g;
    const bool valid = g < NP && p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const uint8_t *ap = a_buf + (valid ? a_start[p] : 0);
    BetaBytes bp;
    bp.init(b_buf, kp, valid ? b_start[p] : 0, valid ? pl.m : 0);
    const int64_t rb_pitch = (int64_t)pl.m + 1;
    const int po = pl.src;
    int bad = 0;
    // walker state (lane 0 of the pair)
    int wi = pl.n, wj = pl.m, wdone = valid ? 0 : 1;
    int64_t cnt = 0, cur_run = 0;
    int cur_op = -1, last_op = -1;
    const int64_t sbase = valid ? scr_off[p] : 0;
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
    bool pexit = false; // the walk has left this panel upwards: it goes on in the panel above
    if (REBASE && mst) {
        virt = mst->virt; row_off = mst->row_off;
        if (mst->resume && valid) { wi = mst->wi; wj = mst->wj; cnt = mst->cnt; cur_run = mst->cur_run; cur_op = mst->cur_op; last_op = mst->last_op; }
    }

    while (true) {
        const int src0 = lane & 48;
        const int ci = __shfl(wi, src0, 64), cj = __shfl(wj, src0, 64), cdone = __shfl(wdone, src0, 64);
        if (__all(cdone)) break;
        if (REBASE && virt > 0 && __any(!cdone && ci <= virt)) { pexit = true; break; }
        const bool gact = !cdone;
        const int s = gact ? (ci - 1) / H : 0;
        const int lw = gact ? (ci - 1 - s * H) / R : 0;
        const int tend = gact ? cj + lw : 0;   // step of the cell the walk is at
        const int c = gact ? (tend - 1) / CK : 0;
        const int tbeg = c * CK;
        const int nblk = gact ? (tend - tbeg + 15) >> 4 : 0;
        int nblk_max = nblk;
        nblk_max = max(nblk_max, __shfl_xor(nblk_max, 16, 64));
        nblk_max = max(nblk_max, __shfl_xor(nblk_max, 32, 64));
        const int m_eff = gact ? pl.m : 0;
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
            __syncthreads(); // table visible; the previous round's walk is over
            if (NP == 4 || g < NP) {
#pragma unroll
                for (int b = 0; b < 5; b++) {
#pragma unroll
                    for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = P16 ? ((lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16)) : lds[a5[k] + b];
                }
            }
            __syncthreads();
        }
        int diag0 = 2;
#pragma unroll
        for (int r = 0; r < R; r++) { val[r] = 2; acc[r] = 0; }
        int v_out = 0, b_out = 0;
        if (gact && c > 0) { // resume from the snapshot of step tbeg
            const uint4 *sp = reinterpret_cast<const uint4 *>(snap + pl.ckpt_off + (((int64_t)(c - 1) * pl.strips + s) * G + l) * SNAPW);
            const uint4 x0 = sp[0], x1 = sp[1], x2 = sp[2];
            val[0] = (int)x0.x; val[1] = (int)x0.y; val[2] = (int)x0.z; val[3] = (int)x0.w;
            val[4] = (int)x1.x; val[5] = (int)x1.y; val[6] = (int)x1.z; val[7] = (int)x1.w;
            val[8] = (int)x2.x; val[9] = (int)x2.y; diag0 = (int)x2.z;
            v_out = val[R - 1];
            const int jb = tbeg - l; // the column this lane processed at step tbeg: its base goes to the next lane
            if (jb >= 1 && jb <= m_eff) { int b = bp.at(jb - 1); if (b >= 5) { bad = 1; b = 4; } b_out = b * (BST * 4); }
        }
        int qv, qb, nv = 0, nb = 0;
        // REBASE: the snapshot's keys are relative to the strip's base of block c; the row above, block by block, to the bases of the strip above
        long long Bt = 0;
        if (REBASE && gact && c > 0) Bt = bases[pl.rowi_off + (int64_t)s * pl.s_pitch + c];
        const int r0v = REBASE ? rbase_const(2, Bt) : 2;
        auto boundary = [&](int cc, int &ov, int &ob) {
            if (s == 0) ov = r0v;
            else if (cc >= 1 && cc <= m_eff) {
                ov = rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + cc];
                if (REBASE) { const int q = (cc + 14) / CK; ov += rbase_delta(bases[pl.rowi_off + (int64_t)(s - 1) * pl.s_pitch + q], Bt); }
            }
            else ov = 0;
            int b = 0;
            if (cc >= 1 && cc <= m_eff) { b = bp.at(cc - 1); if (b >= 5) { bad = 1; b = 4; } }
            ob = b * (BST * 4);
        };
        boundary(tbeg + l + 1, qv, qb);
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
        for (int b = 0; b < nblk_max; b++) {
            const int t0 = tbeg + 16 * b; // per pair
            boundary(t0 + 16 + l + 1, nv, nb);
            if (__all(!gact || (t0 >= 16 && t0 + 16 <= m_eff))) {
#pragma unroll
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::false_type{});
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) step(t0 + u + 1, std::true_type{});
            }
            qv = nv; qb = nb;
            if (gact && b < nblk) {
                const int miss = (t0 + 16 - l) - m_eff; // steps this lane sat idle after its last column
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
                for (int r = 0; r < R; r++) dirg[(b * R + r) * G + l] = acc[r] >> sh;
            }
        }
        __syncthreads();
        if (l == 0 && gact) {
            int i = wi, j = wj;
            while (true) {
                if (i == 0 || j == 0) { wdone = 1; break; }
                const int i0 = i - 1 - s * H;
                if (i0 < 0) break; // left the strip through its top edge
                const int l2 = i0 / R, r2 = i0 - l2 * R;
                const int t1 = j + l2 - 1 - tbeg;
                if (t1 < 0) break; // left the tile through its (skewed) left edge
                const int pos = t1 & 15;
                const unsigned w = dirg[((t1 >> 4) * R + r2) * G + l2];
                int tag = (int)((w >> (2 * pos)) & 3u);
                if (tag == 0) { atomicOr(err, 2); wdone = 1; break; } // impossible direction: the Go code would log.Fatalf
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
    if (l == 0 && valid && REBASE && mst) {
        mst->wi = wi; mst->wj = wj; mst->cnt = cnt; mst->cur_run = cur_run; mst->cur_op = cur_op; mst->last_op = last_op; mst->done = pexit ? 0 : 1;
    }
    if (l == 0 && valid && !pexit) {
        // Step 4 (constGap.go:59-63): the leading gap is appended only if the walk left through exactly one edge of its last
        // checkerboard; a corner exit appends nothing, even when it is not the origin (quirk Q2)
        const int64_t gi = (int64_t)wi + (wi > 0 ? row_off : 0); // (row of the pair; a panel with a stand-in strip ends here only through column 0)
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
