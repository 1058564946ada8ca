// affine_long.hip.h -- affine-gap alignments of long alpha WITHOUT a stored direction matrix: score-only sweep + snapshots, fused re-fill / walk
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.6.
#pragma once
#include "const_long.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// The scheme of const_long.hip.h for the three-state recurrence (align/affineGap.go:151-273, affineGap_highMem.go:181-223, global
// modes, gapOpen <= 0): reads longer than the fast path's 320 rows, 1 kb x 1 kb pairs, the 10 kb x 10 kb pair of cmd/cigarToBed.
//   al_sweep_kernel  score only: the rebased h-form of fill_affine_kernel without tags and without the three accumulators per row
//                    (add, max3, add, max, max = 5 VALU instructions per cell instead of 11; a lane's last row keeps its tags, 9 instructions).  Keeps the bottom row {dn, h} of every strip
//                    (the hand-over buffer, 8 B per column) and a snapshot of the wavefront every CKA steps: rt[10], hold[10], the
//                    diagonal value and the D value last handed to the next lane (24 dwords per lane).
//   al_walk_kernel   one wave per 4 pairs: re-fill the tile (strip s, steps of snapshot interval c up to one step past the cell the walk
//                    is at) with the recording recurrence into three 2-bit planes in LDS (17 KB per pair), walk inside it with the state
//                    machine of traceback_kernel<true> (quirks Q1 / Q2), move on.  The argmax tags of h in the last column -- the
//                    start state, and quirk Q1 when a checkerboard is left upwards in column m -- come from the re-fill itself: lanes
//                    that have passed column m leave their keys in LDS.
// Untagged arithmetic is exact: tag bits are junk < 4 that never changes the value of a max (score differences are multiples of 4);
// a re-fill resumes from the snapshot's values with fresh tags.
// ------------------------------------------------------------------------------------------------------
constexpr int CKA = 128;                          // snapshot spacing in wavefront steps
constexpr int AL_WORDS = CKA / 16 + 1;            // direction words per plane row of a tile (up to CKA + 4 steps: two unusable, one past the entry cell)
constexpr int AL_SNAPW = 24;                      // dwords per lane per snapshot
constexpr int AL_DIRG = AL_WORDS * 3 * R * G + 16; // LDS dwords of one pair's tile

// REBASE (const_long.hip.h): the strip's keys relative to a base it moves along every CKA steps; pl.rowi_off / pl.s_pitch = its slice of `bases`
template <bool P16, bool REBASE = false>
__global__ __launch_bounds__(64) void al_sweep_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                      const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                      const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                      KParams kp, int2 *__restrict__ rowbuf, int *__restrict__ snap, int64_t *__restrict__ hfin,
                                                      int *__restrict__ err, const int2 *__restrict__ strip_map, int *__restrict__ strip_prog,
                                                      long long *__restrict__ bases) {
    using PC = ProfCfg<P16>;
    constexpr int LW = PC::LW, BST = PC::BST, PTOT = PC::TOTAL;
    constexpr int TI = 2, TD = 1;
    __shared__ int lds[32 + PTOT];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    if (lane < 25) lds[lane] = kp.sc4[lane] - 2 * kp.e4; // rebased diagonal move: 4*(s - 2e)
    int *prof = &lds[32 + PC::pair_off(g)];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    const bool piped = strip_map != nullptr;
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
    const int OE4 = kp.oe4, E4 = kp.e4, RB = kp.e4;
    int vO4;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vO4) : "s"(kp.o4));
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
        int rt[R], hold[R];
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
        for (int r = 0; r < R; r++) { // column 0 (fill_affine_kernel): M = I = -inf, D = D00 + i*ecol, rebased
            const int i = row0 + r + 1;
            const int D1c = kp.d00_4 + i * kp.ecol4 + TD - RB * i;
            hold[r] = max3i(NEG4 + 3, NEG4 + TI, D1c);
            rt[r] = max3i(NEG4 + 3 + OE4, NEG4 + TI + E4, D1c + OE4) - RB;
        }
        int diag0 = (row0 == 0) ? max3i(3, kp.o4 + TI, kp.d00_4 + TD) : max3i(NEG4 + 3, NEG4 + TI, kp.d00_4 + row0 * kp.ecol4 + TD - RB * row0);
        int dn_out = 0, h_out = 0, b_out = 0, sq_dn = 0, sq_h = 0;
        int qdn, qh, qb, ndn = 0, nh = 0, nb = 0;
        // REBASE: my base; the bases of the strip above for the two CKA-step blocks the columns being loaded were written in, as differences
        // to mine (dlo: block qp, dhi: block qp + 1 = columns c with c + 14 >= edge); row 0's I key relative to my base
        long long Bown = 0;
        int dlo = 0, dhi = 0, qp = 0, edge = CKA, r0i = kp.o4 + TI;
        bool dhi_ok = false;
        long long *my_bases = REBASE ? bases + pl.rowi_off + (int64_t)s * pl.s_pitch : nullptr;
        auto bprod = [&](int q) -> long long { return s == 0 ? 0LL : rbase_load(my_bases - pl.s_pitch + q, piped); /* (block 0 too: 0 inside a pair, the frame shift of a row panel's stand-in strip, run_device_mega) */ };
        auto boundary = [&](int c, int &odn, int &oh, int &ob) {
            if (s == 0) {
                const int M3 = NEG4 + 3, I2 = REBASE ? r0i : kp.o4 + c * E4 + TI - RB * c, D1 = NEG4 + TD; // row 0: I(0,c) = gapOpen + c*gapExtend
                oh = max3i(M3, I2, D1);
                odn = max3i(M3 + OE4, I2 + OE4, D1 + E4) - RB;
            } else if (c >= 1 && c <= m_eff) {
                const int2 v = rb_load(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], piped);
                odn = v.x; oh = v.y;
                if (REBASE) { const int dd = (c + 14 >= edge) ? dhi : dlo; odn += dd; oh += dd; }
            } else { odn = 0; oh = 0; }
            ob = (c >= 1 && c <= m_eff) ? bp.raw(c - 1) : 0; // RAW base: base_off() turns it into the LDS offset where the queue is needed (no wait on the load here)
        };
        auto base_off = [&](int raw, int c) { int b = (c >= 1 && c <= m_eff) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); }; // LDS byte offset of the base's profile plane (c: the column the raw base was loaded for)
        int rb_seen = 0;
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
        boundary(l + 1, qdn, qh, qb);
        qb = base_off(qb, l + 1);

        // profile entries one step ahead (software pipeline over the LDS round trip, see cl_sweep_kernel)
        int wq[LW], pb_cur;
        auto fetch = [&](int pbv, int *w) {
            const int *pw = reinterpret_cast<const int *>(prof_lane + pbv);
#pragma unroll
            for (int k = 0; k < LW; k++) w[k] = pw[k];
        };
        pb_cur = dpp_shr1(qb, b_out);
        qb = dpp_shl1(qb, qb);
        fetch(pb_cur, wq);
        auto step = [&](const int t, auto chk, const bool take, const int nqv) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_dn = dpp_shr1(qdn, dn_out);
            const int up_h = dpp_shr1(qh, h_out);
            qdn = dpp_shl1(qdn, qdn);
            qh = dpp_shl1(qh, qh);
            if (take) qb = nqv; // (last step of a block: the base queue of the next one takes over)
            const int pb_next = dpp_shr1(qb, pb_cur);
            qb = dpp_shl1(qb, qb);
            int wn[LW];
            fetch(pb_next, wn);
            asm volatile("" ::: "memory"); // the reads stay HERE, ahead of the arithmetic
            const int j = t - l;
            const int *w = wq;
            if (!CHECK || (j >= 1 && j <= m_eff)) {
                int hd = diag0, dnu = up_dn;
#pragma unroll
                for (int r = 0; r < R - 1; r++) {
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    const int hnew = max3i(hd + S4, rt[r], dnu);
                    const int ho = hnew + vO4;
                    rt[r] = max(ho, rt[r]);
                    const int dnn = max(ho, dnu);
                    hd = hold[r];
                    hold[r] = hnew;
                    dnu = dnn;
                }
                { // the lane's last row with argmax tags: what the last lane hands to the next strip (row buffer) must carry them -- the
                  // planes of a strip's first row record where the values from above came from
                    constexpr int r = R - 1;
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
                h_out = hold[R - 1];
            }
            sq_dn = dpp_shl1(dn_out, sq_dn);
            sq_h = dpp_shl1(h_out, sq_h);
#pragma unroll
            for (int k = 0; k < LW; k++) wq[k] = wn[k];
            pb_cur = pb_next;
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            if (REBASE && t0 > 0 && t0 % CKA == 0) { // move the base: everything this strip holds, relative to h of its first row's current cell
                const int rep = __shfl(hold[0], lane & 48, 64);
                const bool rb_on = gact && t0 <= m_eff + 15; // (a pair whose last lane has passed column m is done, and has no base slots beyond)
                const int d = rb_on ? (rep & ~3) : 0;
#pragma unroll
                for (int r = 0; r < R; r++) { rt[r] -= d; hold[r] -= d; }
                diag0 -= d; dn_out -= d; h_out -= d; qdn -= d; qh -= d;
                Bown += d; dlo -= d; dhi -= d;
                r0i = rbase_const((long long)kp.o4 + TI, Bown);
                if (rb_on && l == 0) rbase_store(my_bases + t0 / CKA, Bown, piped);
            }
            if (t0 > 0 && t0 % CKA == 0 && gact && t0 <= m_eff + 15 && (!REBASE || snap != nullptr)) { // snapshot: the state the wave resumes from at step t0 (REBASE, no buffer: a forward pass of row panels, which keeps none)
                uint4 *dst = reinterpret_cast<uint4 *>(snap + pl.ckpt_off + (((int64_t)(t0 / CKA - 1) * pl.strips + s) * G + l) * AL_SNAPW);
                dst[0] = make_uint4((unsigned)rt[0], (unsigned)rt[1], (unsigned)rt[2], (unsigned)rt[3]);
                dst[1] = make_uint4((unsigned)rt[4], (unsigned)rt[5], (unsigned)rt[6], (unsigned)rt[7]);
                dst[2] = make_uint4((unsigned)rt[8], (unsigned)rt[9], (unsigned)hold[0], (unsigned)hold[1]);
                dst[3] = make_uint4((unsigned)hold[2], (unsigned)hold[3], (unsigned)hold[4], (unsigned)hold[5]);
                dst[4] = make_uint4((unsigned)hold[6], (unsigned)hold[7], (unsigned)hold[8], (unsigned)hold[9]);
                dst[5] = make_uint4((unsigned)diag0, (unsigned)dn_out, 0u, 0u);
            }
            wait_rows(t0 + 2 * G);
            if (REBASE && s > 0 && gact) { // the columns loaded now are t0 + 17 .. t0 + 32: written by the strip above in its blocks (c + 14) / CKA
                while (t0 + 31 >= edge) { qp++; edge += CKA; dlo = dhi_ok ? dhi : rbase_delta(bprod(qp), Bown); dhi_ok = false; }
                if (!dhi_ok && t0 + 46 >= edge) { dhi = rbase_delta(bprod(qp + 1), Bown); dhi_ok = true; }
            }
            boundary(t0 + 16 + l + 1, ndn, nh, nb);
            if (t0 >= 16 && t0 + 16 <= m_min) {
#pragma unroll
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::false_type{}, u == 15, nb); }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, u == 15, nb); }
            }
            // consume the loads issued at the top of this block BEFORE the stores below are issued (exact wait, nothing newer in flight;
            // left to their first real use -- the DPP moves of the next block -- the wait becomes a vmcnt(0) behind those stores)
            asm volatile("" :: "v"(ndn), "v"(nh));
            qdn = ndn; qh = nh;
            if (store_row) {
                const int c = t0 + l - 14;
                if (c >= 1 && c <= m_eff) rb_store(&rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + c], sq_dn, sq_h, piped);
            }
            if (piped && ((t0 + 16) & (RB_PUB - 1)) == 0) rb_publish(&strip_prog[bid], t0 + 1, lane);
        }
        if (gact && m_eff >= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) if (row0 + r + 1 == pl.n) hfin[pl.hcol_off] = (Bown + (int64_t)hold[r] + (int64_t)RB * ((int64_t)pl.n + m_eff)) >> 2; // plain score h(n, m)
        }
        if (piped) rb_publish(&strip_prog[bid], 0x7fffffff, lane);
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    if (bad) atomicOr(err, 1);
}

template <bool P16, bool REBASE = false>
__global__ __launch_bounds__(64) void al_walk_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                     const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                     const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                     KParams kp, TbParams tp, const int2 *__restrict__ rowbuf, const int *__restrict__ snap,
                                                     const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                     const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err,
                                                     const long long *__restrict__ bases, MegaState *__restrict__ mst = nullptr) {
    using PC = ProfCfg<P16>;
    constexpr int LW = PC::LW, BST = PC::BST, PTOT = PC::TOTAL;
    constexpr int TI = 2, TD = 1;
    __shared__ int lds[32 + PTOT + 4 * AL_DIRG + 4 * H];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    if (lane < 25) lds[lane] = kp.sc4[lane] - 2 * kp.e4;
    int *prof = &lds[32 + PC::pair_off(g)];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    unsigned *dirg = reinterpret_cast<unsigned *>(&lds[32 + PTOT + g * AL_DIRG]);
    int *hcolT = &lds[32 + PTOT + 4 * AL_DIRG + g * H]; // keys h(i, m) of the strip's rows whose lanes have passed column m
    const int p = blockIdx.x * 4 + g;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const uint8_t *ap = a_buf + (valid ? a_start[p] : 0);
    BetaBytes bp;
    bp.init(b_buf, kp, valid ? b_start[p] : 0, valid ? pl.m : 0);
    const int64_t rb_pitch = (int64_t)pl.m + 1;
    const int po = pl.src;
    const int OE4 = kp.oe4, E4 = kp.e4, RB = kp.e4;
    int vO4;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vO4) : "s"(kp.o4));
    int bad = 0;
    // walker state (lane 0 of the pair); pend: the state k is taken from the argmax tag of h(wi, m) once its tile is there (the start,
    // and quirk Q1 when a checkerboard is left upwards in the last column into another strip)
    int wi = pl.n, wj = pl.m, wk = 0, wdone = valid ? 0 : 1, pend = 1;
    int q1n = 0, q1c = 0; // quirk-Q1 restarts met / that changed the state (gnx_debug_counter(3 / 4))
    int64_t li = (pl.n > 0) ? (int64_t)(pl.n - 1) % tp.ci : 0;
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
    // row panels (MegaState, const_long.hip.h): resume what the walk carried out of the panel below; stop where it leaves this one upwards
    int virt = 0;
    int64_t row_off = 0;
    bool pexit = false;
    if (REBASE && mst) {
        virt = mst->virt; row_off = mst->row_off;
        if (valid) {
            if (mst->resume) { wi = mst->wi; wj = mst->wj; wk = mst->wk; pend = mst->pend; li = mst->li; cnt = mst->cnt; cur_run = mst->cur_run; cur_op = mst->cur_op; last_op = mst->last_op; }
            else li = (pl.n > 0) ? ((int64_t)pl.n + row_off - 1) % tp.ci : 0;
        }
    }

    while (true) {
        const int src0 = lane & 48;
        const int ci = __shfl(wi, src0, 64), cj = __shfl(wj, src0, 64), cdone = __shfl(wdone, src0, 64);
        if (__all(cdone)) break;
        if (REBASE && virt > 0 && __any(!cdone && ci <= virt)) { pexit = true; break; }
        const bool gact = !cdone;
        const int s = gact ? (ci - 1) / H : 0;
        const int lw = gact ? (ci - 1 - s * H) / R : 0;
        const int te = gact ? cj + lw : 0;    // step of the cell the walk is at
        // A snapshot carries values, no argmax tags, so the plane fields of the first two steps after it are not usable (inputs of step 1:
        // the snapshot's keys; the diagonal input of a lane's first row at step 2: what the lane above handed over before step 1).  The
        // tile is chosen so that the walk's cell is at least its third step, and the walk leaves it before its second.
        const int c = (gact && te >= 3) ? (te - 3) / CKA : 0;
        const int tbeg = c * CKA;
        const int tmin = c > 0 ? 2 : 0;       // first usable step of the tile, minus one
        const int tend = te + 1;              // one step further: quirk Q1 looks at the cell to the right of the one a vertical move leaves
        const int nblk = gact ? (tend - tbeg + 15) >> 4 : 0;
        int nblk_max = nblk;
        nblk_max = max(nblk_max, __shfl_xor(nblk_max, 16, 64));
        nblk_max = max(nblk_max, __shfl_xor(nblk_max, 32, 64));
        const int m_eff = gact ? pl.m : 0;
        const int row0 = s * H + l * R;
        int rt[R], hold[R];
        unsigned acc[3 * R];
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
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = P16 ? ((lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16)) : lds[a5[k] + b];
            }
            __syncthreads();
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
        if (gact && c > 0) { // resume from the snapshot of step tbeg
            const uint4 *sp = reinterpret_cast<const uint4 *>(snap + pl.ckpt_off + (((int64_t)(c - 1) * pl.strips + s) * G + l) * AL_SNAPW);
            const uint4 x0 = sp[0], x1 = sp[1], x2 = sp[2], x3 = sp[3], x4 = sp[4], x5 = sp[5];
            rt[0] = (int)x0.x; rt[1] = (int)x0.y; rt[2] = (int)x0.z; rt[3] = (int)x0.w; rt[4] = (int)x1.x; rt[5] = (int)x1.y; rt[6] = (int)x1.z; rt[7] = (int)x1.w;
            rt[8] = (int)x2.x; rt[9] = (int)x2.y; hold[0] = (int)x2.z; hold[1] = (int)x2.w; hold[2] = (int)x3.x; hold[3] = (int)x3.y; hold[4] = (int)x3.z; hold[5] = (int)x3.w;
            hold[6] = (int)x4.x; hold[7] = (int)x4.y; hold[8] = (int)x4.z; hold[9] = (int)x4.w; diag0 = (int)x5.x; dn_out = (int)x5.y;
            h_out = hold[R - 1];
            const int jb = tbeg - l;
            if (jb >= 1 && jb <= m_eff) { int b = bp.at(jb - 1); if (b >= 5) { bad = 1; b = 4; } b_out = b * (BST * 4); }
        }
        int qdn, qh, qb, ndn = 0, nh = 0, nb = 0;
        // REBASE: the snapshot's keys are relative to the strip's base of block c; the row above, block by block, to the bases of the strip above
        long long Bt = 0;
        if (REBASE && gact && c > 0) Bt = bases[pl.rowi_off + (int64_t)s * pl.s_pitch + c];
        const int r0i = REBASE ? rbase_const((long long)kp.o4 + TI, Bt) : 0;
        auto boundary = [&](int cc, int &odn, int &oh, int &ob) {
            if (s == 0) {
                const int M3 = NEG4 + 3, I2 = REBASE ? r0i : kp.o4 + cc * E4 + TI - RB * cc, D1 = NEG4 + TD;
                oh = max3i(M3, I2, D1);
                odn = max3i(M3 + OE4, I2 + OE4, D1 + E4) - RB;
            } else if (cc >= 1 && cc <= m_eff) {
                const int2 v = rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + cc];
                odn = v.x; oh = v.y;
                if (REBASE) { const int q = (cc + 14) / CKA; const int dd = rbase_delta(bases[pl.rowi_off + (int64_t)(s - 1) * pl.s_pitch + q], Bt); odn += dd; oh += dd; }
            } else { odn = 0; oh = 0; }
            ob = (cc >= 1 && cc <= m_eff) ? bp.raw(cc - 1) : 0; // RAW base, see al_sweep_kernel
        };
        auto base_off = [&](int raw, int c) { int b = (c >= 1 && c <= m_eff) ? bp.value(raw, c - 1) : 0; if (b >= 5) { bad = 1; b = 4; } return b * (BST * 4); }; // LDS byte offset of the base's profile plane (c: the column the raw base was loaded for)
        boundary(tbeg + l + 1, qdn, qh, qb);
        qb = base_off(qb, tbeg + l + 1);
        // profile entries one step ahead (software pipeline over the LDS round trip, see cl_sweep_kernel)
        int wq[LW], pb_cur;
        auto fetch = [&](int pbv, int *w) {
            const int *pw = reinterpret_cast<const int *>(prof_lane + pbv);
#pragma unroll
            for (int k = 0; k < LW; k++) w[k] = pw[k];
        };
        pb_cur = dpp_shr1(qb, b_out);
        qb = dpp_shl1(qb, qb);
        fetch(pb_cur, wq);
        auto step = [&](const int t, auto chk, const bool take, const int nqv) {
            constexpr bool CHECK = decltype(chk)::value;
            const int up_dn = dpp_shr1(qdn, dn_out);
            const int up_h = dpp_shr1(qh, h_out);
            qdn = dpp_shl1(qdn, qdn);
            qh = dpp_shl1(qh, qh);
            if (take) qb = nqv; // (last step of a block: the base queue of the next one takes over)
            const int pb_next = dpp_shr1(qb, pb_cur);
            qb = dpp_shl1(qb, qb);
            int wn[LW];
            fetch(pb_next, wn);
            asm volatile("" ::: "memory"); // the reads stay HERE, ahead of the arithmetic
            const int j = t - l;
            const int *w = wq;
            if (!CHECK || (j >= 1 && j <= m_eff)) {
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
        for (int b = 0; b < nblk_max; b++) {
            const int t0 = tbeg + 16 * b; // per pair
            boundary(t0 + 16 + l + 1, ndn, nh, nb);
            if (__all(!gact || b >= nblk || (t0 >= 16 && t0 + 16 <= m_eff))) {
#pragma unroll
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::false_type{}, u == 15, nb); }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, u == 15, nb); }
            }
            qdn = ndn; qh = nh;
            if (gact && b < nblk) {
                const int miss = (t0 + 16 - l) - m_eff;
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
#pragma unroll
                    for (int r = 0; r < R; r++) dirg[((b * 3 + k) * R + r) * G + l] = acc[k * R + r] >> sh;
                }
                // after the pair's own last block: lanes that have passed column m hold h(i, m) of their rows (blocks that only other
                // pairs of the wave still need run unchecked and may spoil the state afterwards)
                if (b == nblk - 1 && t0 + 16 - l >= m_eff) {
#pragma unroll
                    for (int r = 0; r < R; r++) hcolT[l * R + r] = hold[r];
                }
            }
        }
        __syncthreads();
        if (l == 0 && gact) {
            int i = wi, j = wj, k = wk;
            if (pend) { const int kn = 3 - (hcolT[i - 1 - s * H] & 3); if (pend == 2) { q1n++; q1c += (kn != k); } k = kn; pend = 0; } // (the tile of (i, m) has just been filled: its lane has passed m)
            while (true) {
                if (i == 0 || j == 0) { wdone = 1; break; }
                const int i0 = i - 1 - s * H;
                if (i0 < 0) break; // left the strip through its top edge
                const int l2 = i0 / R, r2 = i0 - l2 * R;
                const int t1 = j + l2 - 1 - tbeg;
                if (t1 < tmin) break; // left the (usable part of the) tile through its skewed left edge
                const int pos = t1 & 15;
                const unsigned w = dirg[(((t1 >> 4) * 3 + k) * R + r2) * G + l2];
                int tag = (int)((w >> (2 * pos)) & 3u);
                if (tag == 0) { atomicOr(err, 2); wdone = 1; break; }
                if (k == 1) { // horizontal run inside this word, see traceback_kernel
                    int avail = min(pos + 1, j);
                    if (t1 < 16) avail = min(avail, pos - tmin + 1); // the tile's first word: its first `tmin` fields are not usable
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
                    if (j < pl.m) { // = the M-plane field of (i+1, j+1): the row the walk just left, at most one step past its cell
                        const int l3 = (i0) / R, r3 = i0 - l3 * R, t3 = (j + 1) + l3 - 1 - tbeg;
                        const unsigned w3 = dirg[(((t3 >> 4) * 3 + 0) * R + r3) * G + l3];
                        k = 3 - (int)((w3 >> (2 * (t3 & 15))) & 3u);
                    } else if (i - 1 - s * H >= 0 && pl.m + (i - 1 - s * H) / R - 1 - tbeg >= tmin) k = 3 - (hcolT[i - 1 - s * H] & 3);
                    else pend = 2; // row i belongs to the strip above, or its lane passed column m before this tile began: the next tile has it
                    if (pend != 2) { q1n++; q1c += (k != kt); }
                }
            }
            wi = i; wj = j; wk = k;
        }
    }
    if (l == 0 && valid) q1_report(q1n, q1c);
    if (l == 0 && valid && REBASE && mst) {
        mst->wi = wi; mst->wj = wj; mst->wk = wk; mst->pend = pend; mst->li = li; mst->cnt = cnt; mst->cur_run = cur_run; mst->cur_op = cur_op; mst->last_op = last_op;
        mst->done = pexit ? 0 : 1;
    }
    if (l == 0 && valid && !pexit) {
        // Step 4 (affineGap.go:135-139) -- quirk Q2 when the corner is not the origin
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
