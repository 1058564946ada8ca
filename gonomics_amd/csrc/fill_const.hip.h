// fill_const.hip.h -- constant-gap fill kernel and its GSW (graph-aligner seed extension) variants
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.
#pragma once
#include "gnx_common.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// Constant-gap fill (align/constGap.go:146-157 recurrence), same wavefront mapping as the affine kernel.
// Keys: diag+s -> tag 3, left+g -> tag 2, up+g -> tag 1.
// GSW == 0 (the align package's ConstGap): REBASED values V' = V - g*(i+j).  Both gap moves cost nothing, every border is 0,
// and the value is kept with tag 2 -- it IS the left candidate of the next column; the diagonal candidate is one add of the
// profile entry 4*(s - 2g) + 1 (tag 2 + 1 = 3), the upper candidate one add of -1 (tag 1): add, add, max3, alignbit, and_or =
// 5 instructions per cell instead of 6.  Every max compares candidates of one cell (same offset), so values and argmax tags
// are those of the plain recurrence; the row buffer between strips carries V' as well, hcol is un-rebased when it is stored.
// GSW != 0: plain values (the clamp at 0 / the running maximum need them): the stored value is the clean (tag-free) key, the
// three candidates are three VGPR adds (the profile holds 4*s+3, the penalties 4*g+2 / 4*g+1 live in VGPRs), one v_max3, one
// v_and and one v_alignbit per cell.
// ------------------------------------------------------------------------------------------------------
// GSW (the seed-extension DP of the graph aligner, "next" row N2, /root/reference/genomeGraph/search.go:234-321):
//   1 = LeftDynamicAln: zero borders, cell values clamped at 0 (the trace keeps its direction);
//   2 = RightDynamicAln: the ordinary borders plus, per row, the running maximum of (score << 12 | 4095 - column), i.e. the first
//       column of the row's best score; hcol receives that key instead of the last-column value.
// P16: every profile entry fits int16 (host check): packed profile, half the LDS (24 instead of 12 workgroups per CU)
template <bool MULTI, int GSW = 0, bool P16 = false>
__global__ __launch_bounds__(64) void fill_const_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                        const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                        const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                        KParams kp, uint4 *__restrict__ trace, int *__restrict__ hcol,
                                                        int2 *__restrict__ rowbuf, unsigned *__restrict__ dcol, int *__restrict__ err,
                                                        const int2 *__restrict__ strip_map = nullptr, int *__restrict__ strip_prog = nullptr) {
    // MULTI: one workgroup per (group of 4 pairs, strip), pipelined through the row buffer -- see fill_affine_kernel
    using PC = ProfCfg<P16>;
    constexpr int LW = PC::LW, BST = PC::BST, PTOT = PC::TOTAL;
    __shared__ int lds[32 + PTOT];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    constexpr bool REB = (GSW == 0);
    if (lane < 25) lds[lane] = REB ? kp.sc4[lane] - 2 * kp.g4 + 1 : kp.sc4[lane] + 3; // pre-tagged diagonal candidate
    int *prof = &lds[32 + PC::pair_off(g)];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    const bool piped = MULTI && strip_map != nullptr; // else one wave walks the strips of its 4 pairs one after the other
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
    int vGL, vGU;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(vGL), "=v"(vGU) : "s"(kp.g4 + 2), "s"(kp.g4 + 1));
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
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = P16 ? ((lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16)) : lds[a5[k] + b];
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < R; r++) { val[r] = REB ? 2 : (GSW == 1 ? 0 : (row0 + r + 1) * kp.g4); acc[r] = 0; } // column 0: i*gapPen (rebased: 0, tag 2)
        int best[R];
#pragma unroll
        for (int r = 0; r < R; r++) best[r] = 4095; // score 0: only a positive score replaces it (currMax starts at 0)
        int diag0 = REB ? 2 : (GSW == 1 ? 0 : row0 * kp.g4); // V(row above, 0)
        int v_out = 0, b_out = 0, sq_v = 0;
        int qv, qb, nv = 0, nb = 0;
        auto boundary = [&](int c, int &ov, int &ob) {
            if (!MULTI || s == 0) ov = REB ? 2 : (GSW == 1 ? 0 : c * kp.g4); // row 0: j*gapPen (rebased: 0, tag 2)
            else if (c >= 1 && c <= m_eff) ov = rb_load(&rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + c], piped).x;
            else ov = 0;
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
        if (MULTI && !piped && s > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        wait_rows(G);
        boundary(l + 1, qv, qb);
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
            const int up_v = dpp_shr1(qv, v_out);
            qv = dpp_shl1(qv, qv);
            if (take) qb = nqv; // (last step of a block: the base queue of the next one takes over)
            const int pb_next = dpp_shr1(qb, pb_cur);
            qb = dpp_shl1(qb, qb);
            int wn[LW];
            fetch(pb_next, wn);
            asm volatile("" ::: "memory"); // the reads stay HERE, ahead of the arithmetic
            const int j = t - l;
            const int *w = wq;
            if (!CHECK || (j >= 1 && j <= m_eff)) {
                int vd = diag0, vu = up_v;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int S4 = P16 ? ((r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff)) : w[r];
                    const int k = REB ? max3i(vd + S4, val[r], vu - 1) : max3i(vd + S4, val[r] + vGL, vu + vGU);
                    acc[r] = alignbit2((unsigned)k, acc[r]);
                    vd = val[r];
                    val[r] = REB ? ((k & ~3) | 2) : (k & ~3);
                    if (GSW == 1) val[r] = max(val[r], 0);
                    if (GSW == 2) best[r] = max(best[r], (int)((unsigned)val[r] << 10) + (4095 - j));
                    vu = val[r];
                }
                diag0 = up_v;
                v_out = vu;
            }
            if (MULTI) sq_v = dpp_shl1(v_out, sq_v);
#pragma unroll
            for (int k = 0; k < LW; k++) wq[k] = wn[k];
            pb_cur = pb_next;
        };

        for (int t0 = 0; t0 < Tend; t0 += 16) {
            wait_rows(t0 + 2 * G);
            boundary(t0 + 16 + l + 1, nv, nb);
            if (t0 >= 16 && t0 + 16 <= m_min) {
#pragma unroll
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::false_type{}, u == 15, nb); }
            } else {
#pragma unroll 1
                for (int u = 0; u < 16; u++) { if (u == 15) nb = base_off(nb, t0 + 16 + l + 1); step(t0 + u + 1, std::true_type{}, u == 15, nb); }
            }
            // consume the loads issued at the top of this block BEFORE the stores below are issued (exact wait, nothing newer in flight;
            // left to their first real use -- the DPP moves of the next block -- the wait becomes a vmcnt(0) behind those stores)
            asm volatile("" :: "v"(nv));
            qv = nv;
            const int w = t0 >> 4;
            if (gact && w < pl.words) {
                const int miss = (t0 + 16 - l) - m_eff;
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
                if (t0 + 16 > m_min) {
#pragma unroll
                    for (int d = 0; d < R; d++) acc[d] >>= sh;
                }
                uint4 *dst = trace + pl.trace_off + ((int64_t)(s * pl.words + w) * QC) * G + l;
                trace_store(&dst[0], acc[0], acc[1], acc[2], acc[3], piped);
                trace_store(&dst[G], acc[4], acc[5], acc[6], acc[7], piped);
                trace_store(&dst[2 * G], acc[8], acc[9], 0u, 0u, piped);
            }
            if (store_row) {
                const int c = t0 + l - 14;
                if (c >= 1 && c <= m_eff) rb_store(&rowbuf[pl.rowbuf_off + (int64_t)s * rb_pitch + c], sq_v, 0, piped);
            }
            if (piped && ((t0 + 16) & (RB_PUB - 1)) == 0) rb_publish(&strip_prog[bid], t0 + 1, lane);
        }
        if (gact && m_eff >= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) if (row0 + r < pl.n) hcol[pl.hcol_off + row0 + r] = GSW == 2 ? best[r] : (REB ? (val[r] & ~3) + kp.g4 * (row0 + r + 1 + m_eff) : val[r]);
            const int t0f = ((m_eff + l - 1) >> 4) << 4, missf = t0f + 16 - l - m_eff;
            unsigned dw = 0;
#pragma unroll
            for (int r = 0; r < R; r++) dw |= ((acc[r] >> (30 - 2 * missf)) & 3u) << (2 * r);
            dcol[pl.dcol_off + s * G + l] = dw;
        }
        if (piped) rb_publish(&strip_prog[bid], 0x7fffffff, lane);
        else if (MULTI) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    if (bad) atomicOr(err, 1);
}

} // namespace
