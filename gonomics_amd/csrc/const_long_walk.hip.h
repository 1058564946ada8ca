// const_long_walk.hip.h -- the traceback of the snapshot path with SPECULATIVE tile re-fills: one pair per wave, its four lane groups re-fill
// the tile the walk is in AND the next tiles a diagonal path will enter, in the same instructions
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.5.
#pragma once
#include "const_long.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// cl_walk_kernel<.., NP = 1> (const_long.hip.h) gives every pair a wave of which 16 lanes work: per round they re-fill the tile the walk is in
// (strip s, steps (c*CK, step of the walk's cell]) into LDS, then lane 0 walks inside it until it leaves through the top (next strip) or the
// skewed left edge (previous snapshot interval).  A 20 kb read crosses ~200 tiles, one dependent round each: the stage is a latency
// chain (38 ms of the 193 ms C5 step at 1024 pairs, one wave per SIMD, nothing to hide behind).
// The other 48 lanes of the wave execute the re-fill's instructions anyway.  Here lane group g re-fills the tile the walk will be in g
// tiles from now IF its path keeps to the diagonal: the entry cell of tile g + 1 is where the diagonal through the entry cell of tile g
// leaves that tile (top edge: next strip; left edge: previous interval), and the re-fill runs CLW_MARGIN steps past the predicted cell.
// Lane 0 then walks tile after tile while the cell it stands on lies inside the next group's re-filled range (same strip, same interval,
// step <= the last re-filled one); the first miss ends the round.  A re-filled tile is exact wherever it lies (same recurrence from the
// exact snapshot), so a wrong guess costs a round, never a wrong bit.  ONT-style reads drift a few columns per strip: most rounds take
// all NS tiles.  LDS: NS direction tiles + the four-pair profile layout (one profile per group: the groups are in different strips).
// ------------------------------------------------------------------------------------------------------
constexpr int CLW_MARGIN = 32; // steps re-filled beyond the predicted entry cell of a speculative tile

template <bool P16, int CK, int NS>
__global__ __launch_bounds__(64) void cl_walk_spec_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                          const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                          const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                          KParams kp, TbParams tp, const int *__restrict__ rowbuf, const int *__restrict__ snap,
                                                          const int64_t *__restrict__ hfin, int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                          const int64_t *__restrict__ scr_off, gnx_cigar *__restrict__ scr, int *__restrict__ err) {
    using PC = ProfCfg<P16>;
    static_assert(NS >= 2 && NS <= 4, "tiles per round");
    constexpr int LW = PC::LW, BST = PC::BST, PTOT = PC::TOTAL;
    static_assert(CK % 16 == 0 && CK <= CKC, "snapshot spacing");
    constexpr int DIRG = (CK / 16) * R * G + 16; // LDS dwords of one tile (+16: neighbouring tiles start in different banks)
    __shared__ int lds[32 + PTOT + NS * DIRG];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    if (lane < 25) lds[lane] = kp.sc4[lane] - 2 * kp.g4 + 1; // pre-tagged diagonal candidate (tag 3), see fill_const_kernel
    int *prof = &lds[32 + PC::pair_off(g)];
    const char *prof_lane = reinterpret_cast<const char *>(prof + l * LW);
    unsigned *dir_all = reinterpret_cast<unsigned *>(&lds[32 + PTOT]);
    unsigned *dirg = dir_all + (g < NS ? g : 0) * DIRG;
    const int p = blockIdx.x;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const uint8_t *ap = a_buf + (valid ? a_start[p] : 0);
    BetaBytes bp;
    bp.init(b_buf, kp, valid ? b_start[p] : 0, valid ? pl.m : 0);
    const int64_t rb_pitch = (int64_t)pl.m + 1;
    const int po = pl.src;
    int bad = 0;
    // walker state (lane 0 of the wave)
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
    // where the plain diagonal through the cell (i, j) leaves the tile that cell is in: the entry cell of the next tile
    auto diag_exit = [&](int &i, int &j) {
        const int s = (i - 1) / H, a = i - 1 - s * H, lw = a / R;
        const int tbeg = ((j + lw - 1) / CK) * CK;
        int d = a + 1; // top edge: after a + 1 steps the row is s * H (the strip above)
        for (int q = lw; q >= 0; q--) { // the lanes the diagonal passes, bottom up: inside lane q the step index is j - d + q
            const int d_lo = max(0, a - q * R - (R - 1)), d_hi = a - q * R;
            const int dx = max(j - tbeg + q, d_lo); // first d with (j - d) + q - 1 - tbeg < 0
            if (dx <= d_hi) { d = dx; break; }
        }
        d = min(d, j); // (column 0 ends the walk)
        i -= d; j -= d;
    };

    while (true) {
        const int ci0 = __shfl(wi, 0, 64), cj0 = __shfl(wj, 0, 64), cdone = __shfl(wdone, 0, 64);
        if (cdone) break;
        // this group's tile: the walk's own (g = 0) or the g-th along the diagonal
        int ci = ci0, cj = cj0;
        bool gact = g < NS;
        for (int k = 0; k < g && gact; k++) { diag_exit(ci, cj); if (ci <= 0 || cj <= 0) gact = false; }
        const int s = gact ? (ci - 1) / H : 0;
        const int lw = gact ? (ci - 1 - s * H) / R : 0;
        const int tcell = gact ? cj + lw : 0;            // step of the (predicted) entry cell
        const int c = gact ? (tcell - 1) / CK : 0;
        const int tbeg = c * CK;
        const int tend = gact ? (g == 0 ? tcell : min(tcell + CLW_MARGIN, tbeg + CK)) : 0; // last step to re-fill
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
#pragma unroll
            for (int b = 0; b < 5; b++) {
#pragma unroll
                for (int k = 0; k < LW; k++) prof[b * BST + l * LW + k] = P16 ? ((lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16)) : lds[a5[k] + b];
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
        auto boundary = [&](int cc, int &ov, int &ob) {
            if (s == 0) ov = 2;
            else if (gact && cc >= 1 && cc <= m_eff) ov = rowbuf[pl.rowbuf_off + (int64_t)(s - 1) * rb_pitch + cc];
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
            const int t0 = tbeg + 16 * b; // per group
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
        // what each group re-filled: strip, first step, last step (-1: nothing)
        int ts[NS], tb[NS], te[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) { ts[k] = __shfl(s, 16 * k, 64); tb[k] = __shfl(tbeg, 16 * k, 64); te[k] = __shfl(gact ? tbeg + 16 * nblk : -1, 16 * k, 64); }
        if (lane == 0) {
            int i = wi, j = wj;
            for (int k = 0; k < NS && !wdone; k++) {
                if (k > 0) { // is the walk's cell inside the next group's tile?
                    if (i <= 0 || j <= 0) break; // (handled as done below)
                    const int s2 = (i - 1) / H, t2 = j + (i - 1 - s2 * H) / R;
                    if (te[k] < 0 || s2 != ts[k] || t2 <= tb[k] || t2 > te[k] || (t2 - 1) / CK != tb[k] / CK) break;
                }
                const unsigned *dk = dir_all + k * DIRG;
                const int sk = ts[k], tbk = tb[k];
                while (true) {
                    if (i == 0 || j == 0) { wdone = 1; break; }
                    const int i0 = i - 1 - sk * H;
                    if (i0 < 0) break; // left the strip through its top edge
                    const int l2 = i0 / R, r2 = i0 - l2 * R;
                    const int t1 = j + l2 - 1 - tbk;
                    if (t1 < 0) break; // left the tile through its (skewed) left edge
                    const int pos = t1 & 15;
                    const unsigned w = dk[((t1 >> 4) * R + r2) * G + l2];
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
            }
            if (i == 0 || j == 0) wdone = 1;
            wi = i; wj = j;
        }
    }
    if (lane == 0 && valid) {
        // Step 4 (constGap.go:59-63): the leading gap is appended only if the walk left through exactly one edge of its last
        // checkerboard; a corner exit appends nothing, even when it is not the origin (quirk Q2)
        const bool up_exit = (last_op != 1) && ((int64_t)wi % tp.ci == 0);
        const bool left_exit = (last_op != 2) && ((int64_t)wj % tp.cj == 0);
        if (!up_exit && left_exit) emit(2, wi);
        else if (up_exit && !left_exit) emit(1, wj);
        flush_run();
        nops[po] = cnt;
        score_out[po] = hfin[pl.hcol_off];
    }
    if (bad) atomicOr(err, 1);
}

} // namespace
