// fpc_walk.hip.h -- staged traceback of the constant-gap fast path (see fp_walk.hip.h for the affine one and for the scheme)
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.5.
#pragma once
#include "fp_walk.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// The walk of align.ConstGap (constGap.go:280-311 inside the checkerboard driver :13-68) over what fpc_sweep_kernel kept: stateless --
// the 2-bit field of a cell is its direction (3 diagonal, 2 left = ColI, 1 up = ColD).  On rows n .. n-3 it reads the stored planes
// (all directions, not only a gap state's), anywhere else it asks for a window of the row block holding its row
// (fill_const_kernel<.., WIN>); a walk that uses up a window inside one row block is a straggler and gets the block's remaining
// columns as tiles.  The checkerboards of the low-memory functions only matter at the end (Step 4, constGap.go:59-63: what the last
// move was and whether the walk stopped on a tile edge -- quirk Q2 included); there is no state to restart (no Q1).
// FIRST / TILED / CW, the request lists and force_strag: as in fp_walk_kernel.  FpState: k and li are unused.
// ------------------------------------------------------------------------------------------------------
template <bool FIRST, bool TILED = false, bool CW = false>
__global__ __launch_bounds__(64) void fpc_walk_kernel(const PairPlan *__restrict__ plans, const int *__restrict__ active, int n_active,
                                                      FpState *__restrict__ states, const int *__restrict__ hcol_fwd,
                                                      const unsigned *__restrict__ rowi,
                                                      const PairPlan *__restrict__ wplans, const uint4 *__restrict__ wtrace,
                                                      TbParams tp, gnx_cigar *__restrict__ stage,
                                                      int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                      int *__restrict__ next_active, int *__restrict__ next_count,
                                                      PairPlan *__restrict__ next_wplans, int *__restrict__ err, int p_base,
                                                      int *__restrict__ strag_active, int *__restrict__ strag_count, int force_strag) {
    const int a = CW ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (a >= n_active) return;
    const int lane = threadIdx.x & 63;
    const bool writer = !CW || lane == 0;
    const int p = FIRST ? a + p_base : active[a];
    const PairPlan pl = plans[p];
    FpState st;
    PairPlan wp;
    if (FIRST) {
        if (writer) score_out[p] = (int64_t)(hcol_fwd[pl.hcol_off] >> 2);
        st.i = pl.n; st.j = pl.m; st.k = 0; st.last_op = -1;
        st.cur_op = -1; st.cnt = 0; st.status = 0; st.slot = -1; st.cur_run = 0; st.li = 0;
        st.j_hi = 0; st.jc_lo = 0;
        wp = pl;
    } else {
        st = states[p];
        wp = wplans[TILED ? 0 : a];
        if (TILED) { st.j_hi = 0; st.jc_lo = 0; }
    }
    int wrow = FIRST ? 0 : (int)wp.s_off;
    const int tiles_per = TILED ? (int)wplans[0].rowi_off : 0;
    int i = st.i, j = st.j, last_op = st.last_op, cur_op = st.cur_op, cnt = st.cnt;
    int64_t cur_run = st.cur_run;
    const int cap = fpc_cap(pl.strips);
    gnx_cigar *stg = stage + (int64_t)p * cap;
    auto flush_run = [&]() {
        if (cur_op >= 0) {
            if (cnt < cap && writer) { gnx_cigar c; c.run_length = cur_run; c.op = (uint8_t)cur_op; for (int z = 0; z < 7; z++) c._pad[z] = 0; stg[cnt] = c; }
            cnt++;
        }
    };
    auto emit = [&](int op, int64_t run) {
        if (op == cur_op) cur_run += run;
        else { flush_run(); cur_op = op; cur_run = run; }
    };
    bool done = false, last_win = false;
    while (true) {
        if (i == 0 || j == 0) { done = true; break; }
        unsigned w;
        int pos;
        const bool on_plane = (pl.n - i) < FP_PLANES;
        const bool in_win = !on_plane && j > st.jc_lo && j <= st.j_hi && i > wrow && i <= wrow + wp.n;
        if (on_plane) {
            const int t1 = j + G8 - 1; // step at which the owner lane (the pair's last) was at column j
            w = rowi[pl.rowi_off + (int64_t)(pl.n - i) * pl.words + (t1 >> 4)];
            pos = t1 & 15;
            last_win = false;
        } else if (in_win) {
            w = load_word<false>(wtrace, wp, 0, i - wrow, j - st.jc_lo, pos);
            last_win = true;
        } else if (TILED) {
            if (st.j_hi > 0 && !(i > wrow && i <= wrow + wp.n)) break; // left the tiles' row block through its top
            const int c = (j - 1) / FP_TILE;
            wp = wplans[(int64_t)a * tiles_per + c];
            wrow = (int)wp.s_off;
            st.jc_lo = wp.col_off; st.j_hi = st.jc_lo + wp.m;
            if (j > st.j_hi || j <= st.jc_lo || !(i > wrow && i <= wrow + wp.n)) { atomicOr(err, 2); done = true; break; }
            continue;
        } else break; // needs a (new) window
        int tag = (int)((w >> (2 * pos)) & 3u);
        if (tag == 0) { atomicOr(err, 2); done = true; break; }
        const int op = 3 - tag;
        if (op == 1) { // horizontal run: the fields that read "left" (tag 2), counted with one xor + clz
            int avail = min(pos + 1, j);
            if (!on_plane) avail = min(avail, j - st.jc_lo); // do not run past the window's left edge
            unsigned x = w ^ 0xAAAAAAAAu;
            if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
            const int lowcut = pos + 1 - avail;
            if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
            int steps;
            if (x == 0) steps = avail;
            else {
                const int pnz = (31 - __clz((int)x)) >> 1; // highest field that is not "left": that cell is not part of the run
                if (((w >> (2 * pnz)) & 3u) == 0) { atomicOr(err, 2); done = true; break; }
                steps = pos - pnz;
            }
            if (steps > 0) { emit(1, steps); j -= steps; last_op = 1; }
            if (x == 0 && steps == pos + 1 && j >= 16) {
                // the run goes on below field 0 of this word: whole 16-column words while they are all "left"
                if (on_plane) {
                    const unsigned *wbase = rowi + pl.rowi_off + (int64_t)(pl.n - i) * pl.words;
                    int wi = ((j + steps + G8 - 1) >> 4) - 1;
                    bool more = true;
                    while (CW && more && wi >= 0 && j >= 16) { // 64 words per look
                        const int lim = min(min(wi + 1, j >> 4), 64);
                        const unsigned qv = lane < lim ? wbase[wi - lane] : 0u;
                        const unsigned long long stop = __ballot(!(lane < lim && qv == 0xAAAAAAAAu));
                        const int T = stop ? __ffsll((long long)stop) - 1 : 64;
                        if (T > 0) { emit(1, 16 * (int64_t)T); j -= 16 * T; wi -= T; }
                        more = (T == 64);
                    }
                    while (!CW && more && wi >= 0 && j >= 16) {
                        unsigned q[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) q[u] = (wi - u >= 0) ? wbase[wi - u] : 0u;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (more && q[u] == 0xAAAAAAAAu && j >= 16) { emit(1, 16); j -= 16; wi--; }
                            else more = false;
                        }
                    }
                } else if (CW && TILED && j > st.jc_lo) { // a straggler's long gap inside its tile: 64 tile words per look
                    const int i0 = i - 1 - wrow, l2 = i0 / R, r2 = i0 - l2 * R;
                    const int t1 = (j - st.jc_lo) + l2 - 1;
                    if ((t1 & 15) == 15) {
                        const int wq = (t1 >> 4) - lane;
                        const bool ok = wq >= 0 && 16 * wq - l2 + 1 >= 1; // all 16 fields are columns of the tile
                        unsigned qv = 0;
                        if (ok) qv = reinterpret_cast<const unsigned *>(wtrace + wp.trace_off + ((int64_t)wq * QC + (r2 >> 2)) * G + l2)[r2 & 3];
                        const unsigned long long stop = __ballot(!(ok && qv == 0xAAAAAAAAu));
                        const int T = stop ? __ffsll((long long)stop) - 1 : 64;
                        if (T > 0) { emit(1, 16 * (int64_t)T); j -= 16 * T; }
                    }
                }
            }
            continue;
        }
        emit(op, 1);
        last_op = op;
        i--;
        if (op == 0) j--;
    }
    if (done) {
        // Step 4 (constGap.go:59-63)
        const bool up_exit = (last_op != 1) && ((int64_t)i % tp.ci == 0);
        const bool left_exit = (last_op != 2) && ((int64_t)j % tp.cj == 0);
        if (!up_exit && left_exit) emit(2, i);
        else if (up_exit && !left_exit) emit(1, j);
        flush_run();
        cur_op = -1;
        if (writer) nops[p] = cnt;
        if (cnt > cap) atomicOr(err, 8);
        st.status = 1;
    } else {
        const int b = (pl.n - i) / H, rb = max(0, pl.n - H * (b + 1)), rows = pl.n - H * b - rb;
        const bool strag = force_strag || (!FIRST && (last_win || pl.strips == 1) && i > wrow && i <= wrow + wp.n);
        st.status = 0;
        if (strag && !TILED) {
            int slot = 0;
            if (writer) slot = atomicAdd(strag_count, 1);
            if (CW) slot = __builtin_amdgcn_readfirstlane(slot);
            if (writer) strag_active[slot] = p;
            st.j_hi = 0; st.jc_lo = 0; st.slot = slot;
        } else {
            int slot = 0;
            if (writer) slot = atomicAdd(next_count, 1);
            if (CW) slot = __builtin_amdgcn_readfirstlane(slot);
            int jc = j - FP_SPAN;
            jc = jc <= 0 ? 0 : (jc / CKW) * CKW;
            st.j_hi = j; st.jc_lo = jc; st.slot = slot;
            if (writer) next_active[slot] = p;
            PairPlan q;
            q.n = rows; q.m = j - jc; q.words = (q.m + 15 + 15) / 16; q.strips = 1;
            q.trace_off = (int64_t)slot * FP_WWORDS * QC * G; q.hcol_off = (int64_t)slot * H;
            q.rowbuf_off = pl.rowbuf_off + (int64_t)(pl.strips - 2 - b) * (pl.m + 1); // (ints) the row the block above handed down
            q.dcol_off = (int64_t)slot * G;
            q.src = pl.src; q.col_off = jc; q.ckpt_off = pl.ckpt_off; q.rowi_off = 0; q.s_off = rb; q.s_pitch = pl.n;
            if (writer) next_wplans[slot] = q;
        }
    }
    st.i = i; st.j = j; st.last_op = last_op; st.cur_op = cur_op; st.cnt = cnt; st.cur_run = cur_run;
    if (writer) states[p] = st;
}

// the stragglers' tiles (constant gap: QC direction words per tile word)
__global__ __launch_bounds__(256) void fpc_straggler_plans_kernel(const PairPlan *__restrict__ plans, const int *__restrict__ active, int n_active,
                                                                   int tiles_per, const FpState *__restrict__ states, PairPlan *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_active * tiles_per) return;
    const int a = x / tiles_per, c = x - a * tiles_per;
    const int p = active[a];
    const PairPlan pl = plans[p];
    const int j_cur = states[p].j, i_cur = states[p].i;
    const int b = (pl.n - i_cur) / H, rb = max(0, pl.n - H * (b + 1)), rows = pl.n - H * b - rb;
    PairPlan q = pl;
    const int lo = c * FP_TILE; // (a constant-gap re-fill is usable from its first column: no early start)
    q.n = rows;
    q.m = (j_cur > lo) ? min(FP_TILE, j_cur - lo) : 0;
    q.words = (q.m + 15 + 15) / 16; q.strips = q.m > 0 ? 1 : 0;
    q.trace_off = (int64_t)x * FP_TWORDS * QC * G; q.hcol_off = (int64_t)x * H;
    q.rowbuf_off = pl.rowbuf_off + (int64_t)(pl.strips - 2 - b) * (pl.m + 1);
    q.dcol_off = (int64_t)x * G;
    q.src = pl.src; q.col_off = lo;
    q.rowi_off = (x == 0) ? tiles_per : 0;
    q.s_off = rb; q.s_pitch = pl.n;
    out[x] = q;
}

} // namespace
