// fp_walk.hip.h -- fast-path staged traceback: plane following, window requests, straggler tiles, compaction
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.
#pragma once
#include "traceback.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// Fast path (alpha fits one strip, global affine, gapOpen <= 0): traceback by stages.
//   fp_walk: one lane per pair.  On row n in state I it follows the stored I-plane of row n (the long trailing
//   gap of a short read against a long chunk) a word at a time; anywhere else it needs full direction bits and
//   requests a re-fill of the <= fp_span(strips)+CKW-1 columns left of the current cell from the nearest column
//   checkpoint (window plan appended to a list), which fill_affine_kernel<.., WIN> computes with the normal
//   recording; the next fp_walk call continues inside that window.  Row 0 / column 0 end the walk (Step 4).
//   CIGAR runs are staged per pair in traceback order and reversed into place by fp_compact.
// Same checkerboard-walk emulation (Q1/Q2) as traceback_kernel.
// XP = walk of the transposed free-end-gap sweep (fp_sweep_kernel<.., true>): states stay k = 1 horizontal / 2 vertical, but
//   the tags of the two gap states are swapped (1 = horizontal I', 2 = vertical D') and a horizontal step is the reference's
//   ColD, a vertical one its ColI.
// ------------------------------------------------------------------------------------------------------
struct FpState {
    int32_t i, j, k, last_op;
    int32_t cur_op, cnt, status, slot; // status 0 = needs a window, 1 = done; slot = window slot of the last request
    int64_t cur_run;
    int64_t li;
    int32_t j_hi, jc_lo;
};

// CW = one WAVE per pair instead of one lane: every lane runs the same walk (lane 0 writes), and the three kinds of long runs are
//   taken with one cooperative look each -- 64 plane words (1024 columns) of an I-run on a stored plane or inside a tile, 64 cells
//   of a diagonal run inside a window -- instead of one dependent load per word / per cell.  100 000 waves of a few looks each
//   finish sooner than 1 563 waves of lanes that each walk alone.
template <bool FIRST, bool TILED = false, bool XP = false, bool CW = false>
__global__ __launch_bounds__(64) void fp_walk_kernel(const PairPlan *__restrict__ plans, const int *__restrict__ active, int n_active,
                                                     FpState *__restrict__ states, const int *__restrict__ hcol_fwd,
                                                     const unsigned *__restrict__ rowi, const unsigned *__restrict__ tail,
                                                     const PairPlan *__restrict__ wplans, const uint4 *__restrict__ wtrace,
                                                     const int *__restrict__ whcol, TbParams tp, gnx_cigar *__restrict__ stage,
                                                     int64_t *__restrict__ score_out, int64_t *__restrict__ nops,
                                                     int *__restrict__ next_active, int *__restrict__ next_count,
                                                     PairPlan *__restrict__ next_wplans, int *__restrict__ err, int p_base,
                                                     int *__restrict__ strag_active, int *__restrict__ strag_count, int force_strag,
                                                     const int2 *__restrict__ rowbuf, int spec, int wwords,
                                                     const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                     const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start, KParams kp) {
    // 4 x score table for the diagonal shortcuts (an index held in a VGPR: from LDS rather than from the kernel arguments)
    __shared__ int wsc[32];
    if (threadIdx.x < 25) wsc[threadIdx.x] = kp.sc4[threadIdx.x];
    __syncthreads();
    const int a = CW ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (a >= n_active) return;
    const int lane = threadIdx.x & 63;
    const bool writer = !CW || lane == 0;
    const int p = FIRST ? a + p_base : active[a];
    const PairPlan pl = plans[p];
    BetaSrc wbeta; // (the diagonal shortcuts read bases: FIRST walks, and window walks of reads with whole row blocks between two handed-down rows)
    const bool wb = FIRST || (!TILED && !XP && pl.strips >= 2);
    wbeta.init(b_buf, kp, wb ? b_start[pl.src] : 0, wb ? pl.m : 0);
    auto k_of = [](int tag) { return XP ? (tag == 3 ? 0 : tag) : 3 - tag; }; // state of a direction tag
    auto op_of = [](int k) { return XP ? (k == 0 ? 0 : 3 - k) : k; };         // CIGAR op of a step taken in state k
    constexpr unsigned IRUN = XP ? 0x55555555u : 0xAAAAAAAAu;                // 16 fields "horizontal gap extended"
    FpState st;
    PairPlan wp;
    if (FIRST) {
        const int hc = hcol_fwd[pl.hcol_off];
        if (writer) score_out[p] = (int64_t)(hc >> 2);
        st.i = pl.n; st.j = pl.m; st.k = k_of(hc & 3); st.last_op = -1;
        st.cur_op = -1; st.cnt = 0; st.status = 0; st.slot = -1; st.cur_run = 0;
        st.li = (int64_t)(pl.n - 1) % tp.ci;
        st.j_hi = 0; st.jc_lo = 0; // empty window
        wp = pl;
    } else {
        st = states[p];
        wp = wplans[TILED ? 0 : (int64_t)a * spec];
        if (TILED) { st.j_hi = 0; st.jc_lo = 0; }
    }
    int wrow = FIRST ? 0 : (int)wp.s_off; // row base of the current window: it holds the rows wrow + 1 .. wrow + wp.n of the pair
    // spec > 1 (reads of several row blocks): a request is answered with `spec` windows, plans [a*spec + k]: k = 0 the one asked for,
    // k >= 1 SPECULATIVE ones for the next row blocks up, placed where a near-diagonal path will enter them (fp_spec_margin).  The walk
    // takes window k + 1 when it leaves window k through the top and finds its cell inside; else it asks again (a round trip).
    int kcur = 0;
    // TILED: `a` indexes the straggler; its tiles c = 0.. are the plans [a*tiles_per + c] (tiles_per in wplans[0].rowi_off)
    const int tiles_per = TILED ? (int)wplans[0].rowi_off : 0;
    int i = st.i, j = st.j, k = st.k, last_op = st.last_op, cur_op = st.cur_op, cnt = st.cnt;
    int64_t cur_run = st.cur_run, li = st.li;
    // FIRST: the exact value of the walk's cell in its state -- h(n,m) to begin with, minus what every step taken so far contributed
    // (the steps of a first walk are gap cells on the stored planes and diagonal steps in the corner): see the shortcut after the loop
    int64_t val = FIRST ? (int64_t)(hcol_fwd[pl.hcol_off] >> 2) : 0;
    const int cap = fp_cap(pl.strips);
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
    bool done = false, last_win = false, no_look = false;
    // A re-fill that starts from a column checkpoint reproduces every VALUE, but the checkpoint carries no argmax tags
    // (the sweep computes them only on its plane rows), so the M- and I-plane fields of the re-fill's first column are
    // not usable: the walk uses a window / tile from its second column on (all of it when it starts at column 0).
    auto lo_ok = [](int jc_lo) { return jc_lo + (jc_lo > 0 ? 2 : 1); };
    // argmax tag of h(ii, jj) kept by the sweep for the 4 x 4 cells at the bottom right corner
    const unsigned *trec = tail + (int64_t)pl.hcol_off * FP_TAILW;
    const unsigned tailw = trec[0];
    // Sweeps with GNX_FP_EVENTS keep plane fields only for their tagged tail: steps > tst, i.e. the columns >= jplane of the plane rows.
    // Left of that a plane row has ONE number: the last step at which its gap could have been opened (trec[1 + d]); every cell right of
    // that step's column + 1 is a plain extension.
    const bool evm = trec[6] != 0;
    const int tst = (int)trec[5];
    const int jplane = evm ? (tst > 0 ? tst - (G8 - 2) : 1) : 1;
    bool diag_ok = false; // the plain diagonal from the walk's cell has already been scored and found to be the path (event cell, below)
    // score of the plain diagonal from (ii, jj) in state M up to row 0 plus the leading gap h(0, jj - ii) (FIRST walks: wbeta is bound)
    auto diag_score = [&](int ii, int jj) {
        const uint8_t *ap = a_buf + a_start[pl.src];
        int64_t P = (XP || jj == ii) ? 0 : tp.gap_open + tp.gap_extend * (int64_t)(jj - ii); // (XP: row 0 is free)
        int acc = 0; // (4 x the scores: int32 holds 20 480 of them)
#pragma unroll 8
        for (int t = 0; t < ii; t++) acc += wsc[min((int)ap[t], 4) * 5 + min(wbeta.at((jj - ii) + t), 4)];
        P += (int64_t)(acc >> 2);
        return P;
    };
    auto tail_ok = [&](int ii, int jj) { return ii >= 1 && jj >= 1 && pl.n - ii < FP_PLANES && pl.m - jj < 4; };
    auto tail_tag = [&](int ii, int jj) { return (tailw >> (8 * (pl.m - jj) + 2 * (pl.n - ii))) & 3u; };
    // Round 4, reads of several row blocks: the walk stands, in state M, on the BOTTOM row i of row block bq = (n - i) / 160 >= 1.  The sweep
    // handed that row down (to block bq - 1), so h(i, j) is in the row buffer; and the row above the block as well (handed down by block
    // bq + 1) -- or, for the top block, the row above is row 0, whose h(0, c) is the leading gap.  h(c) >= M(c) = h(c - diagonal) + s(c) in
    // every cell, so h(i, j) - h(i - L, j - L) >= the sum of the L substitution scores along the diagonal, with equality only if h = M in
    // every one of its cells -- tripleMaxTrace gives such ties to M (quirk Q1 re-reads the same argmax), so the path IS that diagonal: L
    // M steps, arriving in the state the handed-down row's argmax tag names (top block: at row 0, where the walk ends).  No window for a
    // block without an indel.  (The keys are rebased with e (i + j): 4 V = key + 4 e (i + j).)  Returns true when the walk moved.
    auto block_diag = [&]() -> bool {
        if (TILED || XP || k != 0 || pl.strips < 2 || i >= pl.n || i < 1 || (pl.n - i) % H != 0) return false;
        const int bq = (pl.n - i) / H;
        if (bq > pl.strips - 1) return false;
        const int hA = rowbuf[pl.rowbuf_off + (int64_t)(pl.strips - 1 - bq) * (pl.m + 1) + j].y; // row i: handed down by block bq
        if ((hA & 3) != 3) return false;
        const uint8_t *ap = a_buf + a_start[pl.src];
        const bool top = bq == pl.strips - 1;
        const int L = top ? i : H;
        if (top ? j < L : j <= L) return false;
        int far4; // 4 h(i - L, j - L) (32 bits hold every key: the DP range check of the entry points)
        int hB = 0;
        if (top) far4 = (j == L) ? 0 : kp.o4 + (j - L) * kp.e4;
        else {
            hB = rowbuf[pl.rowbuf_off + (int64_t)(pl.strips - 2 - bq) * (pl.m + 1) + (j - L)].y; // row i - 160: handed down by block bq + 1
            if ((hB & 3) == 0) return false;
            far4 = (hB & ~3) + kp.e4 * (i - L + j - L);
        }
        int sum = 0;
        if (CW) { // every lane runs the same walk: lane l sums the cells l, l + 64, ..., then the wave adds up
            for (int t = lane; t < L; t += 64) sum += wsc[min((int)ap[i - 1 - t], 4) * 5 + min(wbeta.at(j - 1 - t), 4)];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        } else {
#pragma unroll 2
            for (int t = 0; t < L; t++) sum += wsc[min((int)ap[i - 1 - t], 4) * 5 + min(wbeta.at(j - 1 - t), 4)];
        }
        if ((hA & ~3) + kp.e4 * (i + j) - far4 != sum) return false;
        emit(op_of(0), L); last_op = 0;
        i -= L; j -= L;
        li -= L;
        if (li < 0) { li %= tp.ci; if (li < 0) li += tp.ci; }
        if (!top) k = k_of(hB & 3);
        last_win = false;
        return true;
    };
    while (true) {
        if (i == 0 || j == 0) { done = true; break; }
        unsigned w;
        int pos;
        const bool on_plane = (k == 1) && (pl.n - i) < FP_PLANES && j >= jplane;
        const bool in_win = !on_plane && j >= lo_ok(st.jc_lo) && j <= st.j_hi && i > wrow && i <= wrow + wp.n;
        if (evm && !TILED && !on_plane && !in_win && k == 1 && (pl.n - i) < FP_PLANES) {
            // a plane row of an event sweep, left of the stored fields
            const int te = (int)trec[1 + (pl.n - i)];
            if (te < 0) { // the gap was never re-opened: plain extensions back to column 1, whose gap came out of column 0
                emit(op_of(1), j); last_op = 1; j = 0; k = 2;
                continue; // (j == 0 ends the walk: Step 4)
            }
            const int jc = te - (G8 - 1); // the column of that step: the direction of cell (i, jc + 1) is the first unknown one
            if (j <= jc) break;           // left of the last event nothing is known: a window
            const int steps = j - (jc + 1);
            if (steps > 0) { emit(op_of(1), steps); j -= steps; last_op = 1; last_win = false; if (FIRST) val -= tp.gap_extend * (int64_t)steps; }
            if (FIRST && jc >= i) {
                // Was the gap opened from M there?  Then M(i, jc) = I(i, jc + 1) - gapOpen - gapExtend.  If the plain diagonal from (i, jc)
                // scores exactly that, M(i, jc) >= it; and M(i, jc) + oe <= I(i, jc + 1) bounds it from above: equal, tripleMaxTrace gives
                // the tie to M, and the path is that diagonal (the argument of the shortcut below).  Otherwise: the window decides.
                const int64_t vM = val - tp.gap_open - tp.gap_extend;
                if (diag_score(i, jc) == vM) { emit(op_of(1), 1); last_op = 1; j = jc; k = 0; val = vM; diag_ok = true; }
            }
            break;
        }
        if (!FIRST && !in_win && !on_plane && block_diag()) continue; // (a whole row block, or the rest of the read, without a window)
        if (on_plane || in_win || (k == 0 && tail_ok(i - 1, j - 1))) last_win = in_win; // was the last step taken inside the window? (then the walk leaves it by walking through it)
        if (CW && in_win && k == 0 && !no_look) {
            // a diagonal run inside the window, 64 cells per look: lane t looks at the cell (i - t, j - t); the run goes on while the cells
            // are M cells whose source is M.  Quirk Q1 changes nothing inside such a run (the entry cell's argmax is M, the state we are in).
            const int lim = min(min(i - wrow, j - lo_ok(st.jc_lo) + 1), 64);
            int f = 0;
            if (lane < lim) { int p2; f = (int)((load_word<true>(wtrace, wp, 0, i - lane - wrow, j - lane - st.jc_lo, p2) >> (2 * p2)) & 3u); }
            const unsigned long long stop = __ballot(!(lane < lim && f == 3));
            const int T = stop ? __ffsll((long long)stop) - 1 : 64;
            if (T > 0) {
                emit(op_of(0), T); last_op = 0;
                i -= T; j -= T;
                li -= T;
                if (li < 0) { li %= tp.ci; if (li < 0) li += tp.ci; }
                continue;
            }
            no_look = true; // the cell itself is not part of such a run: one ordinary step, then look again
        }
        if (on_plane) { // stored I-plane of row n-d
            const int t1 = j + G8 - 1; // step at which the owner lane (the pair's last) was at column j
            w = rowi[pl.rowi_off + (int64_t)(pl.n - i) * pl.words + (t1 >> 4)];
            pos = t1 & 15;
        } else if (in_win) { // inside the usable part of the current window (one row block)
            w = load_word<true>(wtrace, wp, k, i - wrow, j - st.jc_lo, pos);
        } else if (k == 0 && tail_ok(i - 1, j - 1)) { // trM(i,j) = argmax of h(i-1,j-1): a diagonal step in the corner needs no window
            w = tail_tag(i - 1, j - 1); pos = 0;
        } else if (TILED) { // switch to the tile holding column j (all tiles of a straggler are filled)
            const int c = (j - 1) / FP_TILE;
            if (st.j_hi > 0 && !(i > wrow && i <= wrow + wp.n)) break; // left the tiles' row block through its top: the next block needs a window
            wp = wplans[(int64_t)a * tiles_per + c];
            wrow = (int)wp.s_off;
            st.jc_lo = wp.col_off; st.j_hi = st.jc_lo + wp.m; // tile c starts one checkpoint before column c*FP_TILE
            if (j > st.j_hi || j < lo_ok(st.jc_lo) || !(i > wrow && i <= wrow + wp.n)) { atomicOr(err, 2); done = true; break; }
            continue;
        } else if (!FIRST && kcur + 1 < spec) { // the next speculative window: is the walk's cell inside it?
            const PairPlan nx = wplans[(int64_t)a * spec + kcur + 1];
            kcur++;
            if (nx.strips > 0 && i > (int)nx.s_off && i <= (int)nx.s_off + nx.n && j >= lo_ok(nx.col_off) && j <= nx.col_off + nx.m) {
                wp = nx; wrow = (int)nx.s_off; st.jc_lo = nx.col_off; st.j_hi = nx.col_off + nx.m;
                continue;
            }
            break; // mispredicted (or the row block was used up sideways): ask again
        } else break; // needs a (new) window
        int tag = (int)((w >> (2 * pos)) & 3u);
        if (tag == 0) { atomicOr(err, 2); done = true; break; }
        if (k == 1) {
            int avail = min(pos + 1, j);
            if (!on_plane) avail = min(avail, j - lo_ok(st.jc_lo) + 1); // do not run past the window's usable left edge
            else if (evm) avail = min(avail, j - jplane + 1);            // ... or past the first stored field of an event sweep
            unsigned x = w ^ IRUN;
            if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
            const int lowcut = pos + 1 - avail;
            if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
            int steps;
            if (x == 0) steps = avail;
            else {
                const int pnz = (31 - __clz((int)x)) >> 1;
                tag = (int)((w >> (2 * pnz)) & 3u);
                if (tag == 0) { atomicOr(err, 2); done = true; break; }
                steps = pos - pnz + 1;
                k = k_of(tag);
                if (FIRST && !(XP && i == pl.n)) val -= tp.gap_open; // the last of these cells opened the gap (XP: steps along the last row are free)
            }
            emit(op_of(1), steps); j -= steps; last_op = 1;
            if (FIRST && !(XP && i == pl.n)) val -= tp.gap_extend * (int64_t)steps;
            if (CW && TILED && !on_plane && x == 0 && steps == pos + 1 && j >= lo_ok(st.jc_lo)) {
                // a straggler's long gap on a row without a stored plane: 64 words of its tile per look (lane t takes the t-th word
                // further left), while they are all-I and lie inside the usable part of the tile
                const int i0 = i - 1 - wrow, s2 = 0, rem2 = i0, l2 = rem2 / R, r2 = rem2 - l2 * R, d = R + r2;
                const int t1 = (j - st.jc_lo) + l2 - 1;
                if ((t1 & 15) == 15) {
                    const int wq = (t1 >> 4) - lane;
                    const bool ok = wq >= 0 && 16 * wq - l2 + 1 >= lo_ok(st.jc_lo) - st.jc_lo; // all 16 fields are usable columns
                    unsigned qv = 0;
                    if (ok) qv = reinterpret_cast<const unsigned *>(wtrace + wp.trace_off + ((int64_t)(s2 * wp.words + wq) * QA + (d >> 2)) * G + l2)[d & 3];
                    const unsigned long long stop = __ballot(!(ok && qv == IRUN));
                    const int T = stop ? __ffsll((long long)stop) - 1 : 64;
                    if (T > 0) { emit(op_of(1), 16 * (int64_t)T); j -= 16 * T; }
                }
            }
            if (on_plane && x == 0 && steps == pos + 1) {
                // the run continues below field 0 of this word: take whole 16-column words while they are all-I, four loads in
                // flight (a 10 kb trailing gap is 600 dependent loads otherwise).  Only on the stored planes, where the lanes of
                // a wave are in this state together; inside windows / tiles the extra control flow costs more than it saves.
                const unsigned *wbase = rowi + pl.rowi_off + (int64_t)(pl.n - i) * pl.words;
                int wi = ((j + steps + G8 - 1) >> 4) - 1;
                bool more = true;
                while (CW && more && wi >= 0 && j >= 16 && (!evm || j - 15 >= jplane)) { // 64 words per look
                    const int lim = min(min(wi + 1, evm ? (j - jplane + 1) >> 4 : j >> 4), 64);
                    const unsigned qv = lane < lim ? wbase[wi - lane] : 0u;
                    const unsigned long long stop = __ballot(!(lane < lim && qv == IRUN));
                    const int T = stop ? __ffsll((long long)stop) - 1 : 64;
                    if (T > 0) { emit(op_of(1), 16 * (int64_t)T); j -= 16 * T; wi -= T; if (FIRST && !(XP && i == pl.n)) val -= tp.gap_extend * 16 * (int64_t)T; }
                    more = (T == 64);
                }
                while (!CW && more && wi >= 0 && j >= 16 && (!evm || j - 15 >= jplane)) {
                    unsigned q[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) q[u] = (wi - u >= 0) ? wbase[wi - u] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (more && q[u] == IRUN && j >= 16 && (!evm || j - 15 >= jplane)) { emit(op_of(1), 16); j -= 16; wi--; if (FIRST && !(XP && i == pl.n)) val -= tp.gap_extend * 16; }
                        else more = false;
                    }
                }
            }
            continue;
        }
        if (FIRST && k == 0) val -= (int64_t)(kp.sc4[min((int)a_buf[a_start[pl.src] + i - 1], 4) * 5 + min(wbeta.at(j - 1), 4)] >> 2); // a diagonal step: M(i,j) = h(i-1,j-1) + s
        emit(op_of(k), 1);
        last_op = k;
        bool up_exit = false;
        if (k != 1) { up_exit = (li == 0); li = up_exit ? tp.ci - 1 : li - 1; i--; }
        if (k != 2) j--;
        k = k_of(tag);
        no_look = false;
        if (up_exit && i > 0 && j > 0) { // quirk Q1: restart in the argmax state of the entry cell (i, j)
            int ht;
            if (tail_ok(i, j)) ht = (int)tail_tag(i, j);
            else if (j <= st.jc_lo) { atomicOr(err, 2); done = true; break; } // cannot happen: column jc_lo + 1 is never walked
            else if (i == wrow) ht = rowbuf[wp.rowbuf_off + j].y & 3; // the entry cell is in the bottom row of the block above: the key that block handed down
            else if (j < st.j_hi) { int p2; ht = (int)((load_word<true>(wtrace, wp, 0, i + 1 - wrow, j + 1 - st.jc_lo, p2) >> (2 * p2)) & 3u); }
            else ht = whcol[wp.hcol_off + i - wrow - 1] & 3;
            k = k_of(ht);
        }
    }
    if (FIRST && !done && k == 0 && i >= 1 && j >= i) {
        // The walk stands in state M at (i, j) and `val` is M(i, j) exactly.  If the plain diagonal from here up to row 0 scores exactly
        // that, the traceback IS that diagonal: every prefix of an optimal alignment is optimal for its end cell, so M(t, c) equals the
        // diagonal's prefix score all the way up, hence h(t-1, c-1) = M(t-1, c-1), and tripleMaxTrace gives a tie to M (quirk Q1 re-reads
        // the same argmax).  No window needed: the reads without an indel (three quarters of the headline batch) skip the re-fill
        // stages altogether.  A read whose best alignment is anything else fails the equality (its optimum is higher) and asks for
        // its window as before.  (XP, the transposed AffineGapLocal: the same with a free row 0 and free steps along the last row.)
        const int64_t P = diag_ok ? val : diag_score(i, j); // (h(0, j - i), the leading gap, + the i diagonal steps)
        if (P == val) {
            emit(op_of(0), i); last_op = 0;
            li -= i;
            if (li < 0) { li %= tp.ci; if (li < 0) li += tp.ci; }
            j -= i; i = 0;
            done = true;
        }
    }
    if (FIRST && !XP && !done && k == 0 && pl.strips >= 2 && i > pl.n - H && i <= pl.n) {
        // ... and the first walk of a read of several row blocks: `val` is M(i, j) exactly and the row above the bottom block was handed
        // down: the same test takes the walk to the top of the bottom block, and block_diag on from there
        const int L = i - (pl.n - H);
        if (j > L) {
            const int hB = rowbuf[pl.rowbuf_off + (int64_t)(pl.strips - 2) * (pl.m + 1) + (j - L)].y;
            if ((hB & 3) != 0) {
                const uint8_t *ap = a_buf + a_start[pl.src];
                int sum = 0;
#pragma unroll 8
                for (int t = 0; t < L; t++) sum += wsc[min((int)ap[i - 1 - t], 4) * 5 + min(wbeta.at(j - 1 - t), 4)];
                if (4 * val - ((int64_t)(hB & ~3) + (int64_t)kp.e4 * (i - L + j - L)) == (int64_t)sum) {
                    emit(op_of(0), L); last_op = 0;
                    i -= L; j -= L;
                    li -= L;
                    if (li < 0) { li %= tp.ci; if (li < 0) li += tp.ci; }
                    k = k_of(hB & 3);
                    while (i > 0 && j > 0 && block_diag()) {}
                    if (i == 0 || j == 0) done = true;
                }
            }
        }
    }
    if (done) {
        // Step 4 (affineGap.go:135-139)
        const bool up_exit = (last_op != 1) && ((int64_t)i % tp.ci == 0);
        const bool left_exit = (last_op != 2) && ((int64_t)j % tp.cj == 0);
        if (!up_exit && left_exit) emit(op_of(2), i);
        else if (up_exit && !left_exit) emit(op_of(1), j);
        flush_run();
        cur_op = -1;
        if (writer) nops[p] = cnt;
        if (cnt > cap) atomicOr(err, 8);
        st.status = 1;
    } else {
        // The walk needs direction bits of the row block holding row i (blocks are counted from the bottom like the sweep's: block b
        // = rows n - 160 (b + 1) + 1 .. n - 160 b, the top block has what is left).  Normally a WINDOW (jc_lo, j] of that block: at
        // least FP_SPAN wide, starting on a checkpoint column (or column 0).  A walk that used up a window without leaving its row
        // block is in a long gap on a row without a stored plane: a STRAGGLER, all remaining columns of its block are re-filled as tiles.
        const int b = (pl.n - i) / H, rb = max(0, pl.n - H * (b + 1)), rows = pl.n - H * b - rb;
        // (a read of one row block: any second request -- a round for a handful of pairs costs more than their tiles)
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
            if (writer) next_active[slot] = p;
            st.slot = slot;
            for (int k2 = 0; k2 < spec; k2++) {
                const int bk = b + k2, rbk = max(0, pl.n - H * (bk + 1)), rowsk = pl.n - H * bk - rbk; // (rbk, rowsk: block bk of the pair)
                // window k2 ends where the path will enter block bk: (i - rb) + 160 (k2 - 1) rows further up, about as many columns to the left
                const int D = fp_spec_margin(k2);
                const int je = k2 == 0 ? j : j - ((i - rb) + H * (k2 - 1)) + D;
                int jc = je - FP_SPAN - 2 * D;
                jc = jc <= 0 ? 0 : (jc / CKW) * CKW;
                if (k2 == 0) { st.j_hi = j; st.jc_lo = jc; }
                PairPlan q;
                const bool real = bk < pl.strips && je >= 1;
                q.n = real ? rowsk : 0; q.m = real ? je - jc : 0; q.words = (q.m + 15 + 15) / 16; q.strips = real ? 1 : 0;
                const int64_t x = (int64_t)slot * spec + k2;
                q.trace_off = x * wwords * QA * G; q.hcol_off = x * H;
                q.rowbuf_off = pl.rowbuf_off + (int64_t)(pl.strips - 2 - bk) * (pl.m + 1); // the row the block above handed down (unused for the top block)
                q.dcol_off = x * G;
                q.src = pl.src; q.col_off = jc; q.ckpt_off = pl.ckpt_off; q.rowi_off = 0; q.s_off = rbk; q.s_pitch = pl.n;
                if (writer) next_wplans[x] = q;
            }
        }
    }
    st.i = i; st.j = j; st.k = k; st.last_op = last_op; st.cur_op = cur_op; st.cnt = cnt; st.cur_run = cur_run; st.li = li;
    if (writer) states[p] = st;
}

// stragglers (the path keeps needing windows, e.g. a long gap on a row without a stored plane): every remaining
// column of such a pair is re-filled as independent FP_TILE-column tiles from the column checkpoints -- one launch,
// a tile deep instead of a matrix deep -- and fp_walk_kernel<false, true> finishes the walk through them.
__global__ __launch_bounds__(256) void fp_straggler_plans_kernel(const PairPlan *__restrict__ plans, const int *__restrict__ active, int n_active,
                                                                  int tiles_per, const FpState *__restrict__ states, PairPlan *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_active * tiles_per) return;
    const int a = x / tiles_per, c = x - a * tiles_per;
    const int p = active[a];
    const PairPlan pl = plans[p];
    const int j_cur = states[p].j, i_cur = states[p].i;
    const int b = (pl.n - i_cur) / H, rb = max(0, pl.n - H * (b + 1)), rows = pl.n - H * b - rb; // the row block of the walk's current row
    PairPlan q = pl;
    const int lo = c * FP_TILE, lo2 = max(0, lo - CKW); // start one checkpoint early: the first re-filled column has no usable tags
    q.n = rows;
    q.m = (j_cur > lo) ? min(FP_TILE, j_cur - lo) + (lo - lo2) : 0;
    q.words = (q.m + 15 + 15) / 16; q.strips = q.m > 0 ? 1 : 0;
    q.trace_off = (int64_t)x * FP_TWORDS * QA * G; q.hcol_off = (int64_t)x * H;
    q.rowbuf_off = pl.rowbuf_off + (int64_t)(pl.strips - 2 - b) * (pl.m + 1);
    q.dcol_off = (int64_t)x * G;
    q.src = pl.src; q.col_off = lo2;
    q.rowi_off = (x == 0) ? tiles_per : 0; // plan 0 carries tiles_per for the walk kernel
    q.s_off = rb; q.s_pitch = pl.n;
    out[x] = q;
}

__global__ __launch_bounds__(256) void fp_compact_kernel(int n_pairs, const FpState *__restrict__ states, const gnx_cigar *__restrict__ stage, const int64_t *__restrict__ nops,
                                                          const int64_t *__restrict__ ops_off, gnx_cigar *__restrict__ ops, int64_t ops_capacity,
                                                          int *__restrict__ err, int cap) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const int64_t cnt = nops[p], base = ops_off[p];
    if (base + cnt > ops_capacity) { atomicOr(err, 4); return; }
    const int64_t m = cnt < cap ? cnt : cap;
    for (int64_t x = 0; x < m; x++) ops[base + (cnt - 1 - x)] = stage[(int64_t)p * cap + x];
}

// Pairs whose CIGAR has more runs than the staging area holds (fp_cap(strips)) are aligned again on the general path -- only they, not
// the batch: gather their window starts, and afterwards put their scores / runs where the compaction left the gaps.
__global__ __launch_bounds__(256) void fp_redo_gather_kernel(const int *__restrict__ idx, int n, const int64_t *__restrict__ as, const int64_t *__restrict__ bs,
                                                             int64_t *__restrict__ oas, int64_t *__restrict__ obs) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { oas[k] = as[idx[k]]; obs[k] = bs[idx[k]]; }
}
__global__ __launch_bounds__(256) void fp_redo_scatter_kernel(const int *__restrict__ idx, int n, const int64_t *__restrict__ sub_score,
                                                              const int64_t *__restrict__ sub_off, const gnx_cigar *__restrict__ sub_ops,
                                                              int64_t *__restrict__ score, const int64_t *__restrict__ ops_off, gnx_cigar *__restrict__ ops,
                                                              int *__restrict__ err) {
    const int k = blockIdx.x; // one block per pair
    if (k >= n) return;
    const int p = idx[k];
    const int64_t len = sub_off[k + 1] - sub_off[k], base = ops_off[p];
    if (len != ops_off[p + 1] - base) { if (threadIdx.x == 0) atomicOr(err, 2); return; } // both paths count the same runs
    if (threadIdx.x == 0) score[p] = sub_score[k];
    for (int64_t x = threadIdx.x; x < len; x += blockDim.x) ops[base + x] = sub_ops[sub_off[k] + x];
}

// A batch that mixes reads of <= 160 and of 161 .. 320 bases is aligned as two uniform sub-batches (each on its fast path) whose
// results are put back in input order: run counts and scores first, the runs after the scan of the counts.
__global__ __launch_bounds__(256) void mix_counts_kernel(const int *__restrict__ idx, int n, const int64_t *__restrict__ sub_off, const int64_t *__restrict__ sub_score,
                                                         int64_t *__restrict__ cnt, int64_t *__restrict__ score) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { const int p = idx[k]; cnt[p] = sub_off[k + 1] - sub_off[k]; score[p] = sub_score[k]; }
}
__global__ __launch_bounds__(256) void mix_copy_kernel(const int *__restrict__ idx, int n, const int64_t *__restrict__ sub_off, const gnx_cigar *__restrict__ sub_ops,
                                                       const int64_t *__restrict__ ops_off, gnx_cigar *__restrict__ ops, int64_t ops_capacity, int *__restrict__ err) {
    const int k = blockIdx.x; // one block per pair
    if (k >= n) return;
    const int p = idx[k];
    const int64_t len = sub_off[k + 1] - sub_off[k], base = ops_off[p];
    if (base + len > ops_capacity) { if (threadIdx.x == 0) atomicOr(err, 4); return; }
    for (int64_t x = threadIdx.x; x < len; x += blockDim.x) ops[base + x] = sub_ops[sub_off[k] + x];
}

} // namespace
