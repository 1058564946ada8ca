// traceback.hip.h -- traceback kernels over the packed direction matrix: checkerboard-walk emulation (Q1/Q2) and the gsw extension walks
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.
#pragma once
#include "gnx_common.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// Traceback.  One lane per pair.  WRITE=false counts CIGAR runs, WRITE=true emits them (reversed into
// alignment order) at ops[ops_off[p] ..).  ci/cj = checkerboard sizes (huge for the highMem modes).
// ------------------------------------------------------------------------------------------------------
struct TbParams {
    int64_t ci, cj;
    int64_t d00, ecol, gap_open, gap_extend; // unscaled, for the empty-sequence closed forms
    int affine;
};

// direction word of cell (i,j) (1-based) for plane k (affine: 0/1/2 = M/I/D; const: 0) and the field position
// of the cell inside it -- see the flush layout in the fill kernels.  Fields of lower columns sit at lower positions.
// LN x RW = lanes x rows per lane of the fill that wrote the matrix: 16 x 10 (fill_affine_kernel / fill_const_kernel) or 64 x 2 (lat_fill_kernel)
template <bool AFFINE, int LN = G, int RW = R>
__device__ __forceinline__ unsigned load_word(const uint4 *trace, const PairPlan &pl, int k, int i, int j, int &pos) {
    constexpr int HS = LN * RW;
    const int i0 = i - 1;
    const int s = i0 / HS, rem = i0 - s * HS;
    const int l = rem / RW, r = rem - l * RW;
    const int t1 = j + l - 1;
    const int w = t1 >> 4;
    pos = t1 & 15;
    const int d = AFFINE ? k * RW + r : r;
    constexpr int Q = ((AFFINE ? 3 : 1) * RW + 3) / 4;
    const unsigned *base = reinterpret_cast<const unsigned *>(trace + pl.trace_off + ((int64_t)(s * pl.words + w) * Q + (d >> 2)) * LN + l);
    return base[d & 3];
}

// COOP: one wave per pair (launches of few, long pairs).  Every lane runs the same walk; a diagonal run is taken 64 cells at a
// time -- lane t looks at cell (i-t, j-t) and the run goes on while the cells are M cells whose source is M -- instead of
// ~480 single-lane cycles per cell.  Quirk Q1 changes nothing inside such a run: the entry cell's argmax is M again.  A horizontal
// run is taken 64 words (1024 columns) at a time.
// SCR (with COOP, WRITE = false): single pass -- the runs are also written, in traceback order, to a scratch area of n + m + 2
// entries per pair (`ops` = scratch, `ops_off[p]` = the pair's scratch offset); reverse_runs_kernel puts them in place after the
// scan.  A latency-bound walk is not worth doing twice.
template <bool AFFINE, bool WRITE, bool COOP = false, bool SCR = false, int LN = G, int RW = R>
__global__ __launch_bounds__(64) void traceback_kernel(const PairPlan *__restrict__ plans, int n_pairs, const uint4 *__restrict__ trace,
                                                       const int *__restrict__ hcol, const unsigned *__restrict__ dcol, TbParams tp,
                                                       int64_t *__restrict__ score_out,
                                                       int64_t *__restrict__ nops, const int64_t *__restrict__ ops_off,
                                                       gnx_cigar *__restrict__ ops, int64_t ops_capacity, int *__restrict__ err) {
    const int p = COOP ? (int)blockIdx.x : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= n_pairs) return;
    const int lane = threadIdx.x & 63;
    const bool writer = !COOP || lane == 0;
    const PairPlan pl = plans[p];
    int i = pl.n, j = pl.m;
    int64_t score;
    int k;
    if (i > 0 && j > 0) {
        const int hc = hcol[pl.hcol_off + pl.n - 1];
        score = (int64_t)(hc >> 2);
        if (AFFINE) k = 3 - (hc & 3);
        else k = 0;
    } else { // highMem modes with an empty sequence: closed forms of row 0 / column 0
        if (AFFINE) {
            if (i == 0 && j == 0) { // tmt(0, gapOpen, D00)
                const int64_t a = 0, b = tp.gap_open, c = tp.d00;
                if (a >= b && a >= c) { score = a; k = 0; } else if (b >= c) { score = b; k = 1; } else { score = c; k = 2; }
            } else if (i == 0) { score = tp.gap_open + (int64_t)j * tp.gap_extend; k = 1; }
            else { score = tp.d00 + (int64_t)i * tp.ecol; k = 2; }
        } else { score = (int64_t)(i + j) * tp.gap_open; k = 0; }
    }
    const int po = pl.src; // output slot (== p except for sub-batches routed here by the fast path)
    if (!WRITE && writer) score_out[po] = score;

    int64_t cnt = 0;           // runs emitted so far (traceback order)
    int cur_op = -1;
    int64_t cur_run = 0;
    const int64_t total = WRITE ? nops[po] : 0;
    const int64_t obase = WRITE ? ops_off[po] : 0;
    const bool fits = WRITE ? (obase + total <= ops_capacity) : false;
    static_assert(!SCR || (COOP && !WRITE), "scratch mode is the single-pass cooperative walk");
    const int64_t sbase = SCR ? ops_off[p] : 0;
    auto flush_run = [&]() {
        if (cur_op >= 0) {
            if (SCR && writer) {
                gnx_cigar c; c.run_length = cur_run; c.op = (uint8_t)cur_op;
                for (int z = 0; z < 7; z++) c._pad[z] = 0;
                ops[sbase + cnt] = c;
            }
            if (WRITE && fits && writer) {
                gnx_cigar c; c.run_length = cur_run; c.op = (uint8_t)cur_op;
                for (int z = 0; z < 7; z++) c._pad[z] = 0;
                ops[obase + (total - 1 - cnt)] = c;
            }
            cnt++;
        }
    };
    auto emit = [&](int op, int64_t run) {
        if (op == cur_op) cur_run += run;
        else { flush_run(); cur_op = op; cur_run = run; }
    };

    // The checkerboard walk in global coordinates.  A tile is left through its top edge when the new row index
    // is a multiple of checkersize_i, through its left edge when the new column index is a multiple of
    // checkersize_j (affineGap.go:121-127).
    int64_t li = (i > 0) ? (int64_t)(i - 1) % tp.ci : 0; // tile-local row of the current cell
    int last_op = -1;
    const bool walked = (i > 0 && j > 0);
    bool have_w0 = false; // COOP: w0 / pos0 = plane-0 word and field position of the current cell (i, j), from the last diagonal look
    unsigned w0 = 0;
    int pos0 = 0;
    while (i > 0 && j > 0) {
        if (j == pl.m && (!AFFINE || k == 2)) {
            // Vertical run in the last column: the packed per-lane word holds the fields of R consecutive rows.
            const int i0 = i - 1, sl = i0 / RW, r = i0 - sl * RW; // sl = strip * lanes + lane
            const unsigned w = dcol[pl.dcol_off + sl];
            int tag = (int)((w >> (2 * r)) & 3u);
            if (tag == 0) { atomicOr(err, 2); break; }
            if (AFFINE || tag == 1) {
                int avail = min(r + 1, i);
                if (li + 1 < (int64_t)avail) avail = (int)(li + 1); // do not run past the tile's top edge (quirk Q1 applies there)
                unsigned x = w ^ 0x55555555u;                        // fields "from D" (tag 1) become 0
                if (r < 15) x &= (1u << (2 * r + 2)) - 1u;
                const int lowcut = r + 1 - avail;
                if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
                int steps;
                bool cont = false; // the walk is still in a D cell after the run
                if (x == 0) { steps = avail; cont = true; }
                else {
                    const int rnz = (31 - __clz((int)x)) >> 1;
                    tag = (int)((w >> (2 * rnz)) & 3u);
                    if (tag == 0) { atomicOr(err, 2); break; }
                    if (AFFINE) steps = r - rnz + 1; else { steps = r - rnz; cont = true; }
                }
                if (steps > 0) {
                    emit(2, steps); i -= steps; last_op = 2;
                    li -= steps;
                    const bool up_exit = li < 0;
                    if (up_exit) li += tp.ci;
                    if (AFFINE) {
                        k = cont ? 2 : 3 - tag;
                        if (up_exit && i > 0) k = 3 - (hcol[pl.hcol_off + i - 1] & 3); // quirk Q1, entry cell (i, m)
                    }
                    continue;
                }
            }
        }
        if (COOP && (!AFFINE || k == 0) && !have_w0) { // diagonal run, 64 cells per look (not when the last look already stopped at this cell)
            const int lim = min(min(i, j), 64);
            int f = 0, p2 = 0;
            unsigned wv = 0;
            if (lane < lim) { wv = load_word<AFFINE, LN, RW>(trace, pl, 0, i - lane, j - lane, p2); f = (int)((wv >> (2 * p2)) & 3u); }
            const unsigned long long stop = __ballot(!(lane < lim && f == 3));
            const int T = stop ? __ffsll((long long)stop) - 1 : 64;
            // the cell the run stops at -- lane T's -- is the next one the walk needs: keep its word, skip the next look
            const int tl = T < lim ? T : 0;
            have_w0 = T < lim; w0 = (unsigned)__builtin_amdgcn_readlane((int)wv, tl); pos0 = __builtin_amdgcn_readlane(p2, tl);
            if (T > 0) {
                emit(0, T); last_op = 0;
                i -= T; j -= T;
                li -= T; // T rows up, tile edges included (Q1 restarts in M, the state we are in)
                if (li < 0) { li %= tp.ci; if (li < 0) li += tp.ci; }
                continue;
            }
        }
        int pos;
        unsigned w;
        if (COOP && have_w0) { w = w0; pos = pos0; have_w0 = false; } // the diagonal look already fetched this cell's plane-0 word
        else w = load_word<AFFINE, LN, RW>(trace, pl, AFFINE ? k : 0, i, j, pos);
        int tag = (int)((w >> (2 * pos)) & 3u);
        const int op = AFFINE ? k : 3 - tag;
        if (tag == 0) { atomicOr(err, 2); break; } // impossible direction: the Go code would log.Fatalf
        if (op == 1) {
            // Horizontal run: every cell visited in state I emits one I and moves left; the walk stays in this word
            // while the fields read "came from I" (tag 2).  Count them with one xor + clz instead of 16 iterations.
            const int avail = min(pos + 1, j);            // fields of columns >= 1 at positions pos .. pos-avail+1
            unsigned x = w ^ 0xAAAAAAAAu;
            if (pos < 15) x &= (1u << (2 * pos + 2)) - 1u;
            const int lowcut = pos + 1 - avail;
            if (lowcut > 0) x &= ~((1u << (2 * lowcut)) - 1u);
            int steps;
            if (x == 0) { steps = avail; if (AFFINE) k = 1; }
            else {
                const int pnz = (31 - __clz((int)x)) >> 1;  // highest field that is not "from I"
                tag = (int)((w >> (2 * pnz)) & 3u);
                if (tag == 0) { atomicOr(err, 2); break; }
                if (AFFINE) { steps = pos - pnz + 1; k = 3 - tag; } // that cell is still in state I; its source decides the next state
                else steps = pos - pnz;                            // const gap: that cell is not an I cell
            }
            if (steps > 0) { emit(1, steps); j -= steps; last_op = 1; }
            if (COOP && x == 0 && j > 0) {
                // the run reached the low end of its word and goes on: lane t looks at the t-th word further left (16 columns each),
                // whole words of "came from I" are taken at once -- the 40 kb leading / trailing gaps of a read inside a long
                // window are ~40 looks instead of 2 500 dependent loads
                constexpr int HS = LN * RW, QQ = ((AFFINE ? 3 : 1) * RW + 3) / 4;
                const int i0 = i - 1, s2 = i0 / HS, rem = i0 - s2 * HS, l2 = rem / RW, r2 = rem - l2 * RW;
                const int t1 = j + l2 - 1;
                if ((t1 & 15) == 15) {
                    const int d = AFFINE ? RW + r2 : r2;
                    const int wq = (t1 >> 4) - lane;
                    const bool ok = wq >= 0 && wq * 16 >= l2; // every field of the word is a column >= 1
                    unsigned wv = 0;
                    if (ok) wv = reinterpret_cast<const unsigned *>(trace + pl.trace_off + ((int64_t)(s2 * pl.words + wq) * QQ + (d >> 2)) * LN + l2)[d & 3];
                    const unsigned long long stop = __ballot(!(ok && wv == 0xAAAAAAAAu));
                    const int T = stop ? __ffsll((long long)stop) - 1 : 64;
                    if (T > 0) { emit(1, 16 * (int64_t)T); j -= 16 * T; }
                }
            }
            continue;
        }
        emit(op, 1);
        last_op = op;
        bool up_exit = false;
        if (op != 1) { up_exit = (li == 0); li = up_exit ? tp.ci - 1 : li - 1; i--; }
        if (op != 2) j--;
        if (AFFINE) {
            k = 3 - tag;
            if (up_exit && i > 0 && j > 0) {
                // quirk Q1 (affineGap.go:305): entering a tile from below restarts in the argmax state of the entry cell
                int ht;
                if (j < pl.m) { int p2; ht = (int)((load_word<true, LN, RW>(trace, pl, 0, i + 1, j + 1, p2) >> (2 * p2)) & 3u); }
                else ht = hcol[pl.hcol_off + i - 1] & 3;
                k = 3 - ht;
            }
        }
    }
    // Step 4 (affineGap.go:135-139 / constGap.go:59-63) and the highMem border walks
    if (walked) {
        const bool up_exit = (last_op != 1) && ((int64_t)i % tp.ci == 0);
        const bool left_exit = (last_op != 2) && ((int64_t)j % tp.cj == 0);
        if (!up_exit && left_exit) emit(2, i);
        else if (up_exit && !left_exit) emit(1, j);
        // both: corner exit -> nothing (quirk Q2 when it is not the origin)
    } else { // empty sequence (highMem modes only)
        if (i == 0 && j > 0) emit(1, j);
        else if (j == 0 && i > 0) emit(2, i);
        else { cur_op = 0; cur_run = 0; } // Go: route == [{0 0}]
    }
    flush_run();
    if (!WRITE) { if (writer) nops[po] = cnt; }
    else if (!fits) atomicOr(err, 4);
}

// ------------------------------------------------------------------------------------------------------
// Traceback of the gsw seed-extension DPs (search.go:252-275, 298-320).  One lane per pair, runs in TRACEBACK order (the
// reference appends them that way; its callers reverse).  Ops use the GNX_COL_* codes (M/I/D).
//   LEFT : from (n, m) while the cell value is > 0.  Values are not stored: the walk rebuilds them from the final value
//          (an unclamped cell is its predecessor plus the score of the move; a clamped cell is 0 and ends the walk).
//   RIGHT: from the first row-major maximum (row scan of the keys the fill kernel left in hcol) back to (0, 0).
// ------------------------------------------------------------------------------------------------------
template <bool RIGHT, bool WRITE>
__global__ __launch_bounds__(64) void gsw_traceback_kernel(const PairPlan *__restrict__ plans, int n_pairs, const uint4 *__restrict__ trace,
                                                           const int *__restrict__ hcol,
                                                           const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                           const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start, KParams kp,
                                                           int64_t *__restrict__ score_out, int2 *__restrict__ endpos,
                                                           int64_t *__restrict__ nops, const int64_t *__restrict__ ops_off,
                                                           gnx_cigar *__restrict__ ops, int64_t ops_capacity, int *__restrict__ err) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const PairPlan pl = plans[p];
    const uint8_t *ap = a_buf + a_start[p];
    const uint8_t *bp = b_buf + b_start[p];
    int i = pl.n, j = pl.m;
    int cur = 0; // LEFT: value of the current cell
    if (RIGHT) {
        int bestv = 0, bi = 0, bj = 0;
        if (pl.m >= 1) {
            for (int r = 1; r <= pl.n; r++) {
                const int key = hcol[pl.hcol_off + r - 1];
                const int v = key >> 12;
                if (v > bestv) { bestv = v; bi = r; bj = 4095 - (key & 4095); }
            }
        }
        i = bi; j = bj;
        if (!WRITE) { score_out[p] = bestv; endpos[p] = make_int2(bi, bj); }
    } else {
        if (pl.n >= 1 && pl.m >= 1) cur = hcol[pl.hcol_off + pl.n - 1] >> 2;
        if (!WRITE) score_out[p] = cur;
    }
    const int64_t base = WRITE ? ops_off[p] : 0;
    int64_t cnt = 0, run = 0;
    int cur_op = -1;
    auto emit = [&](int op, int64_t len) {
        if (op == cur_op) { run += len; return; }
        if (cur_op >= 0) {
            if (WRITE) { if (base + cnt < ops_capacity) { gnx_cigar c; c.run_length = run; c.op = (uint8_t)cur_op; for (int z = 0; z < 7; z++) c._pad[z] = 0; ops[base + cnt] = c; } else atomicOr(err, 4); }
            cnt++;
        }
        cur_op = op; run = len;
    };
    while (RIGHT ? (i > 0 || j > 0) : (cur > 0)) {
        if (RIGHT && i == 0) { emit(GNX_COL_I, j); j = 0; break; } // trace[0][j] = 'I'
        if (RIGHT && j == 0) { emit(GNX_COL_D, i); i = 0; break; } // trace[i][0] = 'D'
        if (i < 1 || j < 1) { atomicOr(err, 2); break; }
        int pos;
        const unsigned w = load_word<false>(trace, pl, 0, i, j, pos);
        const int tag = (int)((w >> (2 * pos)) & 3u);
        if (tag == 3) { emit(GNX_COL_M, 1); if (!RIGHT) cur -= kp.sc4[min((int)ap[i - 1], 4) * 5 + min((int)bp[j - 1], 4)] >> 2; i--; j--; }
        else if (tag == 2) { emit(GNX_COL_I, 1); if (!RIGHT) cur -= kp.g4 >> 2; j--; }
        else if (tag == 1) { emit(GNX_COL_D, 1); if (!RIGHT) cur -= kp.g4 >> 2; i--; }
        else { atomicOr(err, 2); break; }
    }
    emit(-2, 0); // flush
    if (!WRITE) { nops[p] = cnt; if (!RIGHT) endpos[p] = make_int2(i, j); }
}

} // namespace
