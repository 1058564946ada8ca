// fp_sweep.hip.h -- fast-path forward sweep (8 lanes x 19/20 rows per pair, rebased keys): the dominant kernel of the headline workload
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.
#pragma once
#include "gnx_common.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// Fast-path forward sweep: 8 lanes x RR rows per pair, 8 pairs per wave64 (global affine, gapOpen <= 0, n <= 8*RR).
// Same anti-diagonal wavefront as fill_affine_kernel, re-cut for the short-alpha shape:
//  * REBASED keys.  Every cell quantity V(i,j) in {M, I, D, h} is carried as V' = V - e*(i+j).  The recurrences keep
//    all their comparisons (each max compares candidates of the same cell, i.e. with the same offset) and become
//        M'(i,j) = h'(i-1,j-1) + (s - 2e)    I'(i,j+1) = max(h'(i,j) + o, I'(i,j))    D'(i+1,j) = max(h'(i,j) + o, D'(i,j))
//    (h-form, see fill_affine_kernel): both extensions cost nothing, both opens share h' + o, and every boundary
//    (row 0, column 0) is a constant: per cell  add, max3, add, max, max  = 16 issue cycles instead of 20.
//  * rows are RIGHT-ALIGNED: the pair's 8*RR slots end at row n, the first P = 8*RR - n slots are padding that reproduces
//    row 0 (profile entry -32768 so M never wins; I' = h' = o, D' = 2o are fixed points of the recurrences when o <= 0).
//    So rows n .. n-3 (whose I-planes are kept, FP_PLANES) are always the last four slots of the last lane: one kernel
//    for every n, pairs of different length mix freely, and only 4 of RR rows per lane pay the tag arithmetic.
//  * two pairs per 16-lane DPP row, the second one mirrored (lane 15 is its first lane), so that "value of the previous
//    lane of my pair" is row_shr:1 on banks 0-1 plus row_shl:1 on banks 2-3 and the lanes without a source keep the
//    boundary constant passed as `old`.
//  * int16 score profile (4*(s-2e)) read as 5 ds_read_b64 per step: an LDS read costs the issuing SIMD ~2 cycles + 2 per
//    returned dword (tools/lds_ubench.hip), so 20 rows cost 30 cycles for 8 pairs instead of 60 for 4.
// XP = the same sweep for AffineGapLocal(target = long, query = short), TRANSPOSED: rows = query, columns = target, so that the
//    short sequence is again the one held in lanes.  The transposed horizontal state I' is the reference's D and the vertical
//    D' its I, which changes two things:
//      * tie order M >= D' >= I': the tag constants of the two gap chains swap (I' carries 1, D' carries 2);
//      * the free end gaps (affineGap_highMem.go:188-210: D(i,0) = 0, and D(i+1,m) = tmt(M,I,D)(i,m) at no cost in the last
//        column) become a free row 0, h(0,j) = I'(0,j) = 0, and a free horizontal step in the LAST ROW,
//        I'(n,j+1) = h(n,j) with h's tag.  Rebased, row 0 is R(j) = -e*j: the first lane's boundary values grow by -e per
//        step (two extra adds); the last row opens its horizontal gap with -e instead of o (a per-lane constant in the last
//        slot: h' - e > I' always, so the shared max is the reference's single candidate); and the padding slots above row 1
//        are FAKE ROWS with profile entry -e, whose diagonal candidate R(j-1) - e = R(j) reproduces row 0 column by column
//        (their gap candidates R(j) + o and R(j-1) + o never exceed it), so the first real row sees exactly row 0 above it.
//    Needs gapExtend < 0 (strictly) besides gapOpen <= 0.
// Outputs (what fp_walk_kernel and the window re-fills of fill_affine_kernel<.., WIN> consume): column checkpoints
// {I'(i,j+1), h'(i,j)} of every row every CKW columns -- the rebased keys as they stand in the registers (the re-fill shifts them into its own frame).  The checkpoints
// cost 1.1 ms of the 29.8 ms sweep of the headline batch (28.7 ms without them); un-rebasing them in the sweep, 16-byte stores and
// branch-free stores all measure the same within 0.2 ms --, the I-plane words of rows n..n-3
// (word = step >> 4, field = step & 15 with step = j + 7), and h(n,m).
// ------------------------------------------------------------------------------------------------------
#ifndef GNX_FP_PUB
#define GNX_FP_PUB 32
#endif
#ifndef GNX_FP_PRIO
#define GNX_FP_PRIO 0
#endif
// GNX_FP_EVENTS (round 4): reads of ONE row block (ROLE 0, not the transposed form) keep no I-planes in the steady part of the sweep.
// The four plane rows then run the untagged 5-instruction cell like every other row, and each of them tracks only the LAST step at
// which its horizontal gap could have been opened:  event  <=>  h'(i,j) + o >= I'(i,j)  (values; all keys are multiples of 4 there).
// After the last event every cell of the row is a plain extension, so the walk can take a trailing gap in one stride down to the event
// cell and decide there (fp_walk_kernel: the diagonal shortcut proves "opened from M", else a window is re-filled as for any other cell).
// The tagged arithmetic (4 x (or, and_or, and_or, alignbit) = 16 of 121 instructions per step, executed by all 64 lanes for the sake of
// the last lane of each pair) is left to the TAIL of the sweep: the half blocks in which some lane of the wave has run out of columns
// (t0 + 7 > m_min), which hold the last columns of every pair -- corner tags, h(n, m) and the plane fields of the last ~8-22 columns.
// GNX_FP_EVBR: the event update sits behind a wave-uniform branch (taken only when a LAST lane of a pair has an event: a few per cent
// of the steps), so the steady step pays one v_cmp per plane row and nothing else.
#ifndef GNX_FP_EVENTS
#define GNX_FP_EVENTS 1
#endif
#ifndef GNX_FP_EVPIN
#define GNX_FP_EVPIN 1
#endif
#ifndef GNX_FP_EVBR
#define GNX_FP_EVBR 0
#endif
constexpr int G8 = 8;
constexpr int FP8_LW = 10;                 // dwords per lane per base (20 int16 entries)
// LDS layout of the int16 profile.  A ds_read_b64 serves 16 lanes per cycle = one DPP row = a DUO of pairs; lane lp reads dwords
// 10*lp + 2k, +1 of the plane of ITS base.  Conflict-free whatever the bases are when a plane stride is == 0 (mod 32) and the second
// pair of the duo sits 16 banks from the first (10*lp mod 32 = {0,10,20,30,8,18,28,6}: with their +1 neighbours 16 distinct banks, the
// other 16 are theirs + 16).  Compact form (round 3): the two pairs of a duo interleave plane by plane -- pair 0 in dwords [0, 80) of
// a 160-dword plane, pair 1 in [80, 160), 80 == 16 (mod 32) -- so no padding is left: 3200 dwords of profile, 32 of score table, 128
// of base rings = 13 440 B per wave, 11 waves per CU (LDS is handed out in granules of 1280 B on gfx950) where the padded form
// (96-dword planes, pair stride 496: 16 000 B) allowed 9.
#ifndef GNX_FP_LDS_COMPACT
#define GNX_FP_LDS_COMPACT 1
#endif
constexpr int FP8_BST = GNX_FP_LDS_COMPACT ? 160 : 96;   // dwords per base plane (of a duo / of a pair)
constexpr int FP8_PST = 5 * 96 + 16;                      // padded form: dwords per pair
constexpr int FP8_PROF = GNX_FP_LDS_COMPACT ? 4 * 5 * FP8_BST : 8 * FP8_PST; // dwords of profile per wave
#ifndef GNX_FP_RING_SKEW
#define GNX_FP_RING_SKEW 1
#endif
// compact form: 32 uint16 = 16 dwords per pair behind the profile.  A ds_read_u16 serves 32 lanes = 4 pairs per cycle and every pair reads
// the same 4-5 dword window of its ring, so the four rings of a half-wave must start in different banks (mod 32): with a plain stride
// of 16 dwords pairs g and g + 2 sat exactly 32 dwords apart -- a 2-way conflict on every read (SQ_LDS_BANK_CONFLICT 3.1e8 per
// 100 000 pairs in round 3).  Skewed: the third and fourth pair of a half-wave start 8 dwords later -> bases 0, 16, 40, 56
// = 0, 16, 8, 24 (mod 32); a half-wave's four rings take 72 dwords.
constexpr int FP8_RINGS = GNX_FP_LDS_COMPACT ? (GNX_FP_RING_SKEW ? 2 * 72 : 8 * 16) : 0;
__device__ __forceinline__ constexpr int fp8_ring_base(int g) { return GNX_FP_RING_SKEW ? (g >> 2) * 72 + (g & 3) * 16 + ((g >> 1) & 1) * 8 : g * 16; }
constexpr int FP8_LDS = 32 + FP8_PROF + FP8_RINGS;                             // dwords per wave (+ the hand-over staging of the levels kernel)

// lanes 0-7 of a DPP row: from lane-1; lanes 8-15 (mirrored pair): from lane+1; the first lane of each pair keeps oldv
__device__ __forceinline__ int dpp_prev8(int oldv, int src) {
    const int v = __builtin_amdgcn_update_dpp(oldv, src, DPP_ROW_SHR1, 0xf, 0x3, false);
    return __builtin_amdgcn_update_dpp(v, src, DPP_ROW_SHL1, 0xf, 0xc, false);
}
// the opposite direction (queue rotation towards the first lane of the pair)
__device__ __forceinline__ int dpp_next8(int src) { // every lane of the two bank groups has a source: no `old` value needed
    const int v = __builtin_amdgcn_mov_dpp(src, DPP_ROW_SHL1, 0xf, 0x3, false);
    return __builtin_amdgcn_update_dpp(v, src, DPP_ROW_SHR1, 0xf, 0xc, false);
}

// Occupancy: 168 VGPRs = three waves per SIMD by registers (round 1 needed ~200 and lost 20 % when capped; since the base ring and the
// look-ahead of round 3 the kernel fits on its own, and the transposed form is held to 168 by amdgpu_waves_per_eu at the price of
// three spilled values outside the steady loop: -2.4 %), 13 440 B of LDS = 11 waves per CU (see FP8_LDS above).
// ROLE: reads of 161 .. 320 bases are swept as TWO row blocks of 8 x 20 slots, one launch each (pl.strips == 2):
//   1 = the TOP block: rows 1 .. n - 160, right-aligned like a short read in 8 x RR slots (RR = 8, 12, 16 or 20: the host picks the
//       smallest that holds the longest read of the batch); instead of planes / h(n,m) it hands its bottom row down:
//       rowbuf[rowbuf_off + j] = {D'(n-159, j), h'(n-160, j)}, the two values a next lane would get by DPP (rebased like everything);
//   2 = the BOTTOM block: rows n - 159 .. n (no padding); its first lane takes the row above it from that buffer instead of the
//       row-0 constants: the pair's lanes load the 8 columns of the next half block with one 64-byte request and pass them to the
//       first lane through a DPP queue, one per step.  Checkpoints of all blocks are indexed by the row of the pair.
//   3 = a MIDDLE block (reads of more than 320 bases, pl.strips >= 3): 160 rows with `below` blocks under it; takes the row above
//       like the bottom block and hands its own bottom row down like the top block.
//   Blocks that hand their bottom row down stage it in LDS and store 8 columns per half block (64 bytes per pair): per-step 8-byte
//   stores / loads of the hand-over cost 30 % of the sweep when they are agent-scope (uncached) operations.
//   0 = the whole read in one block (n <= 8 * RR).
// `below` = row blocks of the pair under this one (top block: pl.strips - 1).  Blocks that hand their bottom row down keep no
// planes; only their LAST row pays the tag arithmetic, so that the handed-down {D'(r+1,j), h'(r,j)} carry their argmax tags: the
// window re-fill of the block below (fill_affine_kernel<.., WIN> with a row base) records them as the directions of its first row.
// Level L (0 = top) writes row L of the pair's row buffer ((strips - 1) rows of m + 1 entries) and reads row L - 1: the rows stay,
// the walk's re-fills of single blocks start from them.
// wblk = the wave's index among the waves of its row block (8 pairs each).  piped: the row blocks of a batch run as ONE launch
// (fp_sweep_levels_kernel), a block following the one above it through the row buffer as that one publishes its progress
// (prog_out / prog_in: the last step whose hand-over stores are out, INT_MAX at the end; rows and progress word are agent-scope
// atomics like the general path's strips, see rb_store).
template <int RR, bool XP, int ROLE, bool PK>
__device__ __forceinline__ void fp_sweep_body(int *__restrict__ lds, const int wblk, const PairPlan *__restrict__ plans, int n_pairs,
                                              const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                              const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                              const KParams &kp, int *__restrict__ hcol, int2 *__restrict__ ckpt,
                                              unsigned *__restrict__ rowi, unsigned *__restrict__ tail, int *__restrict__ err,
                                              int2 *__restrict__ rowbuf, const int below, const bool piped, const int *prog_in, int *prog_out) {
    static_assert(RR <= 2 * FP8_LW && RR > FP_PLANES, "rows per lane");
    static_assert(ROLE < 2 || RR == 2 * FP8_LW, "the bottom and middle row blocks are full: 8 x 20 slots");
    constexpr bool BOTTOM = (ROLE == 0 || ROLE == 2); // holds row n: planes, corner tags, h(n,m)
    constexpr bool HANDS = (ROLE == 1 || ROLE == 3);  // hands its bottom row down through the row buffer
    constexpr bool TAKES = (ROLE >= 2);               // takes the row above it from the row buffer
    constexpr int BOT = G8 * 2 * FP8_LW; // rows of the bottom block (the top block has the rest, in 8 x RR slots: RR as small as they fit)
    const int lane = threadIdx.x;
    const int g = lane >> 3;
    const int lp = (lane & 8) ? 15 - (lane & 15) : (lane & 7); // position of the lane inside its pair
    const int E4 = kp.e4;
    constexpr int TI = XP ? 1 : 2, TD = XP ? 2 : 1; // tags of the horizontal / vertical gap state (tie order, see above)
    constexpr bool EV = (GNX_FP_EVENTS != 0) && ROLE == 0 && !XP; // plane rows untagged + last-event tracking until the tail (see GNX_FP_EVENTS)
    if (lane < 25) lds[lane] = kp.sc4[lane] - 2 * E4;
    else if (lane < 32) lds[lane] = XP ? -E4 : -32768; // padding rows: the diagonal candidate never wins (XP: fake rows that repeat row 0)
    int *prof = GNX_FP_LDS_COMPACT ? &lds[32 + (g >> 1) * (5 * FP8_BST) + (g & 1) * (G8 * FP8_LW)] : &lds[32 + g * FP8_PST];
    const char *prof_lane = reinterpret_cast<const char *>(prof + lp * FP8_LW);

    const int pbase = wblk * 8;
    int m_max = 0, m_min = 0x7fffffff;
    for (int q = 0; q < 8; q++) {
        const int mq = (pbase + q < n_pairs) ? plans[pbase + q].m : 0;
        m_max = max(m_max, mq); m_min = min(m_min, mq);
    }
    const int p = pbase + g;
    const bool valid = p < n_pairs;
    PairPlan pl;
    if (valid) pl = plans[p]; else { pl.n = 0; pl.m = 0; pl.words = 0; pl.strips = 0; pl.trace_off = 0; pl.hcol_off = 0; pl.rowbuf_off = 0; pl.dcol_off = 0; pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; }
    const int row_base = TAKES ? pl.n - BOT * (below + 1) : 0;                      // rows of the pair above this block
    const int n_loc = ROLE == 1 ? pl.n - BOT * below : (TAKES ? BOT : pl.n);        // rows in this block
    const uint8_t *ap = a_buf + (valid ? a_start[pl.src] + row_base : 0);
    BetaSrcT<PK ? 1 : 0> bp; // (PK: beta = windows of the packed resident reference)
    bp.init(b_buf, kp, valid ? b_start[pl.src] : 0, valid ? pl.m : 0);
    const int m_eff = valid ? pl.m : 0;
    const int P = G8 * RR - n_loc; // padding slots above row 1
    const int q0 = lp * RR;        // first slot of this lane; slot q holds row q - P + 1 of the block
    int bad = 0;
    // EV: every key starts WITHOUT its tag (tg = 0) unless the very first half block is already a tail block (a wave with a window of < 7
    // columns): the event test compares values, and a tag left in the low bits would decide a tie
    const int tg = (!EV || m_min < 7) ? 1 : 0;
    int vO4, cH, cDN; // constants pinned in VGPRs (2-cycle adds, DPP `old` operands)
    // XP: the first lane's boundary is row 0 = R(j) = -e*j: h'(0,t) = R(t), D'(1,t) = R(t) + o at step t, advanced by vInc after
    // every DPP move (the values below are those of "step -1")
    asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=v"(vO4), "=v"(cH), "=v"(cDN)
                 : "s"(kp.o4), "s"(XP ? E4 + TI : kp.o4 + 2 * tg), "s"(XP ? kp.o4 + E4 + TI : 2 * kp.o4 + 2 * tg));
    const int vInc = (XP && !TAKES && lp == 0) ? -E4 : 0;           // (row 0 is above the top block only)
    const int vO4L = (XP && BOTTOM && lp == G8 - 1) ? -E4 : kp.o4;  // horizontal open of the last slot: the step of the pair's last row is free (XP)

    { // int16 profile of this lane's rows: prof[b][lp][r] = 4*(scores[alpha[row]][b] - 2e), padding -32768
        int a5[2 * FP8_LW];
#pragma unroll
        for (int r = 0; r < 2 * FP8_LW; r++) {
            int a = 5; // padding
            const int q = q0 + r;
            if (r < RR && q >= P) { a = ap[q - P]; if (a >= 5) { bad = 1; a = 4; } }
            a5[r] = a * 5;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 5; b++) {
#pragma unroll
            for (int k = 0; k < FP8_LW; k++) prof[b * FP8_BST + lp * FP8_LW + k] = (lds[a5[2 * k] + b] & 0xffff) | (lds[a5[2 * k + 1] + b] << 16);
        }
        __syncthreads();
    }
    int rt[RR], hold[RR];
#pragma unroll
    for (int r = 0; r < RR; r++) {
        const int q = q0 + r;
        // column 0: real row i: h'(i,0) = D'(i,0) = o (tag D), I'(i,1) = 2o (from D); padding: h' = I' = o; the slot above row 1 is h(0,0) = 0 (tag 3)
        // XP: padding = fake rows, h' = R(0) = 0 and an I' that never wins; the last row's I'(n,1) = h(n,0) at no cost
        hold[r] = (q >= P) ? kp.o4 + TD * tg : ((XP || q == P - 1) ? 3 * tg : kp.o4 + 2 * tg);
        rt[r] = (q >= P) ? kp.o4 + TD * tg + (r == RR - 1 ? vO4L : kp.o4) : (XP ? kp.o4 : kp.o4 + 2 * tg);
    }
    int ev[FP_PLANES]; // EV: the last step at which row n - d could have opened its horizontal gap (-1: none)
#pragma unroll
    for (int d = 0; d < FP_PLANES; d++) ev[d] = -1;
    int tstart = EV ? -1 : 0; // first step of the tagged tail
    unsigned accR[FP_PLANES] = {}; // I-planes of rows n-d = slots RR-1-d of the last lane
    unsigned tailw = 0; // argmax tags of h(n-d, m-x), d, x = 0..3, field 4x + d: lets the walk take its first diagonal steps without a window
    int diag0 = (q0 == 0) ? ((XP || P == 0) ? 3 * tg : kp.o4 + 2 * tg) : ((q0 - 1 >= P) ? kp.o4 + TD * tg : ((XP || q0 - 1 == P - 1) ? 3 * tg : kp.o4 + 2 * tg));
    if (TAKES && q0 == 0) diag0 = kp.o4 + TD; // the slot above is a row of the pair, column 0: h' = D' = o
    int dn_out = 0, h_out = 0, b_out = 0;
    int up_dn = cDN, up_h = cH;
    // beta[c] (column c, 1-based) is LOADED at the top of a half block and turned into the LDS byte offset of its profile plane just
    // before the half block's last step, where the next queue is needed: a load whose value is checked on the spot makes the wave
    // wait for the whole memory round trip (the compiler's s_waitcnt vmcnt(0) sat right behind every global_load_ubyte)
    auto base_raw = [&](int c) { return (c >= 1 && c <= m_eff) ? bp.raw(c - 1) : 0; };
    auto base_off = [&](int raw, int c) { int b = (PK && !(c >= 1 && c <= m_eff)) ? 0 : bp.value(raw, c - 1); if (b >= 5) { bad = 1; b = 4; } return b * (FP8_BST * 4); }; // (bytes: raw is 0 outside the window already)
    // The LDS offsets of the bases travel through a RING in LDS instead of a DPP queue: lane lp converts the base of column t0 + 8 + lp once
    // per half block and writes it to ring[column & 15]; at step t every lane reads ring[t + 2 - lp], the offset it needs two steps
    // later (its profile entries are fetched one step ahead of their use, see below).  One ds_read_u16 per step instead of four DPP
    // moves (queue rotation + hand-down, each twice for the mirrored pair).  16 columns are enough -- the reads of a half block reach back
    // to column t0 - 5, its write replaces columns t0 - 8 .. t0 - 1 -- and every entry is stored twice, 16 entries apart, so that the
    // eight reads of a half block are consecutive (immediate offsets, no wrap): 64 bytes per pair behind the profile (padded form:
    // the unused tail of the pair's first profile plane, 8 lanes x 10 dwords of 96).
    unsigned short *ring = reinterpret_cast<unsigned short *>(GNX_FP_LDS_COMPACT ? lds + 32 + FP8_PROF + fp8_ring_base(g) : prof + G8 * FP8_LW);
    static_assert(GNX_FP_LDS_COMPACT || FP8_BST - G8 * FP8_LW >= 16, "padded form: the ring lives in the padding of a profile plane");
    {
        const int o = base_off(base_raw(lp), lp); // columns 0 .. 7 (column 0 and everything left of it: offset 0, never used by a live cell)
        ring[lp] = (unsigned short)o; ring[lp + 16] = (unsigned short)o;
        ring[lp + 8] = 0; ring[lp + 24] = 0;      // columns -8 .. -1
    }
    int nraw = 0;

    const int level = pl.strips - 1 - below; // 0 = top block
    const int2 *rb_in = (TAKES && valid) ? rowbuf + pl.rowbuf_off + (int64_t)(level - 1) * (pl.m + 1) : nullptr; // [j] = what the block above hands down for column j
    int2 *rb_out = (HANDS && valid) ? rowbuf + pl.rowbuf_off + (int64_t)level * (pl.m + 1) : nullptr;
    // the first lane consumes one column per step from a QUEUE across the pair's lanes (like the base queue): lane lp loads column
    // t0 + lp of a half block -- one 64-byte request per pair -- and the queue moves one lane towards the first lane per step
    auto rb_at = [&](int c) { return (TAKES && valid && c >= 1 && c <= m_eff) ? rb_load(&rb_in[c], piped) : make_int2(0, 0); };
    int2 rq = make_int2(0, 0), rqn = make_int2(0, 0);
    // the last lane stages its bottom row in LDS, one column per step; after a half block lane lp stores column t0 - 7 + lp (64 bytes per pair)
    int2 *hand = reinterpret_cast<int2 *>(lds + FP8_LDS) + g * 8;
    // piped: wait until the block above has handed down the columns <= c (it stores column c at its step c + 7)
    int rb_seen = 0;
    auto wait_cols = [&](int c) {
        if (TAKES && piped && rb_seen < c + G8 - 1) {
            const long long t_begin = wall_clock64();
            while ((rb_seen = rb_progress(prog_in)) < c + G8 - 1) {
                __builtin_amdgcn_s_sleep(32);
                if (wall_clock64() - t_begin > 500000000LL) { atomicOr(err, 16); break; } // 5 s at 100 MHz
            }
        }
    };
    // The profile entries of a step are read from LDS ONE STEP AHEAD: the base a lane needs at step t + 1 is the one the previous lane
    // of its pair has at step t, so the DPP moves and the 5 ds_read_b64 for t + 1 are issued before the 19-row chain of step t and land
    // while it runs (issued at the top of their own step, every step began with an LDS round trip in s_waitcnt lgkmcnt).
    int wq[FP8_LW], pb1;
    auto fetch = [&](int pbv, int *w) {
        const int2 *pw = reinterpret_cast<const int2 *>(prof_lane + pbv);
#pragma unroll
        for (int k = 0; k < FP8_LW / 2; k++) { const int2 v = pw[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
    };
    auto ring_at = [&](int c) { return (int)ring[c & 15]; }; // (prologue only; the steps read with an immediate offset from rp)
    fetch(ring_at(0 - lp), wq); // step 0
    pb1 = ring_at(1 - lp);      // step 1
    const unsigned short *rp = ring; // ring + ((t0 + 2 - lp) & 15) of the current half block
    // u = the step's index inside its half block (compile time in the steady loop: the ring read gets an immediate offset)
    auto step = [&](const int t, auto chk, auto tgd, const bool ckflag, const int u) {
        constexpr bool CHECK = decltype(chk)::value; // false: every lane of the wave is inside its matrix (steady state)
        constexpr bool TGD = decltype(tgd)::value;   // the plane rows run the tagged arithmetic and record their I-planes (always, unless EV)
        // ckflag (wave-uniform): this half block crosses a checkpoint column
        // The first lane of a pair keeps the DPP `old` value = the row-0 boundary constant.  Passing the previous step's result
        // as `old` (its first lane already holds that constant) lets the move happen in place, without a copy of the constant.
        // (TAKES: `old` is what the block above left for this step's column of the first lane, j = t.)
        up_dn = dpp_prev8(TAKES ? rq.x : up_dn, dn_out);
        up_h = dpp_prev8(TAKES ? rq.y : up_h, h_out);
        if (TAKES) { rq.x = dpp_next8(rq.x); rq.y = dpp_next8(rq.y); }
        if (XP && !TAKES) { up_dn += vInc; up_h += vInc; }
        int wn[FP8_LW];
        fetch(pb1, wn);                    // profile entries of step t + 1
        const int pb2 = (int)rp[u];        // LDS offset of the base of step t + 2
        asm volatile("" ::: "memory"); // the reads stay HERE, ahead of the arithmetic (the scheduler would sink them next to their first use)
        const int j = t - lp;
        const int *w = wq;
        if (!CHECK || (j >= 1 && j <= m_eff)) {
            int hd = diag0, dnu = up_dn;
#pragma unroll
            for (int r = 0; r < RR; r++) {
                const int S4 = (r & 1) ? (w[r >> 1] >> 16) : (int)(short)(w[r >> 1] & 0xffff);
                if (BOTTOM && TGD && r >= RR - FP_PLANES) accR[RR - 1 - r] = alignbit2((unsigned)rt[r], accR[RR - 1 - r]);
                int hnew, dnn;
                if (BOTTOM ? (r < RR - FP_PLANES || !TGD) : r < RR - 1) { // tag bits are junk < 4 here (EV: zero); they never change the value of a max
                    const int M = hd + S4;
                    hnew = max3i(M, rt[r], dnu);
                    const int ho = hnew + vO4;
                    if (EV && BOTTOM && r >= RR - FP_PLANES) { // a plane row of the untagged part: remember the step if the gap could have been opened here
                        const bool evc = ho >= rt[r];
                        if (GNX_FP_EVBR) { if (__builtin_amdgcn_ballot_w64(evc) & 0x0180018001800180ull) { asm volatile(""); ev[RR - 1 - r] = evc ? t : ev[RR - 1 - r]; } } // (the pairs' last lanes: 7, 8, 23, 24, ...; the empty asm keeps the branch a branch: if-converted, the update costs two v_cndmask per row and step)
                        else {
                            ev[RR - 1 - r] = evc ? t : ev[RR - 1 - r];
#if GNX_FP_EVPIN
                            asm volatile("" : "+v"(ev[RR - 1 - r])); // the update stays in its step (left alone, the scheduler collects the 32 updates of a half block at its end)
#endif
                        }
                    }
                    rt[r] = max(ho, rt[r]);
                    dnn = max(ho, dnu);
                } else {
                    const int M3 = (hd | 3) + S4;
                    const int I2 = (rt[r] & ~3) | TI;
                    const int D1 = (dnu & ~3) | TD;
                    hnew = max3i(M3, I2, D1);
                    const int ho = hnew + ((XP && r == RR - 1) ? vO4L : vO4);
                    rt[r] = max(ho, I2);
                    dnn = max(ho, D1);
                }
                hd = hold[r];
                hold[r] = hnew;
                dnu = dnn;
            }
            diag0 = up_h;
            dn_out = dnu;
            h_out = hold[RR - 1];
            if (HANDS) { if (lp == G8 - 1) hand[t & 7] = make_int2(dn_out, h_out); } // hand the bottom row down (staged, see hand_down)
            if (BOTTOM && CHECK && TGD && j + 3 >= m_eff) { // the last four columns (always in a CHECK half block, and in the tagged tail)
#pragma unroll
                for (int d = 0; d < FP_PLANES; d++) tailw |= (unsigned)(hold[RR - 1 - d] & 3) << (8 * (m_eff - j) + 2 * d);
            }
            if (ckflag) { // wave-uniform: a real scalar branch (the empty asm keeps the per-lane test from being hoisted out of it)
            asm volatile("" ::: "memory");
            if ((j & (CKW - 1)) == 0 && j < m_eff && valid) { // column checkpoint {I'(i,j+1), h'(i,j)}: the keys as they are (rebased with the pair's i + j; tag bits junk)
                int2 *ck = ckpt + pl.ckpt_off + (int64_t)(j / CKW - 1) * pl.n + (q0 - P + row_base); // [r] = row q0 + r - P + 1 + row_base of the pair
                if (q0 >= P) { // every slot of this lane is a row (all lanes but the first ones of a padded block): plain stores
#pragma unroll
                    for (int r = 0; r < RR; r++) ck[r] = make_int2(rt[r], hold[r]);
                } else { // (a branch per row: only the steps at which a padded lane stands on a checkpoint column come here)
#pragma unroll
                    for (int r = 0; r < RR; r++)
                        if (q0 + r >= P) ck[r] = make_int2(rt[r], hold[r]);
                }
            }
            }
        }
#pragma unroll
        for (int k = 0; k < FP8_LW; k++) wq[k] = wn[k];
        pb1 = pb2;
    };
    // the bases of the NEXT half block (loaded at the top of this one) go into the ring at its sixth step: two steps before lane 0 reads
    // column t0 + 8, and after every read of the columns they replace
    auto ring_put = [&](int t0) {
        const int o = base_off(nraw, t0 + 8 + lp);
        const int x = ((t0 + 8) & 15) + lp;
        ring[x] = (unsigned short)o; ring[x + 16] = (unsigned short)o;
    };

    // half blocks of 8 steps t0 .. t0+7 (step t: lane lp is at column t - lp); a plane word is two half blocks
    const int Tend = ((m_max + G8 - 1) / 16 + 1) * 16;
    auto flush = [&](int t0) { // the last lane owns rows n..n-3: after the odd half block, store the plane word of steps t0-8 .. t0+7
        if (BOTTOM && (t0 & 8) && lp == G8 - 1 && valid) {
            const int w = t0 >> 4;
            if (w < pl.words) {
                const int miss = (t0 + 7) - (m_eff + G8 - 1); // steps this lane sat idle after its last column
                const int sh = (miss > 0 && miss < 16) ? 2 * miss : 0;
#pragma unroll
                for (int d = 0; d < FP_PLANES; d++) {
                    accR[d] >>= sh;
                    if (pl.n - d >= 1) rowi[pl.rowi_off + (int64_t)d * pl.words + w] = accR[d];
                }
            }
        }
    };
    auto hand_down = [&](int t0) { // after the steps t0 .. t0 + 7: the last lane was at the columns t0 - 7 .. t0
        if (HANDS) {
            __syncthreads();
            const int c = t0 - (G8 - 1) + lp;
            const int2 v = hand[lp];
            if (valid && c >= 1 && c <= m_eff) rb_store(&rb_out[c], v.x, v.y, piped);
            __syncthreads();
            if (piped && ((t0 + 8) & (GNX_FP_PUB - 1)) == 0) rb_publish(prog_out, t0 + 7, lane); // every GNX_FP_PUB steps: columns <= t0 are out
        }
    };
    auto edge_half_block = [&](int t0) { // head and tail of the sweep: some lanes are outside their matrix
        nraw = base_raw(t0 + 8 + lp); // prefetch the next half block's bases
        if (TAKES) { wait_cols(t0 + 15); rqn = rb_at(t0 + 8 + lp); }
        rp = ring + ((t0 + 2 - lp) & 15);
        const bool tail_blk = !EV || t0 + 7 > m_min; // EV: head blocks run untagged like the steady part; the tail is tagged
        if (tail_blk) {
            if (tstart < 0) tstart = t0;
#pragma unroll 1
            for (int u = 0; u < 8; u++) {
                if (u == 5) ring_put(t0);
                step(t0 + u, std::true_type{}, std::true_type{}, true, u);
            }
        } else {
#pragma unroll 1
            for (int u = 0; u < 8; u++) {
                if (u == 5) ring_put(t0);
                step(t0 + u, std::true_type{}, std::bool_constant<!EV>{}, true, u);
            }
        }
        if (TAKES) rq = rqn;
        if (tail_blk) flush(t0);
        hand_down(t0);
    };
    // Three phases, so that the steady loop is ONE loop over half blocks whose state stays in the same registers (separate
    // inner loops per half-block kind cost ~90 register copies per half block at their boundaries).
    int t0 = 0;
    if (TAKES) { wait_cols(7); rq = rb_at(lp); }
    for (; t0 < Tend && !(t0 >= 8 && t0 + 7 <= m_min); t0 += 8) edge_half_block(t0);
    for (; t0 + 7 <= m_min; t0 += 8) { // steady state: every lane of the wave is inside its matrix
        nraw = base_raw(t0 + 8 + lp);
#if GNX_FP_PRIO
        if (piped) { // issue priority by slack, as in cl_sweep_body: blocks with columns in hand (and the top block) before blocks that run right behind their producer
            const int slack = TAKES ? __builtin_amdgcn_readfirstlane(rb_seen) - t0 : (1 << 30);
            if (slack >= 48) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        }
#endif
        if (TAKES) { wait_cols(t0 + 15); rqn = rb_at(t0 + 8 + lp); }
        const bool ckflag = (t0 & (CKW - 1)) == 0; // steps t0 + lp are the lanes' checkpoint columns
        rp = ring + ((t0 + 2 - lp) & 15);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (u == 5) ring_put(t0);
            step(t0 + u, std::false_type{}, std::bool_constant<!EV>{}, ckflag, u);
        }
        if (TAKES) rq = rqn;
        if (!EV) flush(t0);
        hand_down(t0);
    }
    for (; t0 < Tend; t0 += 8) edge_half_block(t0);
    if (HANDS && piped) rb_publish(prog_out, 0x7fffffff, lane);
    if (BOTTOM && lp == G8 - 1 && valid && m_eff >= 1) {
        hcol[pl.hcol_off] = hold[RR - 1] + E4 * (pl.n + m_eff); // h(n, m)
        unsigned *tw = tail + pl.hcol_off * FP_TAILW;
        tw[0] = tailw;
#pragma unroll
        for (int d = 0; d < FP_PLANES; d++) tw[1 + d] = (unsigned)ev[d];
        tw[5] = (unsigned)tstart; tw[6] = EV ? 1u : 0u;
    }
    if (bad) atomicOr(err, 1);
}

// reads of one row block (n <= 8 * RR)
template <int RR, bool XP = false, bool PK = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void fp_sweep_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                      const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                      const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                      KParams kp, int *__restrict__ hcol, int2 *__restrict__ ckpt,
                                                      unsigned *__restrict__ rowi, unsigned *__restrict__ tail, int *__restrict__ err) {
    __shared__ int lds[FP8_LDS];
    fp_sweep_body<RR, XP, 0, PK>(lds, (int)blockIdx.x, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, hcol, ckpt, rowi, tail, err, nullptr, 0, false, nullptr, nullptr);
}

// reads of S >= 2 row blocks: the grid holds n_levels levels of W waves, block index = level-major, level level0 + blockIdx.x / W (0 = top).
//   piped = 1: ONE launch for all levels (level0 = 0, n_levels = S); wave w of a level follows wave w of the level above through
//              prog[level * W + w].  Nobody waits for a level that has not been taken: claim_items (gnx_common.hip.h).
//              (800 x 10 000, 12 000 pairs: 21.2 ms against 27.8 ms for five launches of 1 500 waves; 32 768 pairs: 51.4 / 52.6 ms.
//              Level-major order beats wave-major -- the levels of a wave group as grid neighbours, in lockstep -- 51.4 / 54.1 ms.)
//   piped = 0: one launch per level in turn (n_levels = 1): nothing to wait for (GNX_NO_PIPE, and the fallback after a timeout).
// RRTOP = slots per lane of the top block (as few as hold the longest read's rows above the full blocks).
template <int RRTOP, bool XP = false, bool PK = false>
__global__ __launch_bounds__(64) void fp_sweep_levels_kernel(const PairPlan *__restrict__ plans, int n_pairs,
                                                             const uint8_t *__restrict__ a_buf, const int64_t *__restrict__ a_start,
                                                             const uint8_t *__restrict__ b_buf, const int64_t *__restrict__ b_start,
                                                             KParams kp, int *__restrict__ hcol, int2 *__restrict__ ckpt,
                                                             unsigned *__restrict__ rowi, unsigned *__restrict__ tail, int *__restrict__ err,
                                                             int2 *__restrict__ rowbuf, int S, int W, int level0, int piped, int *__restrict__ prog) {
    __shared__ int lds[FP8_LDS + 8 * 8 * 2]; // + the staging area of the hand-over (8 pairs x 8 columns x int2)
    // piped: this workgroup sweeps level blockIdx / W of wave column blockIdx % W -- and first every level above it that nobody has
    // claimed yet (claim_items: forward progress without any assumption about dispatch order); none in the normal case.  The claim
    // words sit behind the S * W progress words.
    const int lv_own = (int)blockIdx.x / W, w = (int)blockIdx.x - lv_own * W;
    int n_stolen = 0;
    if (piped) { n_stolen = claim_items(prog + (int64_t)S * W, W, lv_own); if (n_stolen < 0) return; }
    for (int lv = lv_own - n_stolen; lv <= lv_own; lv++) {
        const int level = level0 + lv, below = S - 1 - level;
        int *po = prog + (int64_t)level * W + w;
        const int *pi = po - W;
        if (lv != lv_own - n_stolen) __syncthreads(); // the LDS profile of the level before is no longer read
        if (level == 0) fp_sweep_body<RRTOP, XP, 1, PK>(lds, w, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, hcol, ckpt, rowi, tail, err, rowbuf, below, piped != 0, nullptr, po);
        else if (below == 0) fp_sweep_body<2 * FP8_LW, XP, 2, PK>(lds, w, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, hcol, ckpt, rowi, tail, err, rowbuf, 0, piped != 0, pi, nullptr);
        else fp_sweep_body<2 * FP8_LW, XP, 3, PK>(lds, w, plans, n_pairs, a_buf, a_start, b_buf, b_start, kp, hcol, ckpt, rowi, tail, err, rowbuf, below, piped != 0, pi, po);
    }
}

} // namespace
