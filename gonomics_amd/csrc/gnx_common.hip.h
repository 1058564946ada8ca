// gnx_common.hip.h -- geometry constants, PairPlan / KParams, DPP and max3 helpers
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "gnx_align.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int G = 16;            // lanes per pair (one DPP row)
constexpr int R = 10;            // DP rows per lane
constexpr int H = G * R;         // rows per strip
constexpr int NEG4 = -(1 << 30); // scaled "veryNegNum" (align/align.go:8); finite keys stay above -(1<<29)
constexpr int QA = 8;            // uint4 stores per lane per flush, affine (3*R=30 dwords -> 32)
constexpr int QC = 3;            // const gap (R=10 dwords -> 12)
#ifndef GNX_RB_PUB
#define GNX_RB_PUB 64
#endif
constexpr int RB_PUB = GNX_RB_PUB;  // pipelined strips publish their bottom row every RB_PUB steps (power of two, multiple of 16)

#define DPP_ROW_SHR1 0x111
#define DPP_ROW_SHL1 0x101

struct PairPlan {
    int32_t n, m;
    int32_t words;      // 16-column direction words per strip
    int32_t strips;     // ceil(n / H)
    int64_t trace_off;  // in uint4 units, relative to the chunk's trace buffer
    int64_t hcol_off;   // ints
    int64_t rowbuf_off; // int2
    int64_t dcol_off;   // dwords: per strip and lane one word with the last-column direction fields of the lane's R rows
    // fast path (short alpha) / window re-fill:
    int32_t src;        // index of the pair in the a_start / b_start tables (== own index except for window plans)
    int32_t col_off;    // first column of a window re-fill minus one (0 = whole matrix); multiple of CKW
    int64_t ckpt_off;   // int2: column checkpoints of the pair, [c-1][row] for column c*CKW
    int64_t rowi_off;   // dwords: I-plane of row n (one word per 16 steps of the owner lane)
    int64_t s_off;      // SCORED kernels: explicit 4*score matrix of the pair, column-major: S[s_off + (j-1)*s_pitch + (i-1)]
    int64_t s_pitch;
};

constexpr int CKW = 128;     // column checkpoint spacing of the fast path
constexpr int FP_SPAN = 192; // a re-fill window (one row block of <= 160 rows) is at least this wide: the rows + typical indels
#ifndef GNX_FP_PLANES
#define GNX_FP_PLANES 4
#endif
constexpr int FP_PLANES = GNX_FP_PLANES; // rows n .. n-3 keep their I-plane: a trailing gap sits on row n-d when the last d bases match the chunk end
constexpr int FP_TAILW = 8; // dwords per pair of the sweep's small outputs: [0] corner tags, [1 + d] last event step of row n - d, [5] first tagged step, [6] events in use (fp_sweep.hip.h: GNX_FP_EVENTS)
constexpr int FP_CAP = 64;   // CIGAR runs staged per pair and row block on the fast path (more -> general path)
constexpr int FP_MAXS = 128; // row blocks of 160 rows the fast path sweeps (reads up to 20 480 bases; longer ones: snapshot path)
constexpr int FP_TILE = 1024;                                  // straggler tiles: columns (c*FP_TILE, (c+1)*FP_TILE]
constexpr int FP_TWORDS = (FP_TILE + CKW + 15 + 15) / 16;      // direction words of a tile (plus the checkpoint interval it starts early)
constexpr int FP_WWORDS = (FP_SPAN + CKW + 15 + 15) / 16 + 1;                                         // direction words of the widest window
constexpr int FP_SPEC = 4;                                                                             // windows per request of a read of several row blocks (1 asked for + speculative ones)
__host__ __device__ constexpr int fp_spec_margin(int k) { return k == 0 ? 0 : 16 + 8 * k; }            // how far a path may drift from the diagonal over k row blocks and still find its window
__host__ __device__ constexpr int fp_spec_wwords(int K) { return K <= 1 ? FP_WWORDS : (FP_SPAN + 2 * fp_spec_margin(K - 1) + CKW + 15 + 15) / 16 + 1; }
__host__ __device__ constexpr int fp_cap(int S) { return FP_CAP * (S < 1 ? 1 : S); }                   // staged CIGAR runs per pair of S row blocks

struct KParams {
    int sc4[25]; // 4*scores
    int oe4, e4, o4;
    int d00_4;   // 4*D(0,0): gapOpen, or 0 with free end gaps
    int ecol4;   // 4*(column-0 extension): gapExtend, or 0 with free end gaps
    int g4;      // const gap: 4*gapPen
    int ckc;     // cl_sweep_kernel: snapshot spacing in steps (CKC or CKC_SMALL, set per call)
    int rb_pub;  // cl_sweep_kernel: piped strips publish their bottom row every rb_pub steps (power of two, multiple of 16; set per launch)
    // beta = windows of the PACKED resident reference (gnx_set_reference; all null: plain dna.Base bytes in b_buf):
    const unsigned *b2;              // 2 bits per base, 16 bases per dword (base k: bits 2*(k & 15) of word k >> 4); A C G T = 0 1 2 3
    const unsigned long long *bflag; // one bit per 64-base block: the block holds a base that is not A C G T (N, or a byte >= 5)
    const unsigned *brank;           // per flag word (4096 bases): number of such blocks before it
    const unsigned long long *bexc;  // per such block, in order: {mask of N, mask of bytes >= 5}
};

// ---- beta as the kernels read it -------------------------------------------------------------------------------------------------
// Plain: dna.Base bytes.  Packed (the resident reference, SURVEY 7 step 4 / north_star "packed dna.Base sequences"): 2 bits per base
// plus a sparse exception list -- 4.4e9 bases in 1.1 GB instead of 4.4 GB, on the device and on the RCCL broadcast.  raw(k) is the
// LOAD (the byte, or the dword holding base k), value(raw, k) the arithmetic that turns it into 0 .. 4 (5: a byte >= 5, which the
// kernels flag like before): the kernels issue the load a block ahead and take the value where they need it.  A window without
// exceptions (`dirty` false: all but ~0.02 % of the windows of a genome with its N runs) never touches the exception structures.
// MODE 0 / 1: bytes / packed known at compile time (the headline sweep has an instantiation of each: the run-time form costs it the
// 1.3 % the look-ahead gained, 29.3 -> 29.7 ms per 100 000 pairs); 2: decided per launch (every other kernel).
template <int MODE>
struct BetaSrcT {
    const uint8_t *bytes;
    const unsigned *w2;
    const KParams *kp;
    int64_t off; // absolute index of column 1 of this pair's window
    bool dirty;
    __device__ __forceinline__ bool packed() const { return MODE == 1 || (MODE == 2 && w2 != nullptr); }
    __device__ __forceinline__ void init(const uint8_t *b_buf, const KParams &k, int64_t start, int64_t len) {
        bytes = b_buf; w2 = k.b2; kp = &k; off = start; dirty = false;
        if (packed() && len > 0) { // exception blocks before block x: rank of the flag word + bits below
            auto rank = [&](int64_t x) { const unsigned long long f = k.bflag[x >> 6]; return (int64_t)k.brank[x >> 6] + __popcll(f & ((1ull << (x & 63)) - 1ull)); };
            dirty = rank(((start + len - 1) >> 6) + 1) > rank(start >> 6);
        }
    }
    __device__ __forceinline__ int raw(int64_t k) const { return packed() ? (int)w2[(off + k) >> 4] : (int)bytes[off + k]; }
    __device__ __forceinline__ int value(int r, int64_t k) const {
        if (!packed()) return r;
        const int64_t x = off + k;
        int b = (int)(((unsigned)r >> (2 * ((int)x & 15))) & 3u);
        if (dirty) {
            const unsigned long long f = kp->bflag[x >> 12];
            const int bit = (int)(x >> 6) & 63;
            if ((f >> bit) & 1ull) {
                const int64_t e = (int64_t)kp->brank[x >> 12] + __popcll(f & ((1ull << bit) - 1ull));
                if ((kp->bexc[2 * e] >> (x & 63)) & 1ull) b = 4;
                if ((kp->bexc[2 * e + 1] >> (x & 63)) & 1ull) b = 5;
            }
        }
        return b;
    }
    __device__ __forceinline__ int at(int64_t k) const { return value(raw(k), k); } // load and use on the spot
};
using BetaSrc = BetaSrcT<2>;
// The kernels of the general and the snapshot paths read beta as bytes only: with both forms compiled into their unrolled steps the
// register allocator gives up (fill_affine_kernel 152 -> 512 registers + scratch, one wave per SIMD; fill_const_kernel 113 -> 256);
// run_device unpacks the windows of a packed reference before it launches them.
using BetaBytes = BetaSrcT<0>;

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int dpp_shr1(int oldv, int src) { return __builtin_amdgcn_update_dpp(oldv, src, DPP_ROW_SHR1, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_shl1(int oldv, int src) { return __builtin_amdgcn_update_dpp(oldv, src, DPP_ROW_SHL1, 0xf, 0xf, false); }
// direction-matrix store.  Pipelined strips (MULTI kernels, see fill_affine_kernel) publish their bottom row with an agent-scope
// release, which writes the XCD's dirty L2 lines back: their direction words are stored non-temporally so that they do not pile up
// as dirty lines (they are full lines that the fill never reads back).
__device__ __forceinline__ void trace_store(uint4 *dst, unsigned a, unsigned b, unsigned c, unsigned d, bool streaming) {
    if (streaming) __builtin_nontemporal_store((u32x4){a, b, c, d}, reinterpret_cast<u32x4 *>(dst));
    else *dst = make_uint4(a, b, c, d);
}
// Row-buffer hand-over between the pipelined strips of a pair (fill_affine_kernel / fill_const_kernel, MULTI with a strip map).
// The XCDs' L2 caches are not coherent with each other, so the producer's rows must reach memory and the consumer must not read a
// stale line.  GNX_RB_FENCE: agent-scope release / acquire fences (L2 write-back + invalidate, ~780 of each per strip of a 100 kb
// beta -- they serialize per XCD and cap the fill at ~1.1e12 cells/s whatever the number of strips in flight).  Default: the rows
// themselves are agent-scope relaxed atomics (sc1: write-through stores, loads that bypass the non-coherent lines); the producer
// waits for its stores to be acknowledged (s_waitcnt vmcnt(0)) before it advances the progress word, the consumer reads rows only
// after it has seen the progress word.  No cache-wide operation on either side.
#ifndef GNX_RB_FENCE
#define GNX_RB_FENCE 0
#endif
__device__ __forceinline__ void rb_store(int2 *p, int a, int b, bool piped) {
    if (piped && !GNX_RB_FENCE)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), ((unsigned long long)(unsigned)b << 32) | (unsigned)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = make_int2(a, b);
}
__device__ __forceinline__ int2 rb_load(const int2 *p, bool piped) {
    if (piped && !GNX_RB_FENCE) {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_int2((int)(unsigned)v, (int)(unsigned)(v >> 32));
    }
    return *p;
}
// producer: everything stored so far is out; then the progress word
__device__ __forceinline__ void rb_publish(int *prog, int value, int lane) {
    if (GNX_RB_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(prog, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumer: progress of the producer (columns published so far)
__device__ __forceinline__ int rb_progress(const int *prog) {
    if (GNX_RB_FENCE) return __hip_atomic_load(prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    const int v = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory"); // row loads stay behind the progress load
    return v;
}
// ---- forward progress of piped launches without any assumption about dispatch order: CLAIMS ----------------------------------------
// A workgroup of a piped launch spins on the progress of the workgroup that runs the item before its own (the strip above, the
// level above), and HIP promises nothing about the order in which workgroups are dispatched (MI355X_MICROARCH "Workgroup dispatch":
// placement-independent protocols only).  So nobody waits for work that has not been taken: every item has a claim word (zeroed by
// the host).  A workgroup claims its own item -- the one of its block index, so the placement of the normal case is exactly that of
// a launch that trusted the dispatch order -- and then, going up its chain, every predecessor item that is still unclaimed: those it
// runs itself, first, one after the other (the kernels loop over strips / levels anyway) -- after a short grace period, see below.  A workgroup whose own item was taken by
// a successor exits.  The predecessor of the first item a workgroup runs is therefore always claimed by a workgroup that is running
// -- one that runs its items in chain order and, by induction, never waits for unclaimed work: the lowest unfinished item of a chain
// is always being executed.  In the normal case (dispatch in index order) every claim of a predecessor fails and nothing changes.
// (A first version handed out TICKETS, atomicAdd on a launch counter: as safe, but the item a workgroup got then depended on
// the race to the counter, and the pipelined constant-gap sweep of config C5 lost 5 % -- 234 -> 247 ms for 1024 pairs, same box,
// profiles/r3_experiments.md.)
// cw[0 .. n_items) claim words, cw[n_items] = test switch GNX_TICKET_DELAY: the workgroups of the lower half of the grid sleep before
// they claim, so the upper half finds its predecessors unclaimed and runs them (the suite proves results do not depend on who runs what).
// `stride` = distance of an item from its predecessor in block indices (1: strips of a group; W: levels of a wave column), `depth` =
// number of predecessors of this workgroup's own item.  Returns how many of them it has to run itself (0 in the normal case), -1 if
// its own item is already taken.  One wave per workgroup (all piped kernels).
#ifndef GNX_CLAIM_MODE
#define GNX_CLAIM_MODE 2
#endif
#ifndef GNX_CLAIM_GRACE_US
#define GNX_CLAIM_GRACE_US 20000
#endif
constexpr long long CLAIM_GRACE_TICKS = 100LL * GNX_CLAIM_GRACE_US; // ticks of the 100 MHz wall clock
// quirk Q1 (align/affineGap.go:305) as the walks of the snapshot path met it: [0] checkerboard edges crossed upwards with the restart rule applied, [1] those where the
// argmax state of the entry cell differed from the traced state -- the crossings that change the CIGAR (gnx_debug_counter(3 / 4); tests/test_long_range.py bounds the
// re-score deficit of a megabase CIGAR with it)
__device__ unsigned long long g_dev_q1[2];
__device__ __forceinline__ void q1_report(int n, int changed) {
    if (n > 0) atomicAdd(&g_dev_q1[0], (unsigned long long)n);
    if (changed > 0) atomicAdd(&g_dev_q1[1], (unsigned long long)changed);
}
__device__ unsigned long long g_dev_claims_stolen; // items run by a workgroup other than their own (gnx_debug_counter(0): the tests prove the abnormal paths ran)
// The test switch word cw[n_items]: bits 0-15 = GNX_TICKET_DELAY (sleep units of the lower half of the grid), bits 16-31 =
// GNX_CLAIM_GRACE_US (a grace period in microseconds for this launch instead of the compiled-in 20 ms: a delay longer than the grace
// period makes the upper half really TAKE its predecessors' items; with the default grace the delayed workgroups still arrive in time).
__device__ __forceinline__ int claim_items(int *cw, int stride, int depth) {
    int n = 0;
#if GNX_CLAIM_MODE == 0
    (void)cw; (void)stride; (void)depth; return 0; // experiment: no claims (the round-2 protocol: trust the dispatch order)
#endif
    if (threadIdx.x == 0) {
        const int b = (int)blockIdx.x;
#if GNX_CLAIM_MODE == 1
        (void)stride; (void)depth; if (atomicCAS(&cw[b], 0, 1) != 0) n = -1; // experiment: own item only
        return __builtin_amdgcn_readfirstlane(n);
#endif
        const int sw = cw[gridDim.x];
        const int delay = sw & 0xffff;
        const long long grace = (sw >> 16) ? 100LL * ((sw >> 16) & 0xffff) : CLAIM_GRACE_TICKS;
        // who sleeps: the lower half of the grid; with a grace override (the steal leg) every other item of a chain instead, so that
        // every awake workgroup has a sleeping direct predecessor whatever the grid looks like
        const bool sleeper = (sw >> 16) ? (((b / stride) & 1) == 0) : (b * 2 < (int)gridDim.x);
        if (delay && sleeper) for (int i = 0; i < delay; i++) __builtin_amdgcn_s_sleep(127);
        if (atomicCAS(&cw[b], 0, 1) != 0) n = -1;
        else while (n < depth) {
            // The predecessor's own workgroup gets a GRACE period to arrive (workgroups of one launch start within microseconds of
            // each other, but an XCD hands out its share of the grid as ITS slots free up, so a workgroup can be resident long before its
            // predecessor on the neighbouring XCD): taking an unclaimed item at once, or after 50 us, made whoever won the race run whole
            // chains alone -- config C5 7 x slower.  After 20 ms the item is taken: the wait is bounded, no progress depends on dispatch.
            int *pw = &cw[b - (n + 1) * stride];
            const long long t_begin = wall_clock64();
            int seen;
            while ((seen = __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && wall_clock64() - t_begin < grace) __builtin_amdgcn_s_sleep(8);
            if (seen != 0 || atomicCAS(pw, 0, 1) != 0) break;
            n++;
        }
        if (n > 0) atomicAdd(&g_dev_claims_stolen, (unsigned long long)n);
    }
    return __builtin_amdgcn_readfirstlane(n);
}
__device__ __forceinline__ unsigned alignbit2(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 2); }

} // namespace
