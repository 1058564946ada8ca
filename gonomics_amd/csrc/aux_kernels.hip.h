// aux_kernels.hip.h -- N1 score matrices, run scaling, exclusive scan of the run counts
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 4.
#pragma once
#include "gnx_common.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// N1: per-cell score matrices for the chunk / multiple-alignment variants, one thread per (chunk) cell, written
// column-major as 4*score (+ bias4, see the kernel).  A "group" is an alignment block: nseq sequences of len bases, sequence-major.
//   pairwise (AffineGapChunk):      cell = sum_k scores[a[i*c+k]][b[j*c+k]]                       (ungapped.go:7-13)
//   groups (multipleAffineGap*):    cell = sum_k scoreColumnMatch(column i*c+k, column j*c+k)     (multiAlign.go:82-110)
//     scoreColumnMatch = (sum over sequence pairs, lower case folded, gap columns skipped) / count, Go integer division
// ------------------------------------------------------------------------------------------------------
struct GroupDesc { int64_t off; int32_t nseq; int32_t len; };
struct ScorePair { int64_t a_off, b_off; int32_t a_nseq, b_nseq, a_len, b_len; int32_t nc, mc; int64_t s_off, s_pitch; int64_t pa_off, pb_off; }; // pa_off / pb_off: column profiles of the two groups (score_profiles_kernel), in entries

// S16: every 4*score fits int16 (host check: 4 * chunk * max|score| <= 32767) -- the matrix is stored as int16, which halves the
// HBM traffic of the SCORED fill (it reads one entry per cell and is bandwidth-bound with 4-byte entries)
template <bool S16>
__global__ __launch_bounds__(256) void score_matrix_kernel(const ScorePair *__restrict__ sp, const uint8_t *__restrict__ bases, KParams kp, int chunk,
                                                           int groups, int bias4, int *__restrict__ smat, int *__restrict__ err) {
    // bias4 = -2 * 4 * gapExtend * chunk when the fill runs on rebased keys (fill_affine_kernel, HFORM), else 0
    // block = 64 x 4 threads: x runs over GROUPS OF FOUR rows i (the fast index of the column-major matrix: one 8- / 16-byte store per
    // thread and column instead of four 2- / 4-byte ones), y over 4 columns j; grid.x strides over the columns, grid.y = pair.
    // No per-cell division; the 5 x 5 table sits in LDS.  (Rows nc .. of the last group of four lie inside the pitch and are never read.)
    __shared__ int sc[25];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid < 25) sc[tid] = kp.sc4[tid] / 4;
    __syncthreads();
    const ScorePair q = sp[blockIdx.y];
    for (int j = blockIdx.x * 4 + threadIdx.y; j < q.mc; j += gridDim.x * 4) {
        for (int i0 = threadIdx.x * 4; i0 < q.nc; i0 += 256) {
            int out[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = i0 + r;
                int64_t total = 0;
                for (int k = 0; i < q.nc && k < chunk; k++) {
                    const int64_t ac = (int64_t)i * chunk + k, bc = (int64_t)j * chunk + k;
                    if (!groups) {
                        const int a = bases[q.a_off + ac], b = bases[q.b_off + bc];
                        if (a >= 5 || b >= 5) { atomicOr(err, 1); continue; }
                        total += sc[a * 5 + b];
                    } else {
                        int64_t sum = 0, count = 0;
                        for (int x = 0; x < q.a_nseq; x++) {
                            int a = bases[q.a_off + (int64_t)x * q.a_len + ac];
                            if (a >= 5 && a <= 9) a -= 5;
                            for (int y = 0; y < q.b_nseq; y++) {
                                int b = bases[q.b_off + (int64_t)y * q.b_len + bc];
                                if (b >= 5 && b <= 9) b -= 5;
                                if (a != 10 && b != 10) {
                                    if (a >= 5 || b >= 5) { atomicOr(err, 1); continue; }
                                    sum += sc[a * 5 + b];
                                    count++;
                                }
                            }
                        }
                        if (count == 0) { atomicOr(err, 16); continue; } // Go: integer divide by zero
                        total += sum / count;
                    }
                }
                out[r] = (int)(4 * total) + bias4;
            }
            const int64_t at = q.s_off + (int64_t)j * q.s_pitch + i0; // multiple of 4: s_off and s_pitch are multiples of 160
            if (S16) *reinterpret_cast<uint2 *>(reinterpret_cast<short *>(smat) + at) = make_uint2((unsigned)(out[0] & 0xffff) | ((unsigned)out[1] << 16), (unsigned)(out[2] & 0xffff) | ((unsigned)out[3] << 16));
            else *reinterpret_cast<int4 *>(smat + at) = make_int4(out[0], out[1], out[2], out[3]);
        }
    }
}

// The pairwise case (AffineGapChunk: cell (i, j) = sum over the chunk's positions of scores[alpha][beta]) for chunk sizes 1 .. 4, which is
// what the callers use: a thread keeps the bases of its four rows in registers and walks the columns (the general kernel above
// re-reads them, with 64-bit index arithmetic, for every cell: 6 ms for 4096 pairs of 160 x 3000 chunks, more than the fill).
// block = 64 x 4 threads: x = groups of four rows, y = 4 columns; a block walks the columns blockIdx.x * 4 + y, + 4 * gridDim.x, ...
template <bool S16, int CH>
__global__ __launch_bounds__(256) void score_matrix_pairs_kernel(const ScorePair *__restrict__ sp, const uint8_t *__restrict__ bases, KParams kp, int bias4,
                                                                 int *__restrict__ smat, int *__restrict__ err) {
    __shared__ int sc[32];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid < 25) sc[tid] = kp.sc4[tid]; // 4 * score
    __syncthreads();
    const ScorePair q = sp[blockIdx.y];
    const uint8_t *ap = bases + q.a_off, *bq = bases + q.b_off;
    int bad = 0;
    for (int i0 = threadIdx.x * 4; i0 < q.nc; i0 += 256) {
        int a5[4][CH];
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                int a = (i0 + r < q.nc) ? ap[(i0 + r) * CH + k] : 0; // (rows nc .. of the last group of four lie inside the pitch and are never read)
                if (a >= 5) { bad = 1; a = 0; }
                a5[r][k] = a * 5;
            }
        }
        for (int j = blockIdx.x * 4 + threadIdx.y; j < q.mc; j += gridDim.x * 4) {
            int out[4] = {bias4, bias4, bias4, bias4};
#pragma unroll
            for (int k = 0; k < CH; k++) {
                int b = bq[j * CH + k];
                if (b >= 5) { bad = 1; b = 0; }
#pragma unroll
                for (int r = 0; r < 4; r++) out[r] += sc[a5[r][k] + b];
            }
            const int64_t at = q.s_off + (int64_t)j * q.s_pitch + i0;
            if (S16) *reinterpret_cast<uint2 *>(reinterpret_cast<short *>(smat) + at) = make_uint2((unsigned)(out[0] & 0xffff) | ((unsigned)out[1] << 16), (unsigned)(out[2] & 0xffff) | ((unsigned)out[3] << 16));
            else *reinterpret_cast<int4 *>(smat + at) = make_int4(out[0], out[1], out[2], out[3]);
        }
    }
    if (bad) atomicOr(err, 1);
}

// ---- groups, round 5: column profiles instead of a loop over all sequence pairs per cell ---------------------------------------------
// scoreColumnMatch(u, v) = (sum over members x of A and y of B, gaps skipped, lower case folded, of scores[a_x(u)][b_y(v)]) / count
//                        = (sum_a cntA[u][a] * wB[v][a]) / (nA(u) * nB(v)),   wB[v][a] = sum_b scores[a][b] * cntB[v][b]
// with cnt = how many members show base a in the column and n = how many show a base at all: five multiply-adds and one truncating
// division per column pair whatever the group sizes (and no division at all while both groups are single sequences: the first round of
// AllSeqAffineChunk).  The old kernel above walked every member pair of every cell with 64-bit arithmetic: 25 ms for the 28 pairs of 10 000 x
// 10 000 chunk cells of a cmd/faChunkAlign round, three times the DP itself.
struct ColProfA { short cnt[5]; short n; short bad; short pad; };          // 16 B per column of group A
struct ColProfB { int w[5]; int n; int bad; int pad; };                     // 32 B per column of group B
__global__ __launch_bounds__(256) void score_profiles_kernel(const ScorePair *__restrict__ sp, const uint8_t *__restrict__ bases, KParams kp,
                                                            ColProfA *__restrict__ pa, ColProfB *__restrict__ pb) {
    __shared__ int sc[25];
    if (threadIdx.x < 25) sc[threadIdx.x] = kp.sc4[threadIdx.x] / 4;
    __syncthreads();
    const ScorePair q = sp[blockIdx.y];
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < (int64_t)q.a_len + q.b_len; x += (int64_t)gridDim.x * blockDim.x) {
        const bool isb = x >= q.a_len;
        const int col = (int)(isb ? x - q.a_len : x), nseq = isb ? q.b_nseq : q.a_nseq, len = isb ? q.b_len : q.a_len;
        const uint8_t *src = bases + (isb ? q.b_off : q.a_off) + col;
        int cnt[5] = {0, 0, 0, 0, 0}, bad = 0;
        for (int m = 0; m < nseq; m++) {
            int a = src[(int64_t)m * len];
            if (a >= 5 && a <= 9) a -= 5; // lower case -> upper case (multiAlign.go:87-94)
            if (a == 10) continue;        // dna.Gap
            if (a > 10) { bad = 1; continue; }
#pragma unroll
            for (int z = 0; z < 5; z++) cnt[z] += (a == z);
        }
        const int n = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4];
        if (!isb) {
            ColProfA o;
#pragma unroll
            for (int z = 0; z < 5; z++) o.cnt[z] = (short)cnt[z];
            o.n = (short)n; o.bad = (short)bad; o.pad = 0;
            pa[q.pa_off + col] = o;
        } else {
            ColProfB o;
#pragma unroll
            for (int a = 0; a < 5; a++) { int w = 0; for (int b = 0; b < 5; b++) w += sc[a * 5 + b] * cnt[b]; o.w[a] = w; }
            o.n = n; o.bad = bad; o.pad = 0;
            pb[q.pb_off + col] = o;
        }
    }
}
// block = 64 x 4 threads: x = groups of four chunk rows (their column profiles stay in registers), y = 4 chunk columns; see score_matrix_pairs_kernel
template <bool S16, int CH>
__global__ __launch_bounds__(256) void score_matrix_groups_kernel(const ScorePair *__restrict__ sp, KParams kp, int bias4, const ColProfA *__restrict__ pa,
                                                                  const ColProfB *__restrict__ pb, int *__restrict__ smat, int *__restrict__ err) {
    const ScorePair q = sp[blockIdx.y];
    const bool single = q.a_nseq == 1 && q.b_nseq == 1; // every count is 0 or 1: no division
    int flags = 0;
    for (int i0 = threadIdx.x * 4; i0 < q.nc; i0 += 256) {
        short ca[4][CH][5];
        int na[4][CH];
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int k = 0; k < CH; k++) {
                ColProfA o;
                if (i0 + r < q.nc) o = pa[q.pa_off + (int64_t)(i0 + r) * CH + k];
                else { for (int z = 0; z < 5; z++) o.cnt[z] = 0; o.n = 1; o.bad = 0; } // (rows nc .. of the last group of four lie inside the pitch and are never read)
#pragma unroll
                for (int z = 0; z < 5; z++) ca[r][k][z] = o.cnt[z];
                na[r][k] = o.n;
                flags |= o.bad ? 1 : 0;
            }
        }
        for (int j = blockIdx.x * 4 + threadIdx.y; j < q.mc; j += gridDim.x * 4) {
            int out[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const ColProfB o = pb[q.pb_off + (int64_t)j * CH + k];
                flags |= o.bad ? 1 : 0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int sum = 0;
#pragma unroll
                    for (int z = 0; z < 5; z++) sum += (int)ca[r][k][z] * o.w[z];
                    const int count = na[r][k] * o.n;
                    if (count == 0) { if (i0 + r < q.nc) flags |= 16; continue; } // Go: integer divide by zero (multiAlign.go:101)
                    out[r] += single ? sum : sum / count;                     // Go's truncating division
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) out[r] = 4 * out[r] + bias4;
            const int64_t at = q.s_off + (int64_t)j * q.s_pitch + i0;
            if (S16) *reinterpret_cast<uint2 *>(reinterpret_cast<short *>(smat) + at) = make_uint2((unsigned)(out[0] & 0xffff) | ((unsigned)out[1] << 16), (unsigned)(out[2] & 0xffff) | ((unsigned)out[3] << 16));
            else *reinterpret_cast<int4 *>(smat + at) = make_int4(out[0], out[1], out[2], out[3]);
        }
    }
    if (flags) atomicOr(err, flags);
}

__global__ __launch_bounds__(256) void scale_runs_kernel(gnx_cigar *__restrict__ ops, int64_t total, int64_t factor) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < total) ops[x].run_length *= factor; // expandCigarRunLength, affineGap_highMem.go:91-95
}

// exclusive scan of nops[0..n) + carry[0] -> off[0..n], off[n]; carry[0] = off[n] afterwards.  One block.
__global__ __launch_bounds__(1024) void scan_kernel(const int64_t *__restrict__ nops, int n, int64_t *__restrict__ off, int64_t *__restrict__ carry) {
    // exclusive scan of the run counts, 1024 elements per round: wave-level scans by __shfl_up, then the 16 wave totals
    // (a one-pass version with a contiguous slice per thread measured slower: 227 us vs 139 us per 100 k elements)
    __shared__ int64_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t base = carry[0];
    for (int start = 0; start < n; start += 1024) {
        const int idx = start + threadIdx.x;
        const int64_t v = idx < n ? nops[idx] : 0;
        int64_t sum = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int64_t t = __shfl_up(sum, d, 64); if (lane >= d) sum += t; }
        if (lane == 63) wsum[wave] = sum;
        __syncthreads();
        if (wave == 0) {
            int64_t t = lane < 16 ? wsum[lane] : 0;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) { const int64_t u = __shfl_up(t, d, 64); if (lane >= d) t += u; }
            if (lane < 16) wsum[lane] = t;
        }
        __syncthreads();
        if (idx < n) off[idx] = base + (wave > 0 ? wsum[wave - 1] : 0) + sum - v;
        base += wsum[15];
        __syncthreads();
    }
    if (threadIdx.x == 0) { off[n] = base; carry[0] = base; }
}

// Large inputs: three launches instead of one 1024-thread block walking the whole array (139 us per 100 k elements):
// per-block sums -> scan_kernel over the sums (it adds and updates the carry) -> per-block scans on top of their offsets.
__global__ __launch_bounds__(1024) void scan_sums_kernel(const int64_t *__restrict__ nops, int n, int64_t *__restrict__ sums) {
    __shared__ int64_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 1024 + threadIdx.x;
    int64_t v = idx < n ? nops[idx] : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) { int64_t t = 0; for (int w = 0; w < 16; w++) t += wsum[w]; sums[blockIdx.x] = t; }
}
// block_off[b] = exclusive offset of block b (carry included), from scan_kernel over the sums
__global__ __launch_bounds__(1024) void scan_apply_kernel(const int64_t *__restrict__ nops, int n, const int64_t *__restrict__ block_off, int64_t *__restrict__ off) {
    __shared__ int64_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 1024 + threadIdx.x;
    const int64_t v = idx < n ? nops[idx] : 0;
    int64_t sum = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int64_t t = __shfl_up(sum, d, 64); if (lane >= d) sum += t; }
    if (lane == 63) wsum[wave] = sum;
    __syncthreads();
    if (wave == 0) {
        int64_t t = lane < 16 ? wsum[lane] : 0;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { const int64_t u = __shfl_up(t, d, 64); if (lane >= d) t += u; }
        if (lane < 16) wsum[lane] = t;
    }
    __syncthreads();
    if (idx < n) off[idx] = block_off[blockIdx.x] + (wave > 0 ? wsum[wave - 1] : 0) + sum - v;
    if (idx == n - 1) off[n] = block_off[blockIdx.x] + (wave > 0 ? wsum[wave - 1] : 0) + sum;
}

// single-pass cooperative traceback (traceback_kernel<.., SCR>): runs staged in traceback order -> dense output, alignment order
__global__ __launch_bounds__(256) void reverse_runs_kernel(const PairPlan *__restrict__ plans, int n_pairs, const gnx_cigar *__restrict__ scr,
                                                            const int64_t *__restrict__ scr_off, const int64_t *__restrict__ nops,
                                                            const int64_t *__restrict__ ops_off, gnx_cigar *__restrict__ ops, int64_t ops_capacity,
                                                            int *__restrict__ err) {
    const int p = blockIdx.x;
    if (p >= n_pairs) return;
    const int po = plans[p].src;
    const int64_t cnt = nops[po], base = ops_off[po], sb = scr_off[p];
    if (base + cnt > ops_capacity) { if (threadIdx.x == 0) atomicOr(err, 4); return; }
    for (int64_t x = threadIdx.x; x < cnt; x += blockDim.x) ops[base + (cnt - 1 - x)] = scr[sb + x];
}

// windows of the packed reference back to bytes: for the kernels of the general and the snapshot paths (run_device), for
// AffineGapLocal on the resident reference (its fast path reads the long sequence as the kernels' alpha) and, with GNX_REF_UNPACK=1,
// for every call (the A/B of the packed reads)
__global__ __launch_bounds__(256) void unpack_windows_kernel(KParams kp, const int64_t *__restrict__ start, const int64_t *__restrict__ out_off, int n_pairs,
                                                             uint8_t *__restrict__ out) {
    const int p = blockIdx.x; // one workgroup per window
    if (p >= n_pairs) return;
    const int64_t len = out_off[p + 1] - out_off[p];
    BetaSrc b;
    b.init(nullptr, kp, start[p], len);
    for (int64_t k = threadIdx.x; k < len; k += blockDim.x) out[out_off[p] + k] = (uint8_t)b.at(k);
}

// CIGAR offsets of a sub-batch (they start at 0) moved behind the runs of the sub-batches before it
__global__ __launch_bounds__(256) void add_offset_kernel(int64_t *__restrict__ off, int64_t n, int64_t base) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) off[x] += base;
}

} // namespace
