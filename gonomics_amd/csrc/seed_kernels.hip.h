// seed_kernels.hip.h -- "next" row N4: the k-mer index of the graph aligner and its seed search (hash lookup + exact-match extension)
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md section 5.3.
#pragma once
#include <rocprim/rocprim.hpp>
#include "gnx_common.hip.h"

namespace {
// ------------------------------------------------------------------------------------------------------
// genomeGraph.IndexGenomeIntoMap (/root/reference/genomeGraph/index.go:21-43) for the k-mers that lie inside one node: a Go map from
// the 2-bit code of `seedLen` bases to the list of (node << 32 | pos) it was seen at, positions 0, seedStep, 2 seedStep, ...; k-mers
// with an N are skipped.  Here: one thread per position emits (key, location) or nothing, an exclusive scan compacts, a STABLE
// radix sort by key (rocprim::radix_sort_pairs) makes the map: equal keys keep the insertion order of the reference (by node, by position).  The
// k-mers that run across node borders (index.go:34-38, a recursion over Next edges) are few and stay on the host.
//
// seedMapMemPool (search.go:549-590) + dnaTwoBit.CountLeftMatches / CountRightMatches (dna/dnaTwoBit/perfectAlign.go): for every
// read position and strand the k-mer is looked up (binary search in the sorted keys), and every hit is extended to the left and to
// the right while the bases match.  The reference compares 64-bit words of 32 two-bit bases -- built with `answer<<2 | base`, so
// that an N (4) also sets the low bit of the base before it -- and so do the kernels: nodes and the 32 shifted copies ("rainbow")
// of each read strand are packed the same way first, which makes hits and match lengths those of the reference for any input.
// One thread per (read, position, strand); hits come out in the reference's order of discovery (position, strand, map order).
// ------------------------------------------------------------------------------------------------------
struct SeedHit { int32_t read_start, strand, node, node_start, q_start, right; };

__global__ __launch_bounds__(256) void seed_index_count_kernel(const uint8_t *__restrict__ cat, const int64_t *__restrict__ node_off, int n_nodes,
                                                               const int64_t *__restrict__ slot_off, int seed_len, int seed_step,
                                                               uint64_t *__restrict__ keys, uint64_t *__restrict__ locs, int *__restrict__ flag, int64_t n_slots) {
    // slot = (node, k): position k * seed_step of the node; flag[slot] = 1 if it has a k-mer
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_slots) return;
    int lo = 0, hi = n_nodes - 1; // node of slot x: last node with slot_off <= x
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (slot_off[mid] <= x) lo = mid; else hi = mid - 1; }
    const int node = lo;
    const int64_t pos = (x - slot_off[node]) * seed_step;
    const uint8_t *s = cat + node_off[node] + pos;
    uint64_t key = 0;
    bool ok = true;
    for (int k = 0; k < seed_len; k++) { const uint8_t b = s[k]; if (b >= 4) ok = false; key = (key << 2) | (uint64_t)(b & 3); }
    flag[x] = ok ? 1 : 0;
    keys[x] = key;
    locs[x] = ((uint64_t)node << 32) | (uint64_t)pos;
}
__global__ __launch_bounds__(256) void seed_index_compact_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ locs, const int *__restrict__ flag,
                                                                 const int64_t *__restrict__ off, int64_t n_slots, uint64_t *__restrict__ okeys, uint64_t *__restrict__ olocs) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n_slots && flag[x]) { okeys[off[x]] = keys[x]; olocs[off[x]] = locs[x]; }
}
__global__ __launch_bounds__(256) void flag_to_i64_kernel(const int *__restrict__ flag, int64_t n, int64_t *__restrict__ out) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) out[x] = flag[x];
}

// 32 bases -> one left-aligned word, exactly like dnaTwoBit.BasesToUint64LeftAln (an N spills into the base before it)
__device__ __forceinline__ uint64_t pack_word(const uint8_t *s, int64_t len, int64_t start, int lead_a) {
    // word of the sequence "lead_a x 'A' + s": clone positions [start, start + 32)
    uint64_t w = 0;
    int cnt = 0;
    for (int k = 0; k < 32; k++) {
        const int64_t p = start + k - lead_a;
        if (start + k >= len + lead_a) break;
        const uint64_t b = (p < 0) ? 0 : (uint64_t)s[p];
        w = (w << 2) | b;
        cnt++;
    }
    return cnt ? (w << (2 * (32 - cnt))) : 0;
}
__global__ __launch_bounds__(256) void pack_nodes_kernel(const uint8_t *__restrict__ cat, const int64_t *__restrict__ node_off, const int64_t *__restrict__ word_off,
                                                         int n_nodes, int64_t n_words, uint64_t *__restrict__ words) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_words) return;
    int lo = 0, hi = n_nodes - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (word_off[mid] <= x) lo = mid; else hi = mid - 1; }
    const int64_t len = node_off[lo + 1] - node_off[lo];
    words[x] = pack_word(cat + node_off[lo], len, (x - word_off[lo]) * 32, 0);
}
// rainbow of read r, strand s: 32 shifted copies of RW words each (RW = words of the longest copy), at ((r*2 + s)*32 + o) * RW
__global__ __launch_bounds__(256) void pack_reads_kernel(const uint8_t *__restrict__ cat, const int64_t *__restrict__ read_off, int n_reads, int RW,
                                                         uint8_t *__restrict__ rc, uint64_t *__restrict__ words) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // (read, strand, offset, word)
    const int64_t total = (int64_t)n_reads * 2 * 32 * RW;
    if (x >= total) return;
    const int w = (int)(x % RW), o = (int)((x / RW) % 32), st = (int)((x / ((int64_t)RW * 32)) % 2);
    const int64_t r = x / ((int64_t)RW * 64);
    const int64_t len = read_off[r + 1] - read_off[r];
    const uint8_t *s = (st ? rc : cat) + read_off[r];
    words[x] = pack_word(s, len, (int64_t)w * 32, o);
}
// out[r] = slot_offsets[first_slot[r]] for r in [0, n): where the hits of read r start (gnx_seed_find_batch)
__global__ __launch_bounds__(256) void seed_read_off_kernel(const int64_t *__restrict__ slot_offsets, const int64_t *__restrict__ first_slot, int64_t n, int64_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = slot_offsets[first_slot[r]];
}
__global__ __launch_bounds__(256) void revcomp_kernel(const uint8_t *__restrict__ cat, const int64_t *__restrict__ read_off, int n_reads, int64_t total, uint8_t *__restrict__ rc) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= total) return;
    int lo = 0, hi = n_reads - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (read_off[mid] <= x) lo = mid; else hi = mid - 1; }
    const int64_t b = read_off[lo], e = read_off[lo + 1];
    const uint8_t v = cat[e - 1 - (x - b)];
    rc[x] = v < 4 ? (uint8_t)(3 - v) : v; // dna.ReverseComplement: N stays N
}

struct SeedCtx {
    const uint64_t *keys, *locs; int64_t n_index;
    const uint64_t *node_words; const int64_t *node_off, *word_off;
    const uint64_t *read_words; const int64_t *read_off; int RW;
    int seed_len;
};
__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t *a, int64_t n, uint64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
// slot = (read, position, strand); FILL = false: counts[slot] = number of hits; FILL = true: hits written at off[slot] ..
template <bool FILL>
__global__ __launch_bounds__(128) void seed_find_kernel(SeedCtx c, const int64_t *__restrict__ slot_off, int n_reads, int64_t n_slots,
                                                        int64_t *__restrict__ counts, const int64_t *__restrict__ off, SeedHit *__restrict__ hits) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_slots) return;
    int lo = 0, hi = n_reads - 1; // read of slot x (slot_off[r] = 2 * positions of the reads before r)
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (slot_off[mid] <= x) lo = mid; else hi = mid - 1; }
    const int r = lo;
    const int64_t rel = x - slot_off[r];
    const int readStart = (int)(rel >> 1), st = (int)(rel & 1);
    const int read_len = (int)(c.read_off[r + 1] - c.read_off[r]);
    const uint64_t *rain = c.read_words + ((int64_t)r * 2 + st) * 32 * c.RW; // [offset][word]
    // key = Rainbow[keyOffset].Seq[keyIdx] >> keyShift (search.go:555-559)
    const int keyIdx = (readStart + 31) / 32, keyOffset = 31 - ((readStart + 31) % 32);
    const uint64_t key = rain[(int64_t)keyOffset * c.RW + keyIdx] >> (64 - 2 * c.seed_len);
    const int64_t first = lower_bound_u64(c.keys, c.n_index, key);
    int64_t n = 0;
    while (first + n < c.n_index && c.keys[first + n] == key) n++;
    if (!FILL) { counts[x] = n; return; }
    for (int64_t h = 0; h < n; h++) {
        const uint64_t code = c.locs[first + h];
        const int node = (int)(code >> 32), nodePos = (int)(code & 0xffffffffu);
        const uint64_t *nw = c.node_words + c.word_off[node];
        const int node_len = (int)(c.node_off[node + 1] - c.node_off[node]);
        // leftMatches = min(readStart + 1, CountLeftMatches(node, nodePos, Rainbow[readOffset], readStart + readOffset))
        int readOffset = 31 - ((readStart - (nodePos % 32) + 31) % 32);
        const uint64_t *rw = rain + (int64_t)readOffset * c.RW;
        int left;
        {
            const int startTwo = readStart + readOffset;
            const int offBits = (nodePos % 32) * 2, noLook = 64 - offBits - 2;
            int i = nodePos / 32, j = startTwo / 32;
            uint64_t d = (nw[i] ^ rw[j]) & (~0ull << noLook);
            int bm = d ? __ffsll((long long)d) - 1 : 64;
            int total = bm - noLook;
            for (i--, j--; i >= 0 && j >= 0 && bm == 64; i--, j--) { d = nw[i] ^ rw[j]; bm = d ? __ffsll((long long)d) - 1 : 64; total += bm; }
            left = min(readStart + 1, total / 2);
        }
        const int qStart = readStart - (left - 1), nodeStart = nodePos - (left - 1);
        // extendToTheRightDev's first step: CountRightMatches(node, nodeStart, Rainbow[readOffset'], qStart + readOffset')
        readOffset = 31 - ((qStart - (nodeStart % 32) + 31) % 32);
        rw = rain + (int64_t)readOffset * c.RW;
        int right;
        {
            const int startTwo = qStart + readOffset, two_len = read_len + readOffset;
            const int offBits = (nodeStart % 32) * 2;
            int i = nodeStart / 32, j = startTwo / 32;
            const int iEnd = (node_len + 31) / 32, jEnd = (two_len + 31) / 32;
            uint64_t d = (nw[i] ^ rw[j]) & (~0ull >> offBits);
            int bm = d ? __clzll((long long)d) : 64;
            int total = bm - offBits;
            for (i++, j++; i < iEnd && j < jEnd && bm == 64; i++, j++) { d = nw[i] ^ rw[j]; bm = d ? __clzll((long long)d) : 64; total += bm; }
            right = min(min(total / 2, node_len - nodeStart), two_len - startTwo);
        }
        SeedHit o;
        o.read_start = readStart; o.strand = st; o.node = node; o.node_start = nodeStart; o.q_start = qStart; o.right = right;
        hits[off[x] + h] = o;
    }
}

} // namespace
