// gonomics_genomegraph.hpp -- C++ host mirror of the graph aligner's read path over the C ABI of libgonomics_align_hip.so
// ("next" rows N2 and N4 of SURVEY 8f).  Header-only, C++17.  Same structure and the same parity contract as the Python mirror
// gonomics_amd/genomeGraph.py (see its module docstring: UNPINNED by the reference's tests; Go's shared backing arrays between sibling
// branches are modelled -- CigSlice, SeedSlice --; > 100 seeds sorted stably by TotalLength).
//
//   N2  LeftDynamicAln / RightDynamicAln               /root/reference/genomeGraph/search.go:234-321   -> gnx_gsw_extend_batch
//       LeftAlignTraversal / RightAlignTraversal        search.go:166-232                               -> Traversal (explicit stack)
//       GraphSmithWatermanToGiraf                       genomeGraph/toGiraf.go:17-72                    -> GswBatchToGiraf
//   N4  IndexGenomeIntoMap                              genomeGraph/index.go:21-59                      -> SeedIndex (gnx_seed_index_build)
//       seedMapMemPool, extendToTheRightDev / LeftDev   search.go:425-590                               -> seedMapBatch (gnx_seed_find_batch)
//       dnaTwoBit.CountRightMatches / CountLeftMatches  dna/dnaTwoBit/perfectAlign.go:10-85
//       seedCouldBeBetter                               index.go:102-121
//
// The reference's recursion (a traversal calls the DP at its leaves and hands the route from one sibling branch to the next) is
// turned inside out here: a traversal is a stack machine that stops whenever it needs a DP, so that the DPs of a whole batch of
// reads go to the device together, one gnx_gsw_extend_batch call per side and round.
//
// Worker pool (round 4): the per-read host work between the device calls -- seed continuation across node borders and the seed sort,
// the traversal machines, merging the DP routes -- runs on `threads` host threads, the reads dealt out in chunks (the reference's
// own parallelism: `-t` worker goroutines taking reads off a channel, genomeGraph/routines.go:12-65, cmd/gsw).  Reads are independent
// (every slice a read's branches share belongs to that read), results are in input order and equal the one-thread run's; a GoPanic
// that propagates is the one of the first read (input order) that panics.  threads = 0: GNX_GSW_THREADS, else min(hardware threads, 16).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "gnx_align.h"

namespace gonomics {
namespace genomeGraph {

using Bases = std::vector<uint8_t>; // dna.Base: A C G T N = 0 .. 4

inline int gswThreads(int asked) {
    if (asked > 0) return std::min(asked, 256);
    if (const char *e = getenv("GNX_GSW_THREADS")) { const int v = atoi(e); if (v > 0) return std::min(v, 256); }
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hw, 32u)); // (round 5, tools/gsw_threads.py: 100 000 reads 16 / 32 / 64 / 128 threads = 125 / 88 / 93 / 187 ms per gnx_gsw_map_reads call)
}
// the workers: parked on a condition variable between regions, started on first use, joined at exit
class GswPool {
  public:
    static GswPool &get() { static GswPool p; return p; }
    // job(w) on the caller (w = 0) and on threads - 1 workers (w = 1 ..); returns when all of them are back
    void run(int threads, const std::function<void(int)> &job) {
        std::lock_guard<std::mutex> one(run_mu_); // one region at a time (concurrent callers take turns)
        {
            std::unique_lock<std::mutex> lk(mu_);
            while ((int)th_.size() < threads - 1) { const int id = (int)th_.size(); th_.emplace_back([this, id]() { loop(id); }); }
            job_ = &job; want_ = threads - 1; pending_ = std::min((int)th_.size(), threads - 1); gen_++;
        }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&]() { return pending_ == 0; });
        job_ = nullptr;
    }
    ~GswPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }

  private:
    std::vector<std::thread> th_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *job_ = nullptr;
    uint64_t gen_ = 0;
    int want_ = 0, pending_ = 0;
    bool stop_ = false;
    void loop(int id) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&]() { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            if (id >= want_) continue; // (more workers than this region asked for)
            const std::function<void(int)> *job = job_;
            lk.unlock();
            (*job)(id + 1);
            lk.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
};
// f(k) for k in [0, n) on `threads` threads; the exception of the lowest k is rethrown.  blocks = false: chunks handed out by a counter;
// blocks = true: worker w takes the w-th of `threads` equal blocks -- the regions that run over ALL reads use it, so that a read's
// seeds, task and result are allocated and freed by the same thread (malloc arenas: freeing another thread's blocks contends)
template <class F> inline void parallelFor(size_t n, int threads, F &&f, bool blocks = false) {
    if (threads <= 1 || n < 32) { for (size_t k = 0; k < n; k++) f(k); return; }
    const size_t chunk = std::max<size_t>(4, n / ((size_t)threads * 16));
    std::atomic<size_t> next{0};
    std::mutex mu;
    std::exception_ptr err;
    size_t err_at = n;
    auto one = [&](size_t k) {
        try { f(k); }
        catch (...) { std::lock_guard<std::mutex> lk(mu); if (k < err_at) { err_at = k; err = std::current_exception(); } }
    };
    const std::function<void(int)> work = [&](int w) {
        if (blocks) {
            for (size_t k = n * (size_t)w / (size_t)threads; k < n * ((size_t)w + 1) / (size_t)threads; k++) one(k);
            return;
        }
        for (;;) {
            const size_t k0 = next.fetch_add(chunk);
            if (k0 >= n) return;
            for (size_t k = k0; k < std::min(n, k0 + chunk); k++) one(k);
        }
    };
    GswPool::get().run(threads, work);
    if (err) std::rethrow_exception(err);
}
// wall clock of the stages of one GswBatchToGiraf call (ms), filled when the caller passes one
struct GswTimings {
    double seed_device = 0, seed_host = 0, tasks = 0, dp_pack = 0, dp_device = 0, dp_merge = 0, advance = 0, finish = 0;
    int threads = 0;
};
inline double gswNow() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// a failed library call, with its return code (so that a device failure or an allocation failure does not come out as a caller error)
struct GnxFailure : std::runtime_error {
    int rc;
    GnxFailure(int r, const std::string &what) : std::runtime_error(what), rc(r) {}
};
inline void gnxCheck(int rc) {
    if (rc != GNX_OK) throw GnxFailure(rc, std::string("libgonomics_align_hip: ") + gnx_last_error());
}

// a situation in which the Go code panics (the process dies): reported, never papered over
struct GoPanic : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- cigar.Cigar (cigar/cigar.go:15-35): SAM-style ops as bytes ------------------------------------------------------------
struct Cigar {
    int64_t RunLength;
    uint8_t Op; // 'M', 'I', 'D', 'S'
    bool operator==(const Cigar &o) const { return RunLength == o.RunLength && Op == o.Op; }
};
inline uint8_t opFromCol(uint8_t col) { return col == GNX_COL_M ? 'M' : (col == GNX_COL_I ? 'I' : 'D'); }

// ---- dnaTwoBit (dna/dnaTwoBit/dnaTwoBit.go:10-13, 66-76; rainbow.go:27-45; perfectAlign.go) -------------------------------
struct TwoBit {
    std::vector<uint64_t> Seq;
    int Len = 0;
    TwoBit() = default;
    explicit TwoBit(const Bases &b) : Len((int)b.size()) {
        for (size_t start = 0; start < b.size(); start += 32) {
            uint64_t w = 0;
            const size_t cnt = std::min<size_t>(32, b.size() - start);
            for (size_t x = 0; x < cnt; x++) w = (w << 2) | (uint64_t)b[start + x]; // BasesToUint64LeftAln: an N (4) also sets the low bit of the base before it
            if (cnt < 32) w <<= 2 * (32 - cnt);
            Seq.push_back(w);
        }
    }
};
inline std::vector<TwoBit> NewTwoBitRainbow(const Bases &bases) {
    std::vector<TwoBit> out;
    Bases clone = bases;
    for (int k = 0; k < 32; k++) {
        out.emplace_back(clone);
        clone.insert(clone.begin(), (uint8_t)0);
    }
    return out;
}
inline int lz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
inline int tz64(uint64_t x) { return x ? __builtin_ctzll(x) : 64; }
inline int CountRightMatches(const TwoBit &one, int startOne, const TwoBit &two, int startTwo) {
    const int offsetOne = (startOne % 32) * 2, offsetTwo = (startTwo % 32) * 2;
    if (offsetOne != offsetTwo) throw std::runtime_error("Error: Different offsets when comparing sequences");
    int i = startOne / 32, j = startTwo / 32;
    const int iEnd = (one.Len + 31) / 32, jEnd = (two.Len + 31) / 32;
    const uint64_t seqDiff = (one.Seq[i] ^ two.Seq[j]) & (~(uint64_t)0 >> offsetOne);
    int bitMatches = lz64(seqDiff);
    int total = bitMatches - offsetOne;
    i++; j++;
    while (i < iEnd && j < jEnd && bitMatches == 64) {
        bitMatches = lz64(one.Seq[i] ^ two.Seq[j]);
        total += bitMatches;
        i++; j++;
    }
    return std::min(total / 2, std::min(one.Len - startOne, two.Len - startTwo));
}
inline int CountLeftMatches(const TwoBit &one, int startOne, const TwoBit &two, int startTwo) {
    const int offsetOne = (startOne % 32) * 2, offsetTwo = (startTwo % 32) * 2;
    if (offsetOne != offsetTwo) throw std::runtime_error("Different offsets when comparing sequences");
    const int firstBitsNoLook = 64 - offsetOne - 2;
    int i = startOne / 32, j = startTwo / 32;
    const uint64_t seqDiff = (one.Seq[i] ^ two.Seq[j]) & (firstBitsNoLook >= 64 ? 0 : (~(uint64_t)0 << firstBitsNoLook));
    int bitMatches = tz64(seqDiff);
    int total = bitMatches - firstBitsNoLook;
    i--; j--;
    while (i >= 0 && j >= 0 && bitMatches == 64) {
        bitMatches = tz64(one.Seq[i] ^ two.Seq[j]);
        total += bitMatches;
        i--; j--;
    }
    return total / 2;
}
inline int GetBase(const TwoBit &frag, int pos) { return (int)((frag.Seq[pos / 32] >> (64 - 2 * (pos % 32 + 1))) & 3); }

// ---- the graph (genomeGraph/genomeGraph.go:25-47) and reads (fastq/fastqBig.go:15-50) ---------------------------------------
struct Node;
struct Edge {
    Node *Dest;
    float Prob;
};
struct Node {
    uint32_t Id = 0;
    Bases Seq;
    TwoBit SeqTwoBit;
    std::vector<Edge> Prev, Next;
};
struct GenomeGraph {
    std::vector<std::unique_ptr<Node>> Nodes;
    Node *AddNode(const Bases &seq) {
        auto n = std::make_unique<Node>();
        n->Id = (uint32_t)Nodes.size();
        n->Seq = seq;
        n->SeqTwoBit = TwoBit(seq);
        Nodes.push_back(std::move(n));
        return Nodes.back().get();
    }
    static void AddEdge(Node *u, Node *v, float p = 1.0f) {
        u->Next.push_back(Edge{v, p});
        v->Prev.push_back(Edge{u, p});
    }
};
struct FastqBig {
    std::string Name;
    Bases Seq, SeqRc;
    std::vector<TwoBit> Rainbow, RainbowRc;
    FastqBig(std::string name, const Bases &seq) : Name(std::move(name)), Seq(seq), SeqRc(seq.size()) {
        static const uint8_t comp[5] = {3, 2, 1, 0, 4};
        for (size_t k = 0; k < seq.size(); k++) SeqRc[seq.size() - 1 - k] = comp[seq[k]];
    }
    void rainbows() {
        if (Rainbow.empty()) { Rainbow = NewTwoBitRainbow(Seq); RainbowRc = NewTwoBitRainbow(SeqRc); }
    }
};

// ---- N4: the index ------------------------------------------------------------------------------------------------------------
inline uint64_t ChromAndPosToNumber(uint64_t chrom, uint64_t start) { return (chrom << 32) | start; }
inline uint64_t dnaToNumber(const Bases &seq, size_t start, size_t end) { // align.go:170-177 (`answer << 2 | base`)
    uint64_t ans = seq[start];
    for (size_t i = start + 1; i < end; i++) ans = (ans << 2) | (uint64_t)seq[i];
    return ans;
}
inline void packNodes(const GenomeGraph &g, Bases &cat, std::vector<int64_t> &off) {
    off.assign(1, 0);
    cat.clear();
    for (const auto &n : g.Nodes) { cat.insert(cat.end(), n->Seq.begin(), n->Seq.end()); off.push_back((int64_t)cat.size()); }
    cat.push_back(0); // (never an empty buffer)
}
// IndexGenomeIntoMap (index.go:21-59) as two arrays sorted by key, equal keys in the reference's insertion order: the k-mers inside
// nodes from the device, the few that run across node borders from the host recursion over Next edges (indexGenomeIntoMapHelper).
struct SeedIndex {
    int seedLen = 0, seedStep = 0;
    std::vector<uint64_t> keys, locs;
    // the library holds ONE resident index; every set gives it a new generation, and a search names the generation it expects
    // (gnx_seed_index_set_gen / gnx_seed_find_batch_gen: checked inside the library's lock, whoever else sets an index in between)
    mutable std::mutex residentMu;
    mutable uint64_t residentGen = 0;              // the library's generation of this index' last upload (0: never uploaded)
    mutable const void *residentGraph = nullptr;
    SeedIndex(const GenomeGraph &g, int seed_len, int seed_step) : seedLen(seed_len), seedStep(seed_step) {
        if (seed_len < 2 || seed_len > 32) throw std::runtime_error("Error: seed length needs to be greater than 1 and less than 33.");
        Bases cat;
        std::vector<int64_t> off;
        packNodes(g, cat, off);
        uint64_t *k = nullptr, *l = nullptr;
        int64_t n = 0;
        gnxCheck(gnx_seed_index_build(cat.data(), off.data(), (int64_t)g.Nodes.size(), seed_len, seed_step, &k, &l, &n));
        keys.assign(k, k + n);
        locs.assign(l, l + n);
        gnx_free(k);
        gnx_free(l);
        std::vector<std::pair<uint64_t, uint64_t>> border;
        for (size_t nodeIdx = 0; nodeIdx < g.Nodes.size(); nodeIdx++) {
            const Node &nd = *g.Nodes[nodeIdx];
            const int64_t L = (int64_t)nd.Seq.size();
            const int64_t first = (L - seed_len + 1 <= 0) ? 0 : ((L - seed_len) / seed_step + 1) * seed_step;
            for (int64_t pos = first; pos < L; pos += seed_step)
                for (const Edge &e : nd.Next) helper(Bases(nd.Seq.begin() + pos, nd.Seq.end()), *e.Dest, ChromAndPosToNumber(nodeIdx, (uint64_t)pos), border);
        }
        if (!border.empty()) { // a location code (node << 32 | pos) sorts like the reference's insertion order within one key
            std::vector<std::pair<uint64_t, uint64_t>> all(keys.size());
            for (size_t x = 0; x < keys.size(); x++) all[x] = {keys[x], locs[x]};
            all.insert(all.end(), border.begin(), border.end());
            std::stable_sort(all.begin(), all.end());
            keys.resize(all.size());
            locs.resize(all.size());
            for (size_t x = 0; x < all.size(); x++) { keys[x] = all[x].first; locs[x] = all[x].second; }
        }
    }

  private:
    void helper(Bases prevSeq, const Node &curr, uint64_t loc, std::vector<std::pair<uint64_t, uint64_t>> &out) const {
        if ((int)(prevSeq.size() + curr.Seq.size()) >= seedLen) {
            Bases cs = prevSeq;
            cs.insert(cs.end(), curr.Seq.begin(), curr.Seq.begin() + (seedLen - (int)prevSeq.size()));
            if (std::find(cs.begin(), cs.end(), (uint8_t)4) == cs.end()) out.push_back({dnaToNumber(cs, 0, (size_t)seedLen), loc});
        } else {
            prevSeq.insert(prevSeq.end(), curr.Seq.begin(), curr.Seq.end());
            for (const Edge &e : curr.Next) helper(prevSeq, *e.Dest, loc, out);
        }
    }
};

// ---- N4: seeds (index.go:10-18, search.go:338-370, 425-590) ----------------------------------------------------------------------
struct SeedDev;
using SeedPtr = std::shared_ptr<SeedDev>;
struct SeedDev {
    uint32_t TargetId, TargetStart, QueryStart, Length;
    bool PosStrand;
    uint32_t TotalLength;
    SeedPtr NextPart;
};
inline SeedPtr mkSeed(uint32_t id, uint32_t ts, uint32_t qs, uint32_t len, bool pos, uint32_t total, SeedPtr next = nullptr) {
    return std::make_shared<SeedDev>(SeedDev{id, ts, qs, len, pos, total, std::move(next)});
}
inline const SeedDev *getLastPart(const SeedDev *a) {
    while (a->NextPart) a = a->NextPart.get();
    return a;
}
inline std::vector<uint32_t> getSeedPath(const SeedDev *s) {
    std::vector<uint32_t> p;
    for (; s; s = s->NextPart.get()) p.push_back(s->TargetId);
    return p;
}
// Go's append capacities (runtime/slice.go nextslicecap + roundupsize over the allocator's size classes; go.mod: go 1.25)
inline int64_t goNextCap(int64_t newLen, int64_t oldCap, int64_t elemSize = 16) {
    static const int64_t classes[] = {0, 8, 16, 24, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352, 384, 416, 448, 480, 512, 576, 640,
                                      704, 768, 896, 1024, 1152, 1280, 1408, 1536, 1792, 2048, 2304, 2688, 3072, 3200, 3456, 4096, 4864, 5376, 6144, 6528, 6784, 6912,
                                      8192, 9472, 9728, 10240, 10880, 12288, 13568, 14336, 16384, 18432, 19072, 20480, 21760, 24576, 27264, 28672, 32768};
    int64_t newcap = oldCap;
    if (newLen > 2 * oldCap) newcap = newLen;
    else if (oldCap < 256) newcap = 2 * oldCap;
    else while (newcap < newLen) newcap += (newcap + 3 * 256) >> 2;
    int64_t mem = newcap * elemSize;
    if (mem <= 32768) { for (int64_t c : classes) if (c >= mem) { mem = c; break; } }
    else mem = (mem + 8191) / 8192 * 8192;
    return mem / elemSize;
}
// []SeedDev as extendToTheRightDev uses it (search.go:425-461): the cells are SeedDev objects with identity -- `NextPart: &nextParts[j]`
// points AT a cell -- and the slice is handed back into the call for the next edge of node.Next, where `answer = answer[:0]` + append
// overwrite the cells in place: a seed made for the first edge can end up pointing at the part made for the second one (round 4; rounds
// 1-3 built fresh vectors).  Capacities follow Go's append (SeedDev: 32 bytes).
struct SeedSlice {
    std::vector<SeedPtr> arr; // the backing array: arr.size() == cap
    size_t n = 0;
    bool nil = true;
    SeedSlice append(const SeedDev &v) const {
        SeedSlice s = *this; s.nil = false;
        if (n < arr.size()) { *s.arr[n] = v; s.n = n + 1; return s; } // in place: whoever points at this cell sees the new seed
        const size_t cap = (size_t)goNextCap((int64_t)n + 1, (int64_t)arr.size(), 32);
        s.arr.assign(cap, nullptr);
        for (size_t k = 0; k < cap; k++) s.arr[k] = std::make_shared<SeedDev>(k < n ? *arr[k] : SeedDev{0, 0, 0, 0, true, 0, nullptr});
        *s.arr[n] = v; s.n = n + 1;
        return s;
    }
};
inline SeedSlice extendRightSlices(const Node &node, FastqBig &read, int readStart, int nodeStart, bool posStrand, SeedSlice answer) {
    answer.n = 0; // answer = answer[:0]
    read.rainbows();
    const int nodeOffset = nodeStart % 32;
    const int readOffset = 31 - ((readStart - nodeOffset + 31) % 32);
    const auto &rain = posStrand ? read.Rainbow : read.RainbowRc;
    const int rightMatches = CountRightMatches(node.SeqTwoBit, nodeStart, rain[readOffset], readStart + readOffset);
    if (rightMatches == 0) return SeedSlice(); // nil
    SeedSlice nextParts;
    if (readStart + rightMatches < (int)read.Seq.size() && nodeStart + rightMatches == node.SeqTwoBit.Len && !node.Next.empty()) {
        for (const Edge &e : node.Next) {
            nextParts = extendRightSlices(*e.Dest, read, readStart + rightMatches, 0, posStrand, nextParts);
            for (size_t j = 0; j < nextParts.n; j++)
                answer = answer.append(SeedDev{node.Id, (uint32_t)nodeStart, (uint32_t)readStart, (uint32_t)rightMatches, posStrand, (uint32_t)rightMatches + nextParts.arr[j]->TotalLength, nextParts.arr[j]});
        }
    }
    if (answer.n == 0) { // answer = []SeedDev{currNode}: a fresh array of one
        SeedSlice one;
        one.arr.push_back(std::make_shared<SeedDev>(SeedDev{node.Id, (uint32_t)nodeStart, (uint32_t)readStart, (uint32_t)rightMatches, posStrand, (uint32_t)rightMatches, nullptr}));
        one.n = 1; one.nil = false;
        return one;
    }
    return answer;
}
// the seeds (value copies, as `append(finalSeeds, tempSeeds...)` takes them) that start at (node, nodeStart) / readStart and run to the right
inline std::vector<SeedPtr> extendToTheRightDev(const Node &node, FastqBig &read, int readStart, int nodeStart, bool posStrand) {
    const SeedSlice sl = extendRightSlices(node, read, readStart, nodeStart, posStrand, SeedSlice());
    std::vector<SeedPtr> out;
    for (size_t k = 0; k < sl.n; k++) out.push_back(std::make_shared<SeedDev>(*sl.arr[k]));
    return out;
}
inline std::vector<SeedPtr> leftHelper(const Node &node, FastqBig &read, const SeedPtr &nextPart) {
    read.rainbows();
    const auto &rain = nextPart->PosStrand ? read.Rainbow : read.RainbowRc;
    const int nodePos = node.SeqTwoBit.Len - 1;
    const int readPos = (int)nextPart->QueryStart - 1;
    const int nodeOffset = nodePos % 32;
    const int readOffset = 31 - ((readPos - nodeOffset + 31) % 32);
    const int leftMatches = std::min(readPos + 1, CountLeftMatches(node.SeqTwoBit, nodePos, rain[readOffset], readPos + readOffset));
    if (leftMatches == 0) throw std::runtime_error("Error: should not have zero matches to the left");
    SeedPtr currPart = mkSeed(node.Id, nodePos - (leftMatches - 1), readPos - (leftMatches - 1), leftMatches, nextPart->PosStrand, leftMatches + nextPart->TotalLength, nextPart);
    std::vector<SeedPtr> answer;
    if (currPart->QueryStart > 0 && currPart->TargetStart == 0) {
        for (const Edge &e : node.Prev) {
            const int readBase = GetBase(rain[0], (int)currPart->QueryStart - 1);
            if (readBase == GetBase(e.Dest->SeqTwoBit, e.Dest->SeqTwoBit.Len - 1)) {
                auto more = leftHelper(*e.Dest, read, currPart);
                answer.insert(answer.end(), more.begin(), more.end());
            }
        }
    }
    if (answer.empty()) answer.push_back(currPart);
    return answer;
}
inline std::vector<SeedPtr> extendToTheLeftDev(const Node &node, FastqBig &read, const SeedPtr &currPart) {
    std::vector<SeedPtr> answer;
    if (currPart->QueryStart > 0 && currPart->TargetStart == 0 && !node.Prev.empty()) {
        read.rainbows(); // (only a seed that starts at its node's first base with read to spare needs the shifted copies: most reads never do)
        const auto &rain = currPart->PosStrand ? read.Rainbow : read.RainbowRc;
        for (const Edge &e : node.Prev) {
            const int readBase = GetBase(rain[0], (int)currPart->QueryStart - 1);
            if (readBase == GetBase(e.Dest->SeqTwoBit, e.Dest->SeqTwoBit.Len - 1)) {
                auto more = leftHelper(*e.Dest, read, currPart);
                answer.insert(answer.end(), more.begin(), more.end());
            }
        }
    }
    if (answer.empty()) answer.push_back(currPart);
    return answer;
}
inline void heapSortSeeds(std::vector<SeedPtr> &a) { // search.go:338-370, literally
    auto heapify = [&](int n, int i) {
        while (true) {
            const int l = 2 * i + 1, r = 2 * i + 2;
            int mx = (l < n && a[l]->TotalLength < a[i]->TotalLength) ? l : i;
            if (r < n && a[r]->TotalLength < a[mx]->TotalLength) mx = r;
            if (mx == i) return;
            std::swap(a[i], a[mx]);
            i = mx;
        }
    };
    const int n = (int)a.size();
    for (int i = n / 2 - 1; i >= 0; i--) heapify(n, i);
    int size = n;
    for (int i = n - 1; i > 0; i--) {
        std::swap(a[0], a[i]);
        size--;
        heapify(size, 0);
    }
}
inline void sortSeeds(std::vector<SeedPtr> &seeds) { // seedMapMemPool's tail (search.go:583-589); > 100: see the parity contract
    if (seeds.size() > 100) std::stable_sort(seeds.begin(), seeds.end(), [](const SeedPtr &x, const SeedPtr &y) { return x->TotalLength > y->TotalLength; });
    else heapSortSeeds(seeds);
}
// seedMapMemPool for a batch of reads: hash lookups and in-node exact-match extensions on the device, continuation into
// neighbouring nodes here
inline std::vector<std::vector<SeedPtr>> seedMapBatch(const SeedIndex &index, const GenomeGraph &g, std::vector<FastqBig> &reads, int threads = 1, GswTimings *tm = nullptr) {
    const double t0 = gswNow();
    Bases rcat;
    std::vector<int64_t> roff(reads.size() + 1, 0);
    for (size_t r = 0; r < reads.size(); r++) roff[r + 1] = roff[r] + (int64_t)reads[r].Seq.size();
    rcat.resize((size_t)roff.back() + 1, 0);
    parallelFor(reads.size(), threads, [&](size_t r) { if (!reads[r].Seq.empty()) memcpy(rcat.data() + roff[r], reads[r].Seq.data(), reads[r].Seq.size()); });
    gnx_seed_hit *hits = nullptr;
    int64_t *hoff = nullptr;
    {
        std::lock_guard<std::mutex> lk(index.residentMu);
        for (int attempt = 0;; attempt++) {
            if (index.residentGen == 0 || index.residentGraph != (const void *)&g) { // (a later batch against the same index: it is still there -- the search will say if not)
                Bases ncat;
                std::vector<int64_t> noff;
                packNodes(g, ncat, noff);
                uint64_t gen = 0;
                gnxCheck(gnx_seed_index_set_gen(index.keys.data(), index.locs.data(), (int64_t)index.keys.size(), ncat.data(), noff.data(), (int64_t)g.Nodes.size(), index.seedLen, &gen));
                index.residentGen = gen; index.residentGraph = (const void *)&g;
            }
            const int rc = gnx_seed_find_batch_gen(index.residentGen, rcat.data(), roff.data(), (int64_t)reads.size(), &hits, &hoff);
            if (rc == GNX_ESTALE && attempt < 8) { index.residentGen = 0; continue; } // somebody else's index is on the device now: upload again
            gnxCheck(rc);
            break;
        }
    }
    for (int64_t h = 0, hn = hoff[reads.size()]; h < hn; h++) // (the hits index g.Nodes below)
        if (hits[h].node < 0 || (size_t)hits[h].node >= g.Nodes.size()) { gnx_free(hits); gnx_free(hoff); throw std::runtime_error("seed hit outside the graph: the resident index is not this graph's"); }
    std::vector<std::vector<SeedPtr>> out(reads.size());
    const double t1 = gswNow();
    parallelFor(reads.size(), threads, [&](size_t r) {
        FastqBig &read = reads[r];
        std::vector<SeedPtr> fin;
        for (int64_t h = hoff[r]; h < hoff[r + 1]; h++) {
            const gnx_seed_hit &x = hits[h];
            const Node &node = *g.Nodes[(size_t)x.node];
            const bool pos = x.strand == 0;
            if (x.right == 0) continue; // (inner loop)
            std::vector<SeedPtr> temp;
            if (x.q_start + x.right < (int)read.Seq.size() && x.node_start + x.right == node.SeqTwoBit.Len && !node.Next.empty())
                temp = extendToTheRightDev(node, read, x.q_start, x.node_start, pos); // crosses into the next node(s)
            else temp.push_back(mkSeed(node.Id, (uint32_t)x.node_start, (uint32_t)x.q_start, (uint32_t)x.right, pos, (uint32_t)x.right));
            if (pos) {
                for (const SeedPtr &t : temp) {
                    auto more = extendToTheLeftDev(node, read, t);
                    fin.insert(fin.end(), more.begin(), more.end());
                }
            } else fin.insert(fin.end(), temp.begin(), temp.end()); // (the reference does not extend minus-strand seeds to the left across nodes)
        }
        sortSeeds(fin);
        out[r] = std::move(fin);
    }, /*blocks=*/true);
    gnx_free(hits);
    gnx_free(hoff);
    if (tm) { tm->seed_device += t1 - t0; tm->seed_host += gswNow() - t1; }
    return out;
}
inline bool seedCouldBeBetter(int64_t seedLen, int64_t currBestScore, int64_t perfectScore, int64_t queryLen, int64_t maxMatch, int64_t minMatch,
                              int64_t leastSevereMismatch, int64_t leastSevereMatchMismatchChange) { // index.go:102-121
    const int64_t seeds = queryLen / (seedLen + 1), remainder = queryLen % (seedLen + 1);
    if (seedLen * maxMatch >= currBestScore && perfectScore - ((queryLen - seedLen) * minMatch) >= currBestScore) return true;
    if (seedLen * seeds * maxMatch + seeds * leastSevereMismatch >= currBestScore &&
        perfectScore - remainder * minMatch + seeds * leastSevereMatchMismatchChange >= currBestScore) return true;
    if (seedLen * seeds * maxMatch + remainder * maxMatch + (seeds + 1) * leastSevereMismatch >= currBestScore &&
        perfectScore + (seeds + 1) * leastSevereMatchMismatchChange >= currBestScore) return true;
    return false;
}

// ---- Go slices ----------------------------------------------------------------------------------------------------------------------
// LeftAlignTraversal / RightAlignTraversal hand ONE route slice from sibling to sibling, keep headers of it (sk.leftAlignment =
// dynamicScore.route), reverse it in place, and GraphSmithWatermanToGiraf appends to it in place (cigar.Append / Concat): what a later
// sibling's DP writes through the shared backing array shows through every header that still points into it, until an append outgrows
// the capacity and moves to a new array.  Rounds 1-3 copied vectors ("value semantics") and differed from the Go program on ~12 % of
// the reads of a variant graph (VERDICT r3 missing 1).  The model: (backing array, offset, len, cap) + Go 1.25's growth rule
// (runtime/slice.go nextslicecap + roundupsize over the allocator's size classes; go.mod: go 1.25; cigar.Cigar is 16 bytes).
class CigSlice { // []cigar.Cigar as Go sees it
  public:
    CigSlice() = default;
    static CigSlice make(int64_t len, int64_t cap) { CigSlice s; s.arr_ = std::make_shared<std::vector<Cigar>>((size_t)cap); s.n_ = len; s.cap_ = cap; return s; }
    int64_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    Cigar &operator[](int64_t i) const { return (*arr_)[(size_t)(off_ + i)]; } // (through the shared array, also from a const header)
    CigSlice tail(int64_t k) const { CigSlice s = *this; s.off_ += k; s.n_ -= k; s.cap_ -= k; return s; }
    // append(s, v...): in place while the capacity lasts, else a new array (old elements copied) of Go's next capacity
    CigSlice append(const Cigar *v, int64_t cnt) const {
        const int64_t need = n_ + cnt;
        if (need <= cap_) { for (int64_t k = 0; k < cnt; k++) (*arr_)[(size_t)(off_ + n_ + k)] = v[k]; CigSlice s = *this; s.n_ = need; return s; }
        CigSlice s = make(need, goNextCap(need, cap_));
        for (int64_t k = 0; k < n_; k++) (*s.arr_)[(size_t)k] = (*this)[k];
        for (int64_t k = 0; k < cnt; k++) (*s.arr_)[(size_t)(n_ + k)] = v[k];
        return s;
    }
    CigSlice append(const Cigar &v) const { return append(&v, 1); }
    CigSlice append(const CigSlice &o) const { std::vector<Cigar> tmp = o.toVector(); return append(tmp.data(), (int64_t)tmp.size()); } // (the values first: o may share the array)
    void reverse() const { for (int64_t i = 0, j = n_ - 1; i < n_ / 2; i++, j--) std::swap((*this)[i], (*this)[j]); } // cigar.ReverseCigar: in place
    std::vector<Cigar> toVector() const { std::vector<Cigar> v((size_t)n_); for (int64_t k = 0; k < n_; k++) v[(size_t)k] = (*this)[k]; return v; }

  private:
    std::shared_ptr<std::vector<Cigar>> arr_;
    int64_t off_ = 0, n_ = 0, cap_ = 0;
};

// ---- N2: the DPs, batched ---------------------------------------------------------------------------------------------------------
struct DpRequest {
    int side = GNX_GSW_LEFT;
    Bases target;
    const uint8_t *read = nullptr;
    size_t readLen = 0;
    CigSlice route; // dynamicScore.route on entry (carried over from the sibling branch before, search.go:104-107)
};
struct DpResult {
    int64_t score = 0;
    CigSlice route;
    int64_t i = 0, j = 0;
};
// the route-building loop of search.go:252-262 / 298-308 applied to the traced runs (traceback order): increments go through the
// shared array of `route`, appends follow Go's capacities
inline CigSlice mergeRoute(CigSlice route, const gnx_cigar *runs, int64_t nRuns) {
    int64_t idx = 0;
    for (int64_t k = 0; k < nRuns; k++) {
        const uint8_t op = opFromCol(runs[k].op);
        for (int64_t x = 0; x < runs[k].run_length; x++) {
            if (route.empty()) route = route.append(Cigar{1, op});
            else if (route[idx].Op == op) route[idx].RunLength++;
            else { route = route.append(Cigar{1, op}); idx++; }
        }
    }
    return route;
}
// What runs a batch of extension DPs: the library (gnx_gsw_extend_batch), and nothing else in a product build.  Only a translation unit
// compiled with -DGNX_TEST_BACKEND (tests/cpp/gsw_mirror_test.cpp: the CPU restatement behind the same read path, for the baseline
// beside it) gets a settable function pointer; without the macro there is no seam to set.
using GswExtendFn = int (*)(int, const int64_t *, int64_t, int64_t, const uint8_t *, const int64_t *, const uint8_t *, const int64_t *,
                            int64_t *, int64_t *, int64_t *, gnx_cigar **, int64_t **);
#ifdef GNX_TEST_BACKEND
inline GswExtendFn &gswExtendBackend() { static GswExtendFn f = gnx_gsw_extend_batch; return f; }
#else
inline GswExtendFn gswExtendBackend() { return gnx_gsw_extend_batch; }
#endif
// LeftDynamicAln / RightDynamicAln (search.go:234-321) for a batch of requests of one side
inline std::vector<DpResult> DynamicAlnBatch(int side, const std::vector<const DpRequest *> &reqs, const int64_t *scores25, int64_t gapPen, int threads = 1, GswTimings *tm = nullptr) {
    const size_t n = reqs.size();
    const double t0 = gswNow();
    std::vector<int64_t> aoff(n + 1, 0), boff(n + 1, 0);
    for (size_t p = 0; p < n; p++) { aoff[p + 1] = aoff[p] + (int64_t)reqs[p]->target.size(); boff[p + 1] = boff[p] + (int64_t)reqs[p]->readLen; }
    Bases acat((size_t)aoff[n] + 1, 0), bcat((size_t)boff[n] + 1, 0);
    parallelFor(n, threads, [&](size_t p) {
        if (!reqs[p]->target.empty()) memcpy(acat.data() + aoff[p], reqs[p]->target.data(), reqs[p]->target.size());
        if (reqs[p]->readLen) memcpy(bcat.data() + boff[p], reqs[p]->read, reqs[p]->readLen);
    });
    std::vector<int64_t> sc(n ? n : 1), ei(n ? n : 1), ej(n ? n : 1);
    gnx_cigar *ops = nullptr;
    int64_t *off = nullptr;
    const double t1 = gswNow();
    gnxCheck(gswExtendBackend()(side, scores25, gapPen, (int64_t)n, acat.data(), aoff.data(), bcat.data(), boff.data(), sc.data(), ei.data(), ej.data(), &ops, &off));
    const double t2 = gswNow();
    std::vector<DpResult> out(n);
    parallelFor(n, threads, [&](size_t p) {
        out[p].score = sc[p]; out[p].i = ei[p]; out[p].j = ej[p];
        out[p].route = mergeRoute(reqs[p]->route, ops + off[p], off[p + 1] - off[p]);
    });
    gnx_free(ops);
    gnx_free(off);
    if (tm) { tm->dp_pack += t1 - t0; tm->dp_device += t2 - t1; tm->dp_merge += gswNow() - t2; }
    return out;
}

// ---- N2: LeftAlignTraversal / RightAlignTraversal (search.go:166-232) as a stack machine ---------------------------------------
struct TraversalResult {
    CigSlice aln;
    int64_t score = 0, t = 0, q = 0; // left: targetStart, queryStart; right: targetEnd, queryEnd
    std::vector<uint32_t> path;
};
class Traversal {
  public:
    DpRequest req;          // valid while advance() returns true
    TraversalResult result; // valid once advance() returned false
    void start(bool left, const Node *n, int64_t pos, int64_t extension, const uint8_t *read, size_t readLen) {
        left_ = left; ext_ = extension; read_ = read; readLen_ = readLen;
        stack_.clear();
        push(n, Bases(), pos, {}, {});
    }
    // res: the answer to `req` (nullptr on the first call).  true: `req` holds the next DP; false: done, see `result`.
    bool advance(const DpResult *res) {
        TraversalResult r;
        bool haveRet = false;
        if (res) { // the top frame is a leaf that was waiting for its DP
            Frame &f = stack_.back();
            r.aln = res->route; r.score = res->score; r.path = f.sPath;
            if (left_) { r.t = f.pos - (int64_t)f.sSeq.size() - (int64_t)f.seq.size() + res->i; r.q = res->j; }
            else { r.t = res->i + f.pos; r.q = res->j; }
            stack_.pop_back();
            haveRet = true;
        } else if (stack_.back().leaf) { issue(stack_.back()); return true; }
        while (true) {
            if (haveRet) {
                if (stack_.empty()) { result = std::move(r); return false; }
                Frame &f = stack_.back();
                f.route = r.aln; // the loop variable `route`: what the next sibling's DP starts from
                if (r.score > f.bestScore) {
                    f.bestScore = r.score; f.haveBest = true;
                    f.best = r;
                    if (left_) f.best.t = f.pos - (int64_t)f.sSeq.size() - (int64_t)f.seq.size() + r.t;
                }
                f.child++;
                haveRet = false;
            }
            Frame &f = stack_.back();
            const auto &edges = left_ ? f.n->Prev : f.n->Next;
            if (f.child < edges.size()) {
                const Node *d = edges[f.child].Dest;
                push(d, f.sSeq, left_ ? (int64_t)d->Seq.size() : 0, f.sPath, f.route);
                if (stack_.back().leaf) { issue(stack_.back()); return true; }
            } else { // every branch tried: hand the best one up (search.go:196-198, 228-231: ReverseCigar; the left one reverses its path too)
                r = std::move(f.best);
                r.score = f.bestScore;
                r.aln.reverse(); // in place: every header into this array sees it
                if (left_) std::reverse(r.path.begin(), r.path.end());
                else r.t += f.pos;
                stack_.pop_back();
                haveRet = true;
            }
        }
    }

  private:
    struct Frame {
        const Node *n;
        Bases seq, sSeq;
        int64_t pos; // refEnd (left) / start (right)
        std::vector<uint32_t> sPath;
        CigSlice route;
        bool leaf;
        size_t child = 0;
        int64_t bestScore = INT64_MIN;
        bool haveBest = false;
        TraversalResult best;
    };
    bool left_ = true;
    int64_t ext_ = 0;
    const uint8_t *read_ = nullptr;
    size_t readLen_ = 0;
    std::vector<Frame> stack_;
    void push(const Node *n, const Bases &seq, int64_t pos, const std::vector<uint32_t> &path, const CigSlice &route) {
        Frame f;
        f.n = n; f.seq = seq; f.pos = pos; f.route = route;
        f.sPath = path; // search.go:174-176 calls AddPath(s.Path, n.Id) and drops its result: the node is never recorded
        const int64_t have = (int64_t)seq.size() + (left_ ? pos : (int64_t)n->Seq.size() - pos);
        int64_t take = std::min(have, ext_) - (int64_t)seq.size();
        if (take < 0) take = 0;
        if (left_) { // getLeftTargetBases (search.go:135-140), the expression as written: refEnd - Min(len(seq)+refEnd, extension) - len(seq)
            // is left-associative, so with bases already collected the slice starts len(seq) further left than the commented-out
            // intent (extension + len(seq) bases of a Prev node), and a negative start is a Go panic
            const int64_t start = pos - std::min(have, ext_) - (int64_t)seq.size();
            if (start < 0 || start > pos) throw GoPanic("runtime error: slice bounds out of range (getLeftTargetBases, search.go:139)");
            f.sSeq.assign(n->Seq.begin() + start, n->Seq.begin() + pos);
            f.sSeq.insert(f.sSeq.end(), seq.begin(), seq.end());
            f.leaf = have >= ext_ || n->Prev.empty();
        } else { // getRightBases (search.go:142-147)
            f.sSeq = seq;
            f.sSeq.insert(f.sSeq.end(), n->Seq.begin() + pos, n->Seq.begin() + pos + take);
            f.leaf = have >= ext_ || n->Next.empty();
        }
        stack_.push_back(std::move(f));
    }
    void issue(const Frame &f) {
        req.side = left_ ? GNX_GSW_LEFT : GNX_GSW_RIGHT;
        req.target = f.sSeq; req.read = read_; req.readLen = readLen_; req.route = f.route;
    }
};

// ---- N2: GraphSmithWatermanToGiraf (toGiraf.go:17-72) ------------------------------------------------------------------------------
struct Giraf { // giraf.Giraf, the fields the function fills (giraf/giraf.go:16-33)
    std::string QName;
    int64_t QStart = 0, QEnd = 0;
    bool PosStrand = true;
    int64_t TStart = 0, TEnd = 0;
    std::vector<uint32_t> Nodes;
    bool hasCigar = false;
    std::vector<Cigar> Cig;
    int64_t AlnScore = 0;
    int MapQ = 255;
    const Bases *Seq = nullptr;
    uint8_t Flag = 0;      // set by WrapPairGirafBatch only (toGiraf.go:130-140)
    bool Panicked = false; // GswBatchToGiraf(.., markPanics = true): the Go code panics on this read (see GoPanic); the other fields are void
    std::string PanicText;
};
inline void AddPath(std::vector<uint32_t> &all, uint32_t p) {
    if (all.empty() || all.back() != p) all.push_back(p);
}
inline std::vector<uint32_t> CatPaths(std::vector<uint32_t> curr, const std::vector<uint32_t> &more) {
    if (more.empty()) return curr;
    if (curr.empty()) return more;
    AddPath(curr, more[0]);
    curr.insert(curr.end(), more.begin() + 1, more.end());
    return curr;
}
inline int64_t queryLength(const CigSlice &c) {
    int64_t s = 0;
    for (int64_t k = 0; k < c.size(); k++) { const Cigar &x = c[k]; if (x.Op == 'M' || x.Op == 'I' || x.Op == 'S' || x.Op == '=' || x.Op == 'X') s += x.RunLength; }
    return s;
}
// cigar.AppendSoftClips (cigar/tools.go:26-40), literally -- including that a front clip without a back clip returns only the clip
inline CigSlice appendSoftClips(int64_t front, int64_t lengthOfRead, const CigSlice &cigs) {
    const int64_t run = queryLength(cigs);
    if (front == 0 && run >= lengthOfRead) return cigs;
    CigSlice answer = CigSlice::make(0, cigs.size() + 2); // make([]Cigar, 0, len(cigars)+2)
    if (front > 0) answer = answer.append(Cigar{front, 'S'});
    if (front + run < lengthOfRead) answer = answer.append(cigs).append(Cigar{lengthOfRead - front - run, 'S'});
    return answer;
}
// cigar.Append (cigar/tools.go:4-11): the last cell is incremented IN the shared array, or beta is appended
inline CigSlice cigAppend(CigSlice alpha, const Cigar &beta) {
    if (!alpha.empty() && alpha[alpha.size() - 1].Op == beta.Op) alpha[alpha.size() - 1].RunLength += beta.RunLength;
    else alpha = alpha.append(beta);
    return alpha;
}
// cigar.Concat (cigar/tools.go:14-23)
inline CigSlice cigConcat(CigSlice alpha, CigSlice beta) {
    if (alpha.empty()) return beta;
    if (!beta.empty()) { alpha = cigAppend(alpha, beta[0]); beta = beta.tail(1); }
    return alpha.append(beta);
}
// one read's loop over its sorted seeds, stopping whenever a traversal needs a DP
class ReadTask {
  public:
    Giraf best;
    DpRequest *req = nullptr; // the pending DP (while advance() returns true)
    ReadTask(const GenomeGraph &g, const FastqBig &read, std::vector<SeedPtr> seeds, const int64_t *scores25)
        : g_(g), read_(read), seeds_(std::move(seeds)), sc_(scores25) {
        best.QName = read.Name; best.Seq = &read.Seq;
        perfect_ = 0;
        for (uint8_t b : read.Seq) perfect_ += sc_[b * 5 + b];
        extension_ = perfect_ / 600 + (int64_t)read.Seq.size();
    }
    bool advance(const DpResult *res) {
        while (true) {
            if (phase_ == 0) { // next seed
                if (si_ >= seeds_.size()) return false;
                seed_ = seeds_[si_].get();
                if (!seedCouldBeBetter(seed_->TotalLength, best.AlnScore, perfect_, (int64_t)read_.Seq.size(), 100, 90, -196, -296)) return false;
                tail_ = getLastPart(seed_);
                currSeq_ = seed_->PosStrand ? &read_.Seq : &read_.SeqRc;
                seedScore_ = 0;
                for (uint32_t x = seed_->QueryStart; x < tail_->QueryStart + tail_->Length; x++) seedScore_ += sc_[(*currSeq_)[x] * 5 + (*currSeq_)[x]];
                if (seed_->TotalLength == currSeq_->size()) {
                    targetStart_ = seed_->TargetStart; targetEnd_ = tail_->TargetStart + tail_->Length; queryStart_ = seed_->QueryStart;
                    currScore_ = seedScore_;
                    finishSeed();
                    continue;
                }
                const int64_t ext = extension_ - seed_->TotalLength;
                trav_.start(true, g_.Nodes[seed_->TargetId].get(), seed_->TargetStart, ext, currSeq_->data(), seed_->QueryStart);
                phase_ = 1;
                if (trav_.advance(nullptr)) { req = &trav_.req; return true; }
                res = nullptr;
            }
            if (phase_ == 1) { // in the left traversal
                if (res && trav_.advance(res)) { req = &trav_.req; return true; }
                leftAln_ = trav_.result.aln; leftScore_ = trav_.result.score; targetStart_ = trav_.result.t; queryStart_ = trav_.result.q; leftPath_ = trav_.result.path;
                const int64_t ext = extension_ - seed_->TotalLength;
                const size_t from = tail_->QueryStart + tail_->Length;
                trav_.start(false, g_.Nodes[tail_->TargetId].get(), tail_->TargetStart + tail_->Length, ext, currSeq_->data() + from, currSeq_->size() - from);
                phase_ = 2;
                if (trav_.advance(nullptr)) { req = &trav_.req; return true; }
                res = nullptr;
            }
            if (phase_ == 2) { // in the right traversal
                if (res && trav_.advance(res)) { req = &trav_.req; return true; }
                rightAln_ = trav_.result.aln; targetEnd_ = trav_.result.t; queryEnd_ = trav_.result.q; rightPath_ = trav_.result.path;
                currScore_ = leftScore_ + seedScore_ + trav_.result.score;
                finishSeed();
                res = nullptr;
            }
        }
    }

  private:
    const GenomeGraph &g_;
    const FastqBig &read_;
    std::vector<SeedPtr> seeds_;
    const int64_t *sc_;
    int64_t perfect_ = 0, extension_ = 0;
    size_t si_ = 0;
    int phase_ = 0;
    const SeedDev *seed_ = nullptr, *tail_ = nullptr;
    const Bases *currSeq_ = nullptr;
    int64_t seedScore_ = 0, currScore_ = 0, leftScore_ = 0;
    // scoreKeeper fields that survive from one seed to the next (resetScoreKeeper gets its argument by value: a no-op): a seed that
    // covers the whole read re-uses the alignments, paths and queryEnd of the seed before it (toGiraf.go:47-51)
    CigSlice leftAln_, rightAln_; // (slice HEADERS: cigar.Append below writes through sk.leftAlignment's own array, and a later whole-read seed re-uses the stale header)
    std::vector<uint32_t> leftPath_, rightPath_;
    int64_t targetStart_ = 0, targetEnd_ = 0, queryStart_ = 0, queryEnd_ = 0;
    Traversal trav_;
    void finishSeed() {
        if (currScore_ > best.AlnScore) {
            best.QStart = queryStart_;
            best.QEnd = (int64_t)seed_->QueryStart + queryStart_ + queryEnd_ + (int64_t)seed_->TotalLength - 1;
            best.PosStrand = seed_->PosStrand;
            best.TStart = targetStart_; best.TEnd = targetEnd_;
            best.Nodes = CatPaths(CatPaths(leftPath_, getSeedPath(seed_)), rightPath_);
            best.Cig = appendSoftClips(queryStart_, (int64_t)currSeq_->size(), cigConcat(cigAppend(leftAln_, Cigar{(int64_t)seed_->TotalLength, 'M'}), rightAln_)).toVector();
            best.hasCigar = true;
            best.AlnScore = currScore_;
            best.Seq = currSeq_;
        }
        si_++;
        phase_ = 0;
    }
};
// GraphSmithWatermanToGiraf for a batch of reads: seeds from the device, then rounds of batched DPs (gapPen: gsw passes -600)
// markPanics: a read on which the Go code panics (getLeftTargetBases with a short Prev node) is marked Panicked and the others go on;
// default: the GoPanic propagates (the Go process would die there)
inline std::vector<Giraf> GswBatchToGiraf(const GenomeGraph &g, std::vector<FastqBig> &reads, const SeedIndex &index, const int64_t *scores25,
                                          int64_t gapPen = -600, int *outRounds = nullptr, bool markPanics = false, int threads = 0, GswTimings *tm = nullptr) {
    const int T = gswThreads(threads);
    if (tm) tm->threads = T;
    auto seeds = seedMapBatch(index, g, reads, T, tm);
    const size_t n = reads.size();
    std::vector<std::unique_ptr<ReadTask>> tasks(n);
    std::vector<char> alive(n, 0); // (one byte per read: the workers write the bytes of their own reads only)
    auto advance = [&](size_t k, const DpResult *res) -> bool {
        try {
            return tasks[k]->advance(res);
        } catch (const GoPanic &gp) {
            if (!markPanics) throw;
            tasks[k]->best.Panicked = true; tasks[k]->best.PanicText = gp.what();
            return false;
        }
    };
    double t0 = gswNow();
    parallelFor(n, T, [&](size_t k) {
        tasks[k] = std::make_unique<ReadTask>(g, reads[k], std::move(seeds[k]), scores25);
        alive[k] = advance(k, nullptr) ? 1 : 0;
    }, /*blocks=*/true);
    if (tm) tm->tasks += gswNow() - t0;
    int rounds = 0;
    size_t n_alive = 0;
    for (size_t k = 0; k < n; k++) n_alive += alive[k];
    while (n_alive > 0) { // a round: the pending left DPs of all reads in one call, then the right ones (incl. those the left answers led to)
        for (int side = GNX_GSW_LEFT; side <= GNX_GSW_RIGHT; side++) {
            std::vector<size_t> ks;
            std::vector<const DpRequest *> rq;
            for (size_t k = 0; k < n; k++)
                if (alive[k] && tasks[k]->req->side == side) { ks.push_back(k); rq.push_back(tasks[k]->req); }
            if (ks.empty()) continue;
            auto outs = DynamicAlnBatch(side, rq, scores25, gapPen, T, tm);
            t0 = gswNow();
            parallelFor(ks.size(), T, [&](size_t y) { if (!advance(ks[y], &outs[y])) alive[ks[y]] = 0; });
            if (tm) tm->advance += gswNow() - t0;
        }
        n_alive = 0;
        for (size_t k = 0; k < n; k++) n_alive += alive[k];
        rounds++;
    }
    if (outRounds) *outRounds = rounds;
    t0 = gswNow();
    std::vector<Giraf> out(n);
    parallelFor(n, T, [&](size_t k) { out[k] = std::move(tasks[k]->best); tasks[k].reset(); }, /*blocks=*/true); // (the workers free what they allocated: seeds, stacks, slices)
    if (tm) tm->finish += gswNow() - t0;
    return out;
}

// WrapPairGiraf (toGiraf.go:117-128) for a batch of read pairs -- reads[2k] / reads[2k+1] are the forward / reverse mate of pair k: both
// mates of every pair go through ONE GswBatchToGiraf call, then setGirafFlags (toGiraf.go:130-140) as written: the forward mate gets
// +8 and +16 twice, the reverse mate no pairing flag, arithmetic in uint8; getGirafFlags :183-192, isProperPairAlign :171-181.
inline uint8_t getGirafFlags(const Giraf &ag) { return (uint8_t)((ag.PosStrand ? 4 : 0) + (ag.AlnScore < 1200 ? 2 : 0)); }
inline bool isProperPairAlign(const Giraf &fwd, const Giraf &rev) {
    const double d = (double)(fwd.TStart - rev.TStart);
    if ((d < 0 ? -d : d) < 10000) {
        if (fwd.TStart < rev.TStart && fwd.PosStrand && !rev.PosStrand) return true;
        if (fwd.TStart > rev.TStart && !fwd.PosStrand && rev.PosStrand) return true;
    }
    return false;
}
inline std::vector<Giraf> WrapPairGirafBatch(const GenomeGraph &g, std::vector<FastqBig> &reads, const SeedIndex &index, const int64_t *scores25,
                                             int64_t gapPen = -600, int *outRounds = nullptr, bool markPanics = false, int threads = 0, GswTimings *tm = nullptr) {
    std::vector<Giraf> res = GswBatchToGiraf(g, reads, index, scores25, gapPen, outRounds, markPanics, threads, tm);
    for (size_t k = 0; k + 1 < res.size(); k += 2) {
        Giraf &fwd = res[k], &rev = res[k + 1];
        if (fwd.Panicked || rev.Panicked) continue;
        fwd.Flag = getGirafFlags(fwd);
        rev.Flag = getGirafFlags(rev);
        fwd.Flag = (uint8_t)(fwd.Flag + 8); fwd.Flag = (uint8_t)(fwd.Flag + 16); fwd.Flag = (uint8_t)(fwd.Flag + 16);
        if (isProperPairAlign(fwd, rev)) { fwd.Flag = (uint8_t)(fwd.Flag + 1); rev.Flag = (uint8_t)(rev.Flag + 1); }
    }
    return res;
}

} // namespace genomeGraph
} // namespace gonomics
