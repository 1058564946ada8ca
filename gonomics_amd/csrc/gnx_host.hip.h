// gnx_host.hip.h -- host-buffer entry points: pipelined sub-batches, resident reference, sharding over several GPUs of one node
// Part of libgonomics_align_hip.so; included by gnx_align.hip (one translation unit).  See DESIGN.md sections 4.7 and 6.
//
// What a cgo shim calls: host buffers in, host buffers out.  Pairs are independent, so a batch is cut twice:
//   * across the contexts (one per GPU, gnx_init_devices) into contiguous blocks of equal DP cells -- one worker thread per
//     context, no data-path exchange; the shared beta buffer / the resident reference is uploaded once to device 0 and broadcast
//     over RCCL (xGMI), results are gathered on device 0 with grouped ncclSend / ncclRecv and leave through one D2H;
//   * inside a context into sub-batches: while sub-batch k is in the kernels, a stager thread copies the reads of k+1 into
//     pinned memory and starts their H2D on a second stream.  Results accumulate on the device and leave once.
// RCCL is loaded with dlopen on first multi-GPU use: a single-GPU process never touches it.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <chrono>
#include <map>
#include <thread>
#include <unordered_map>

namespace {

// ---- pinned result buffers handed to the caller (gnx_free gives them back to the pool) -----------------------------------------
struct PinPool {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;
    std::multimap<size_t, void *> idle;
    size_t idle_bytes = 0;
    void *get(size_t bytes) {
        size_t cap = 4096;
        while (cap < bytes) cap <<= 1;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = idle.lower_bound(cap);
            if (it != idle.end() && it->first <= 2 * cap) {
                void *p = it->second;
                live[p] = it->first; idle_bytes -= it->first; idle.erase(it);
                return p;
            }
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        live[p] = cap;
        return p;
    }
    bool put(void *p) { // false: not one of ours
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) return false;
        const size_t cap = it->second;
        live.erase(it);
        if (idle_bytes + cap > ((size_t)2 << 30)) { (void)hipHostFree(p); return true; }
        idle.emplace(cap, p); idle_bytes += cap;
        return true;
    }
    void drain() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &kv : idle) (void)hipHostFree(kv.second);
        idle.clear(); idle_bytes = 0;
    }
};
PinPool g_pool;

// ---- RCCL, loaded on demand -----------------------------------------------------------------------------------------------------
struct RcclApi {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr; // optional: used instead of CommDestroy after a failure
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::vector<ncclComm_t> comms; // one per context when active
};
RcclApi g_rccl;
std::mutex g_api_mu;     // the sharded entry points and the life-cycle calls run one at a time
int g_nctx = 1;          // contexts in use (gnx_init_devices)
bool g_shared_dev = false; // several contexts on one device (flow tests on a 1-GPU box): copies instead of RCCL

int rccl_load() {
    if (g_rccl.h) return GNX_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_err("cannot load librccl.so (%s)", dlerror()); return GNX_EDEVICE; }
#define GNX_SYM(field, name) do { g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); if (!g_rccl.field) { set_err("librccl.so lacks %s", name); dlclose(h); return GNX_EDEVICE; } } while (0)
    GNX_SYM(CommInitAll, "ncclCommInitAll"); GNX_SYM(CommDestroy, "ncclCommDestroy"); GNX_SYM(Broadcast, "ncclBroadcast");
    GNX_SYM(Send, "ncclSend"); GNX_SYM(Recv, "ncclRecv"); GNX_SYM(GroupStart, "ncclGroupStart"); GNX_SYM(GroupEnd, "ncclGroupEnd");
    GNX_SYM(GetErrorString, "ncclGetErrorString");
#undef GNX_SYM
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
    g_rccl.h = h;
    return GNX_OK;
}
#define RCCLCHK(call)                                                                                                     \
    do {                                                                                                                  \
        ncclResult_t r_ = (call);                                                                                         \
        if (r_ != ncclSuccess) { set_err("RCCL error: %s", g_rccl.GetErrorString(r_)); return GNX_EDEVICE; }              \
    } while (0)
bool g_rccl_broken = false; // a RCCL call failed in this process: every later exchange uses peer copies (transport 3)
int g_transport = 0;        // what carried the last broadcast / gather, see gnx_timing.transport
double g_bcast_ms = 0;      // broadcast time of the current call
bool rccl_active() { return !g_rccl.comms.empty() && !g_rccl_broken; }
// drop the communicators (after a failure: abort, a destroy may wait for the peers of a collective that never completed)
void rccl_drop_comms() {
    for (ncclComm_t cm : g_rccl.comms) {
        if (!cm) continue;
        if (g_rccl_broken && g_rccl.CommAbort) (void)g_rccl.CommAbort(cm);
        else (void)g_rccl.CommDestroy(cm);
    }
    g_rccl.comms.clear();
}
// after a failed RCCL call (the caller has already closed any open group): remember the text, drain whatever was enqueued, abort
// the communicators, and keep RCCL off until gnx_shutdown -- every later exchange of the process uses peer copies (transport 3)
void rccl_give_up() {
    g_rccl_broken = true;
    fprintf(stderr, "[gnx] %s -- falling back to peer copies until gnx_shutdown\n", g_err);
    for (int d = 0; d < g_nctx; d++) { Ctx &c = ctx_at(d); if (c.inited && hipSetDevice(c.device) == hipSuccess) (void)hipDeviceSynchronize(); }
    (void)hipGetLastError();
    rccl_drop_comms();
    if (ctx_at(0).inited) (void)hipSetDevice(ctx_at(0).device);
}

// the synthetic reference is a pure function of the position (splitmix64), 2 bits per base, an N run of
// 1000 bases every 5e7 -- generated where it is used instead of crossing PCIe
__device__ __host__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// out[0 .. n) = bases [pos0, pos0 + n) of the synthetic reference (pos0 a multiple of 32)
__global__ __launch_bounds__(256) void synth_ref_kernel(uint8_t *__restrict__ out, int64_t pos0, int64_t n, uint64_t seed) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // 32 bases per thread: one 64-bit draw
    const int64_t p0 = w * 32;
    if (p0 >= n) return;
    const uint64_t bits = splitmix64(seed ^ (uint64_t)(pos0 / 32 + w));
    for (int k = 0; k < 32 && p0 + k < n; k++) {
        const int64_t pos = pos0 + p0 + k;
        out[p0 + k] = (pos % 50000000 < 1000 && pos >= 50000000) ? 4 : (uint8_t)((bits >> (2 * k)) & 3u);
    }
}

// ---- the resident reference, packed --------------------------------------------------------------------------------------------
// 2 bits per base (A C G T), 16 bases per dword; one flag bit per 64-base block that holds anything else (N = 4, or a byte >= 5 that
// the Go code would panic on WHEN AN ALIGNMENT TOUCHES IT -- so it is recorded, not refused), the rank of every flag word, and per
// flagged block the two masks.  4.4e9 bases: 1.1 GB + 13 MB + 16 B per flagged block, instead of 4.4 GB.
size_t ref_words(int64_t len) { return (size_t)((len + 15) / 16) + 4; }
size_t ref_flagwords(int64_t len) { return (size_t)(((len + 63) / 64 + 63) / 64) + 2; }
constexpr int64_t REF_CHUNK = (int64_t)256 << 20; // bases packed per pass (multiple of 4096): staging 256 MB

// pass 1: one thread per 64-base block of the chunk: the four 2-bit words, the block's flag bit
__global__ __launch_bounds__(256) void pack_ref_kernel(const uint8_t *__restrict__ bytes, int64_t base0, int64_t n, unsigned *__restrict__ w2,
                                                       unsigned long long *__restrict__ flag) {
    const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // block index inside the chunk
    if (blk * 64 >= n) return;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    bool exc = false;
    for (int k = 0; k < 64; k++) {
        const int64_t x = blk * 64 + k;
        const int b = x < n ? bytes[x] : 0;
        if (b >= 4) exc = true;
        w[k >> 4] |= (unsigned)(b & 3) << (2 * (k & 15));
    }
    const int64_t gblk = base0 / 64 + blk;
    unsigned *dst = w2 + gblk * 4;
    dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3];
    if (exc) atomicOr(&flag[gblk >> 6], 1ull << (gblk & 63));
}
// pass 2 (ranks known): the masks of the flagged blocks of the chunk
__global__ __launch_bounds__(256) void pack_exc_kernel(const uint8_t *__restrict__ bytes, int64_t base0, int64_t n, const unsigned long long *__restrict__ flag,
                                                       const unsigned *__restrict__ rank, unsigned long long *__restrict__ exc) {
    const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk * 64 >= n) return;
    const int64_t gblk = base0 / 64 + blk;
    const unsigned long long f = flag[gblk >> 6];
    if (!((f >> (gblk & 63)) & 1ull)) return;
    unsigned long long nm = 0, bm = 0;
    for (int k = 0; k < 64; k++) {
        const int64_t x = blk * 64 + k;
        const int b = x < n ? bytes[x] : 0;
        if (b == 4) nm |= 1ull << k;
        if (b >= 5) bm |= 1ull << k;
    }
    const int64_t e = (int64_t)rank[gblk >> 6] + __popcll(f & ((1ull << (gblk & 63)) - 1ull));
    exc[2 * e] = nm; exc[2 * e + 1] = bm;
}
int64_t g_ref_epoch = 0;
int ensure_reference(int nc);
// ref != nullptr: host bytes; else the synthetic reference of SURVEY 8d (generated chunk by chunk on the device)
int set_reference_packed(const uint8_t *ref, int64_t len, uint64_t seed) {
    g_ref_epoch++;
    Ctx &c0 = ctx_at(0);
    {
        CtxScope sc(c0);
        int rc = ensure_init();
        if (rc) return rc;
        HIPCHK(hipSetDevice(c0.device));
        hipStream_t st = c0.own_stream;
        c0.ref_len = -1;
        if (len == 0) { // release: every context gives the reference (and the unpacked-window scratch) back; nothing is re-allocated
            int n_all0; { std::lock_guard<std::mutex> lk(g_ctxs_mu); n_all0 = (int)g_ctxs.size(); }
            for (int d = 0; d < n_all0; d++) {
                Ctx &c = ctx_at(d);
                if (!c.inited) continue;
                std::unique_lock<std::mutex> lkd;
                if (d > 0) lkd = std::unique_lock<std::mutex>(c.mu); // (context 0 is held by this call's scope)
                HIPCHK(hipSetDevice(c.device));
                HIPCHK(hipDeviceSynchronize());
                c.ref.release(); c.ref_flag.release(); c.ref_rank.release(); c.ref_exc.release(); c.unpk_b.release(); c.unpk_off.release();
                c.ref_len = -1;
            }
            HIPCHK(hipSetDevice(c0.device));
            g_bcast_ms = 0;
            return GNX_OK;
        }
        const size_t nw = ref_words(len), nf = ref_flagwords(len);
        if ((rc = c0.ref.ensure(nw * 4))) return rc;
        if ((rc = c0.ref_flag.ensure(nf * 8))) return rc;
        if ((rc = c0.ref_rank.ensure(nf * 4))) return rc;
        HIPCHK(hipMemsetAsync(c0.ref.p, 0, nw * 4, st));
        HIPCHK(hipMemsetAsync(c0.ref_flag.p, 0, nf * 8, st));
        DevBuf stage;
        const int64_t chunk = std::min<int64_t>(REF_CHUNK, std::max<int64_t>(len, 1));
        if ((rc = stage.ensure((size_t)chunk + 64))) return rc;
        std::vector<unsigned long long> hflag(nf, 0ull);
        std::vector<unsigned> hrank(nf, 0u);
        int64_t nexc = 0;
        // exception masks: grown as needed (a genome's N runs are a few per cent of its blocks at most)
        auto grow_exc = [&](int64_t want, int64_t keep) -> int {
            if ((size_t)want * 16 <= c0.ref_exc.cap) return GNX_OK;
            DevBuf nb;
            int r = nb.ensure((size_t)std::max<int64_t>(want * 2, 1024) * 16);
            if (r) return r;
            if (keep > 0) { HIPCHK(hipMemcpyAsync(nb.p, c0.ref_exc.p, (size_t)keep * 16, hipMemcpyDeviceToDevice, st)); HIPCHK(hipStreamSynchronize(st)); }
            c0.ref_exc.release(); c0.ref_exc = nb;
            return GNX_OK;
        };
        if ((rc = grow_exc(1, 0))) { stage.release(); return rc; }
        auto fail = [&](int code) { stage.release(); return code; };
        for (int64_t b0 = 0; b0 < len; b0 += chunk) {
            const int64_t n = std::min(chunk, len - b0);
            if (ref) { if (hipMemcpyAsync(stage.p, ref + b0, (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) { set_err("upload of the reference failed%s", ""); return fail(GNX_EDEVICE); } }
            else hipLaunchKernelGGL(synth_ref_kernel, dim3((unsigned)(((n + 31) / 32 + 255) / 256)), dim3(256), 0, st, (uint8_t *)stage.p, b0, n, seed);
            const int64_t nblk = (n + 63) / 64;
            hipLaunchKernelGGL(pack_ref_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st, (const uint8_t *)stage.p, b0, n, (unsigned *)c0.ref.p, (unsigned long long *)c0.ref_flag.p);
            // ranks of this chunk's flag words (the chunk starts on a flag-word boundary: REF_CHUNK is a multiple of 4096)
            const size_t f0 = (size_t)(b0 / 4096), f1 = (size_t)((b0 + n + 4095) / 4096);
            if (hipMemcpyAsync(hflag.data() + f0, (const unsigned long long *)c0.ref_flag.p + f0, (f1 - f0) * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { set_err("packing the reference failed%s", ""); return fail(GNX_EDEVICE); }
            const int64_t before = nexc;
            for (size_t f = f0; f < f1; f++) { hrank[f] = (unsigned)nexc; nexc += __builtin_popcountll(hflag[f]); }
            if (nexc > 0xfffffff0LL) { set_err("too many non-ACGT blocks in the reference%s", ""); return fail(GNX_ENOMEM); }
            if (nexc > before) {
                if ((rc = grow_exc(nexc, before))) return fail(rc);
                if (hipMemcpyAsync((unsigned *)c0.ref_rank.p + f0, hrank.data() + f0, (f1 - f0) * 4, hipMemcpyHostToDevice, st) != hipSuccess) { set_err("packing the reference failed%s", ""); return fail(GNX_EDEVICE); }
                hipLaunchKernelGGL(pack_exc_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, st, (const uint8_t *)stage.p, b0, n, (const unsigned long long *)c0.ref_flag.p,
                                   (const unsigned *)c0.ref_rank.p, (unsigned long long *)c0.ref_exc.p);
            }
            if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { set_err("packing the reference failed%s", ""); return fail(GNX_EDEVICE); }
        }
        // the ranks of ALL flag words (also behind the last chunk: BetaSrc::init looks one block past a window's end)
        for (size_t f = (size_t)((len + 4095) / 4096); f < nf; f++) hrank[f] = (unsigned)nexc;
        if (hipMemcpy(c0.ref_rank.p, hrank.data(), nf * 4, hipMemcpyHostToDevice) != hipSuccess) { set_err("packing the reference failed%s", ""); return fail(GNX_EDEVICE); }
        stage.release();
        c0.ref_len = len; c0.ref_nexc = nexc; c0.ref_epoch = g_ref_epoch;
    }
    // contexts beyond the ones in use no longer hold the current reference; the others get it now
    int n_all; { std::lock_guard<std::mutex> lk(g_ctxs_mu); n_all = (int)g_ctxs.size(); }
    for (int d = 1; d < n_all; d++) { Ctx &c = ctx_at(d); CtxScope sc(c); c.ref_len = -1; }
    g_bcast_ms = 0;
    return ensure_reference(g_nctx);
}

// ---- one context's share of a batch ---------------------------------------------------------------------------------------------
struct HostJob {
    Ctx *c = nullptr;
    const gnx_params *prm = nullptr;
    int64_t p0 = 0, p1 = 0; // pairs [p0, p1) of the call
    const uint8_t *a_buf = nullptr; const int64_t *a_start = nullptr, *a_len = nullptr; // host windows
    const uint8_t *b_buf = nullptr; const int64_t *b_start = nullptr, *b_len = nullptr; // host windows, or windows into b_dev
    const uint8_t *b_dev = nullptr; // != nullptr: the whole beta buffer / the resident reference is on this context's device
    bool packed = false;            // b_dev is the PACKED resident reference (beta windows = base positions in it)
    int64_t total_ops = 0;
    int rc = GNX_OK;
    char err[512] = "";
    gnx_timing timing = {};
};

int grow_ops(Ctx &c, int64_t keep_elems, int64_t want_elems, hipStream_t st) {
    if ((size_t)want_elems * sizeof(gnx_cigar) <= c.res_ops.cap) return GNX_OK;
    DevBuf nb;
    int rc = nb.ensure((size_t)want_elems * sizeof(gnx_cigar));
    if (rc) return rc;
    if (keep_elems > 0) {
        HIPCHK(hipMemcpyAsync(nb.p, c.res_ops.p, (size_t)keep_elems * sizeof(gnx_cigar), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    c.res_ops.release();
    c.res_ops = nb;
    return GNX_OK;
}

// runs on the thread that owns the context (t_ctx == job.c, device current)
int run_host_job(HostJob &j) {
    Ctx &c = *j.c;
    const int64_t n = j.p1 - j.p0;
    int rc;
    if ((rc = c.res_score.ensure((size_t)std::max<int64_t>(n, 1) * 8))) return rc;
    if ((rc = c.res_off.ensure((size_t)(n + 1) * 8))) return rc;
    if (n == 0) { HIPCHK(hipMemsetAsync(c.res_off.p, 0, 8, c.own_stream)); HIPCHK(hipStreamSynchronize(c.own_stream)); return GNX_OK; }
    int64_t sub = 131072;
    if (const char *e = getenv("GNX_HOST_SUB")) sub = (std::max<int64_t>(atoll(e), 8) + 7) & ~(int64_t)7;
    const int64_t K0 = (n + sub - 1) / sub;
    const int64_t size = (((n + K0 - 1) / K0) + 7) & ~(int64_t)7; // equal sub-batches (the fast path re-uses its plans), whole waves
    const int64_t K = (n + size - 1) / size;                      // rounding `size` up can save a sub-batch: never index past n
    const int64_t *as = j.a_start + j.p0, *al = j.a_len + j.p0, *bs = j.b_start + j.p0, *bl = j.b_len + j.p0;

    // stage(k): inputs of sub-batch k -> device buffers of slot k & 1 (H2D on s_in, event ev_in[slot]).  k == 0 copies straight from
    // the caller's memory (nothing to overlap with); later ones go through pinned memory so that the DMA runs under the kernels.
    std::vector<int64_t> tmp_as, tmp_bs;
    int stage_rc = GNX_OK;
    char stage_err[512] = "";
    auto stage = [&](int64_t k, bool pinned) -> int {
        const int slot = (int)(k & 1);
        const int64_t b = k * size, e = std::min(n, b + size), cnt = e - b;
        int64_t alo = INT64_MAX, ahi = 0, blo = INT64_MAX, bhi = 0;
        for (int64_t q = b; q < e; q++) {
            alo = std::min(alo, as[q]); ahi = std::max(ahi, as[q] + al[q]);
            if (!j.b_dev) { blo = std::min(blo, bs[q]); bhi = std::max(bhi, bs[q] + bl[q]); }
        }
        if (ahi < alo) { alo = 0; ahi = 0; }
        if (bhi < blo) { blo = 0; bhi = 0; }
        int r;
        if ((r = c.pin_a[slot].ensure((size_t)(ahi - alo) + 16))) return r;
        if ((r = c.pin_as[slot].ensure((size_t)cnt * 8))) return r;
        if ((r = c.pin_bs[slot].ensure((size_t)cnt * 8))) return r;
        if (!j.b_dev && (r = c.pin_b[slot].ensure((size_t)(bhi - blo) + 16))) return r;
        int64_t *has, *hbs;
        const uint8_t *src_a = j.a_buf + alo, *src_b = j.b_dev ? nullptr : j.b_buf + blo;
        if (pinned) {
            if ((r = c.st_a[slot].ensure((size_t)(ahi - alo) + 16))) return r;
            if ((r = c.st_as[slot].ensure((size_t)cnt * 8))) return r;
            if ((r = c.st_bs[slot].ensure((size_t)cnt * 8))) return r;
            if (!j.b_dev && (r = c.st_b[slot].ensure((size_t)(bhi - blo) + 16))) return r;
            memcpy(c.st_a[slot].p, src_a, (size_t)(ahi - alo)); src_a = (const uint8_t *)c.st_a[slot].p;
            if (!j.b_dev) { memcpy(c.st_b[slot].p, src_b, (size_t)(bhi - blo)); src_b = (const uint8_t *)c.st_b[slot].p; }
            has = (int64_t *)c.st_as[slot].p; hbs = (int64_t *)c.st_bs[slot].p;
        } else {
            tmp_as.resize((size_t)cnt); tmp_bs.resize((size_t)cnt);
            has = tmp_as.data(); hbs = tmp_bs.data();
        }
        for (int64_t q = 0; q < cnt; q++) { has[q] = as[b + q] - alo; hbs[q] = j.b_dev ? bs[b + q] : bs[b + q] - blo; }
        if (ahi > alo) HIPCHK(hipMemcpyAsync(c.pin_a[slot].p, src_a, (size_t)(ahi - alo), hipMemcpyHostToDevice, c.s_in));
        if (!j.b_dev && bhi > blo) HIPCHK(hipMemcpyAsync(c.pin_b[slot].p, src_b, (size_t)(bhi - blo), hipMemcpyHostToDevice, c.s_in));
        HIPCHK(hipMemcpyAsync(c.pin_as[slot].p, has, (size_t)cnt * 8, hipMemcpyHostToDevice, c.s_in));
        HIPCHK(hipMemcpyAsync(c.pin_bs[slot].p, hbs, (size_t)cnt * 8, hipMemcpyHostToDevice, c.s_in));
        if (!pinned) HIPCHK(hipStreamSynchronize(c.s_in)); // tmp_* / the caller's pageable memory: do not rely on the staging
        HIPCHK(hipEventRecord(c.ev_in[slot], c.s_in));
        return GNX_OK;
    };

    const auto t_stage = std::chrono::steady_clock::now();
    if ((rc = stage(0, false))) return rc;
    const double stage0_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_stage).count();
    int64_t done = 0, total = 0;
    // CIGAR capacity: a guess that fits every workload of the path's callers, grown (with a retry of the sub-batch) when it does not
    int64_t worst = 0;
    for (int64_t q = 0; q < n; q++) worst += al[q] + bl[q] + 1;
    if ((rc = grow_ops(c, 0, std::max<int64_t>(std::min<int64_t>(worst, std::max<int64_t>((int64_t)1 << 20, 64 * std::min(n, size))), 1), c.own_stream))) return rc;
    gnx_timing tsum = {};
    for (int64_t k = 0; k < K; k++) {
        const int slot = (int)(k & 1);
        const int64_t b = k * size, e = std::min(n, b + size), cnt = e - b;
        std::thread stager;
        if (k + 1 < K) {
            stager = std::thread([&, k]() {
                t_ctx = &c;
                if (hipSetDevice(c.device) != hipSuccess) { stage_rc = GNX_EDEVICE; return; }
                g_err[0] = 0;
                stage_rc = stage(k + 1, true);
                if (stage_rc) memcpy(stage_err, g_err, sizeof(stage_err));
            });
        }
        rc = GNX_OK;
        if (hipStreamWaitEvent(c.own_stream, c.ev_in[slot], 0) != hipSuccess) { set_err("hipStreamWaitEvent failed%s", ""); rc = GNX_EDEVICE; }
        int64_t tot = 0;
        const uint8_t *db = j.b_dev ? j.b_dev : (const uint8_t *)c.pin_b[slot].p;
        const int64_t *dbs = (const int64_t *)c.pin_bs[slot].p;
        bool packed = j.packed;
        if (packed && rc == GNX_OK && (j.prm->mode == GNX_AFFINE_GAP_LOCAL || getenv("GNX_REF_UNPACK"))) {
            // the windows as bytes (AffineGapLocal's transposed fast path reads the long sequence as the kernels' alpha; GNX_REF_UNPACK: A/B)
            std::vector<int64_t> uoff((size_t)cnt + 1, 0);
            for (int64_t q = 0; q < cnt; q++) { uoff[(size_t)q + 1] = uoff[(size_t)q] + bl[b + q]; }
            if ((rc = c.in_b.ensure((size_t)uoff[(size_t)cnt] + 64)) == GNX_OK && (rc = c.in_bl.ensure((size_t)(cnt + 1) * 8)) == GNX_OK) {
                KParams ukp;
                memset(&ukp, 0, sizeof(ukp));
                ukp.b2 = (const unsigned *)c.ref.p; ukp.bflag = (const unsigned long long *)c.ref_flag.p; ukp.brank = (const unsigned *)c.ref_rank.p; ukp.bexc = (const unsigned long long *)c.ref_exc.p;
                if (hipMemcpyAsync(c.in_bl.p, uoff.data(), (size_t)(cnt + 1) * 8, hipMemcpyHostToDevice, c.own_stream) != hipSuccess || hipStreamSynchronize(c.own_stream) != hipSuccess) { set_err("upload of the window table failed%s", ""); rc = GNX_EDEVICE; }
                else {
                    hipLaunchKernelGGL(unpack_windows_kernel, dim3((unsigned)cnt), dim3(256), 0, c.own_stream, ukp, dbs, (const int64_t *)c.in_bl.p, (int)cnt, (uint8_t *)c.in_b.p);
                    if (hipGetLastError() != hipSuccess) { set_err("unpack_windows_kernel failed to launch%s", ""); rc = GNX_EDEVICE; }
                    db = (const uint8_t *)c.in_b.p; dbs = (const int64_t *)c.in_bl.p; packed = false;
                }
            }
        }
        for (int attempt = 0; rc == GNX_OK; attempt++) {
            const int64_t cap = (int64_t)(c.res_ops.cap / sizeof(gnx_cigar)) - total;
            c.beta_packed = packed;
            rc = run_device(j.prm, cnt, (const uint8_t *)c.pin_a[slot].p, (const int64_t *)c.pin_as[slot].p,
                            db, dbs, al + b, bl + b,
                            (int64_t *)c.res_score.p + done, (gnx_cigar *)c.res_ops.p + total, cap, (int64_t *)c.res_off.p + done, &tot, c.own_stream);
            c.beta_packed = false;
            if (rc != GNX_ECAPACITY || attempt >= 8) break;
            // the reported total is exact (every sub-batch of the device flow is counted); leave room for the sub-batches to come
            const int64_t left = (n - done + cnt - 1) / cnt;
            rc = grow_ops(c, total, total + tot * std::min<int64_t>(left, 4) + 1024, c.own_stream);
        }
        if (rc == GNX_OK) {
            tsum.fill_ms += c.timing.fill_ms; tsum.traceback_ms += c.timing.traceback_ms; tsum.total_ms += c.timing.total_ms; tsum.cells += c.timing.cells;
            tsum.n_launches += c.timing.n_launches; tsum.trace_bytes += c.timing.trace_bytes; tsum.dominant_ms += c.timing.dominant_ms;
            tsum.dominant_launches += c.timing.dominant_launches; tsum.fast_path = c.timing.fast_path;
            if (total > 0) { // offsets of a sub-batch start at 0
                hipLaunchKernelGGL(add_offset_kernel, dim3((unsigned)((cnt + 1 + 255) / 256)), dim3(256), 0, c.own_stream, (int64_t *)c.res_off.p + done, cnt + 1, total);
                if (hipGetLastError() != hipSuccess) { set_err("add_offset_kernel failed to launch%s", ""); rc = GNX_EDEVICE; }
            }
            total += tot; done += cnt;
        }
        if (stager.joinable()) stager.join();
        if (rc) return rc;
        if (stage_rc) { memcpy(g_err, stage_err, sizeof(stage_err)); publish_err(); return stage_rc; }
    }
    HIPCHK(hipStreamSynchronize(c.own_stream));
    j.total_ops = total;
    tsum.stage0_ms = stage0_ms;
    j.timing = tsum;
    c.timing = tsum;
    return GNX_OK;
}

// contiguous blocks of (nearly) equal DP cells, 8-aligned so that waves stay whole
std::vector<int64_t> partition_by_cells(const int64_t *a_len, const int64_t *b_len, int64_t n, int parts) {
    std::vector<int64_t> bounds((size_t)parts + 1, n);
    bounds[0] = 0;
    if (parts <= 1 || n == 0) return bounds;
    long double total = 0;
    for (int64_t q = 0; q < n; q++) total += (long double)a_len[q] * (long double)b_len[q];
    long double acc = 0;
    int next = 1;
    for (int64_t q = 0; q < n && next < parts; q++) {
        acc += (long double)a_len[q] * (long double)b_len[q];
        while (next < parts && acc >= total * next / parts) { bounds[(size_t)next] = std::min<int64_t>(n, (q + 1 + 7) & ~(int64_t)7); next++; }
    }
    for (int r = 1; r <= parts; r++) bounds[(size_t)r] = std::max(bounds[(size_t)r], bounds[(size_t)r - 1]);
    bounds[(size_t)parts] = n;
    return bounds;
}

// broadcast `bytes` from context 0's buffer src0 to dst[d] of every other context (RCCL over xGMI; plain copies when the contexts
// share a device or RCCL is off)
// GNX_RCCL_INJECT_FAIL=1: the next RCCL exchange reports a failure before it starts; =2: INSIDE its group, after the first operation
// has been enqueued (tests of the fall-back to peer copies; `where` = 1 before ncclGroupStart, 2 inside the group)
bool rccl_injected_failure(int where) {
    const char *e = getenv("GNX_RCCL_INJECT_FAIL");
    if (!e || atoi(e) != where) return false;
    set_err("RCCL error: injected failure (GNX_RCCL_INJECT_FAIL=%s)", e);
    return true;
}
// A grouped exchange: `body` enqueues the operations between ncclGroupStart and ncclGroupEnd.  Whatever fails inside the group, the
// group is CLOSED again before the error is reported (an open group would swallow the next ncclCommInitAll of this thread) and the
// current device is context 0's; the caller then gives RCCL up and repeats the exchange with peer copies.
template <class Body>
int rccl_grouped(Body body) {
    if (rccl_injected_failure(1)) return GNX_EDEVICE;
    RCCLCHK(g_rccl.GroupStart());
    const int rc = body();
    const ncclResult_t re = g_rccl.GroupEnd(); // also on the error path; its own result only matters when the body was fine
    (void)hipSetDevice(ctx_at(0).device);
    if (rc) return rc;
    if (re != ncclSuccess) { set_err("RCCL error: %s", g_rccl.GetErrorString(re)); return GNX_EDEVICE; }
    return GNX_OK;
}
int broadcast_rccl(const void *src0, std::vector<void *> &dst, size_t bytes) {
    const int nc = (int)dst.size();
    int rc = rccl_grouped([&]() -> int {
        for (int d = 0; d < nc; d++) {
            Ctx &c = ctx_at(d);
            HIPCHK(hipSetDevice(c.device));
            RCCLCHK(g_rccl.Broadcast(src0, d == 0 ? const_cast<void *>(src0) : dst[(size_t)d], bytes, ncclUint8, 0, g_rccl.comms[(size_t)d], c.own_stream));
            if (d == 0 && rccl_injected_failure(2)) return GNX_EDEVICE;
        }
        return GNX_OK;
    });
    if (rc) return rc;
    for (int d = 0; d < nc; d++) { Ctx &c = ctx_at(d); HIPCHK(hipSetDevice(c.device)); HIPCHK(hipStreamSynchronize(c.own_stream)); }
    HIPCHK(hipSetDevice(ctx_at(0).device));
    return GNX_OK;
}
int broadcast_from_ctx0(const void *src0, std::vector<void *> &dst, size_t bytes) {
    const int nc = (int)dst.size();
    if (bytes == 0 || nc == 0) return GNX_OK;
    const auto t0 = std::chrono::steady_clock::now();
    // RCCL needs every rank of the communicator in the collective: a call that uses fewer contexts than gnx_init_devices created
    // (a batch too small to cut) copies instead
    bool via_rccl = rccl_active() && (int)g_rccl.comms.size() == nc; // (also with one rank: a no-op that checks the plumbing)
    const bool tried = via_rccl;
    if (via_rccl && broadcast_rccl(src0, dst, bytes) != GNX_OK) { rccl_give_up(); via_rccl = false; }
    if (!via_rccl) for (int d = 1; d < nc; d++) if (dst[(size_t)d] != src0) HIPCHK(hipMemcpyPeer(dst[(size_t)d], ctx_at(d).device, src0, ctx_at(0).device, bytes));
    HIPCHK(hipSetDevice(ctx_at(0).device));
    if (nc > 1 || tried) g_transport = via_rccl ? 1 : (g_rccl_broken ? 3 : 2);
    g_bcast_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return GNX_OK;
}
// the resident reference on contexts [0, nc): a context that lacks the current one (created after gnx_set_reference -- ADVICE r2 --
// or simply not context 0, where the packing happens) gets its copy from context 0: four arrays, RCCL over xGMI or peer copies
int ensure_reference(int nc) {
    Ctx &c0 = ctx_at(0);
    if (c0.ref_len < 0) return GNX_OK;
    const size_t sz[4] = {ref_words(c0.ref_len) * 4, ref_flagwords(c0.ref_len) * 8, ref_flagwords(c0.ref_len) * 4, (size_t)std::max<int64_t>(c0.ref_nexc, 1) * 16};
    bool missing = false;
    std::vector<void *> dst[4];
    for (int k = 0; k < 4; k++) dst[k].assign((size_t)nc, nullptr);
    for (int d = 0; d < nc; d++) {
        Ctx &c = ctx_at(d);
        CtxScope sc(c);
        int rc = ensure_init();
        if (rc) return rc;
        DevBuf *bufs[4] = {&c.ref, &c.ref_flag, &c.ref_rank, &c.ref_exc};
        if (d > 0 && (c.ref_len != c0.ref_len || c.ref_epoch != c0.ref_epoch || c.ref.p == nullptr)) {
            missing = true;
            for (int k = 0; k < 4; k++) if ((rc = bufs[k]->ensure(sz[k]))) { c.ref_len = -1; return rc; }
            c.ref_len = c0.ref_len; c.ref_nexc = c0.ref_nexc; c.ref_epoch = c0.ref_epoch;
        }
        for (int k = 0; k < 4; k++) dst[k][(size_t)d] = bufs[k]->p;
    }
    if (!missing) return GNX_OK;
    for (int k = 0; k < 4; k++) { int rc = broadcast_from_ctx0(dst[k][0], dst[k], sz[k]); if (rc) return rc; }
    return GNX_OK;
}

// The sharded host flow.  b_buf == nullptr: beta windows index the resident reference (gnx_set_reference).
int run_host_sharded(const gnx_params *prm, int64_t n_pairs,
                     const uint8_t *a_buf, int64_t a_len_total, const int64_t *a_start, const int64_t *a_lens,
                     const uint8_t *b_buf, int64_t b_len_total, const int64_t *b_start, const int64_t *b_lens,
                     int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    const auto t_entry = std::chrono::steady_clock::now();
    if (!prm || n_pairs < 0 || !out_score || !out_ops || !out_ops_off || a_len_total < 0 || b_len_total < 0) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    if (n_pairs > 0 && (!a_start || !a_lens || !b_start || !b_lens)) { set_err("null window table%s", ""); return GNX_EINVAL; }
    const bool resident = (b_buf == nullptr);
    Ctx &c0 = ctx_at(0);
    int rc;
    { CtxScope sc(c0); if ((rc = ensure_init())) return rc; }
    g_transport = 0; g_bcast_ms = 0;
    if (resident) {
        if (c0.ref_len < 0) { set_err("no resident reference: call gnx_set_reference first%s", ""); return GNX_EINVAL; }
        b_len_total = c0.ref_len;
    }
    long double sum_b = 0;
    for (int64_t p = 0; p < n_pairs; p++) {
        if (a_start[p] < 0 || a_lens[p] < 0 || a_start[p] + a_lens[p] > a_len_total || b_start[p] < 0 || b_lens[p] < 0 || b_start[p] + b_lens[p] > b_len_total) {
            set_err("window out of bounds at pair %s%lld", "", (long long)p); return GNX_EINVAL;
        }
        sum_b += (long double)b_lens[p];
    }
    const int nc = std::max(1, std::min<int>(g_nctx, (int)std::max<int64_t>(n_pairs / 8, 1)));
    const std::vector<int64_t> bounds = partition_by_cells(a_lens, b_lens, n_pairs, nc);
    // a beta buffer that the pairs share (one chunk for every read; windows that overlap) goes to the devices whole -- once to
    // device 0, from there over RCCL; disjoint windows (one per pair) travel with their sub-batches instead
    const bool whole_b = !resident && b_len_total > 0 && (sum_b > 1.5L * (long double)b_len_total || b_len_total <= ((int64_t)16 << 20));
    std::vector<void *> bdev((size_t)nc, nullptr);
    if (resident) {
        if ((rc = ensure_reference(nc))) return rc;
        for (int d = 0; d < nc; d++) {
            bdev[(size_t)d] = ctx_at(d).ref.p;
            if (!bdev[(size_t)d] && b_len_total > 0) { set_err("context %s%lld has no resident reference", "", (long long)d); return GNX_EINVAL; }
        }
    } else if (whole_b) {
        for (int d = 0; d < nc; d++) {
            Ctx &c = ctx_at(d);
            CtxScope sc(c);
            if ((rc = ensure_init())) return rc;
            if ((rc = c.in_b.ensure((size_t)b_len_total + 16))) return rc;
            bdev[(size_t)d] = c.in_b.p;
        }
        { CtxScope sc(c0); HIPCHK(hipSetDevice(c0.device)); HIPCHK(hipMemcpy(c0.in_b.p, b_buf, (size_t)b_len_total, hipMemcpyHostToDevice)); }
        if ((rc = broadcast_from_ctx0(c0.in_b.p, bdev, (size_t)b_len_total))) return rc;
    }
    std::vector<HostJob> jobs((size_t)nc);
    for (int d = 0; d < nc; d++) {
        HostJob &j = jobs[(size_t)d];
        j.c = &ctx_at(d); j.prm = prm; j.p0 = bounds[(size_t)d]; j.p1 = bounds[(size_t)d + 1];
        j.a_buf = a_buf; j.a_start = a_start; j.a_len = a_lens; j.b_buf = b_buf; j.b_start = b_start; j.b_len = b_lens;
        j.b_dev = (const uint8_t *)bdev[(size_t)d];
        j.packed = resident;
    }
    auto work = [](HostJob *j) {
        CtxScope sc(*j->c);
        g_err[0] = 0;
        j->rc = ensure_init();
        if (!j->rc) j->rc = run_host_job(*j);
        if (j->rc) memcpy(j->err, g_err, sizeof(j->err));
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nc; d++) th.emplace_back(work, &jobs[(size_t)d]);
    work(&jobs[0]);
    for (auto &t : th) t.join();
    for (int d = 0; d < nc; d++) if (jobs[(size_t)d].rc) { memcpy(g_err, jobs[(size_t)d].err, sizeof(g_err)); publish_err(); return jobs[(size_t)d].rc; }
    int64_t total = 0;
    for (int d = 0; d < nc; d++) total += jobs[(size_t)d].total_ops;
    // ---- gather on device 0 (input order == context order), then one D2H into pinned result arrays ----
    const auto t_fetch = std::chrono::steady_clock::now();
    gnx_cigar *ops = (gnx_cigar *)g_pool.get((size_t)std::max<int64_t>(total, 1) * sizeof(gnx_cigar));
    int64_t *off = (int64_t *)g_pool.get((size_t)(n_pairs + 1) * 8);
    if (!ops || !off) { if (ops) g_pool.put(ops); if (off) g_pool.put(off); set_err("pinned host allocation failed%s", ""); return GNX_ENOMEM; }
    auto fail = [&](int code) { g_pool.put(ops); g_pool.put(off); return code; };
    CtxScope sc(c0);
    if (hipSetDevice(c0.device) != hipSuccess) { set_err("hipSetDevice failed%s", ""); return fail(GNX_EDEVICE); }
    double gather_ms = 0;
    const int64_t *d_score = (const int64_t *)c0.res_score.p, *d_off = (const int64_t *)c0.res_off.p;
    const gnx_cigar *d_ops = (const gnx_cigar *)c0.res_ops.p;
    if (nc > 1) {
        if ((rc = c0.gat_score.ensure((size_t)std::max<int64_t>(n_pairs, 1) * 8))) return fail(rc);
        if ((rc = c0.gat_off.ensure((size_t)(n_pairs + nc) * 8))) return fail(rc);
        if ((rc = c0.gat_ops.ensure((size_t)std::max<int64_t>(total, 1) * sizeof(gnx_cigar)))) return fail(rc);
        // context 0's own share never goes through RCCL (no send-to-self): a device-to-device copy on its stream
        auto gather_local = [&]() -> int {
            HostJob &j = jobs[0];
            const int64_t nd = j.p1 - j.p0;
            HIPCHK(hipSetDevice(c0.device));
            if (nd > 0) {
                HIPCHK(hipMemcpyAsync((int64_t *)c0.gat_score.p + j.p0, c0.res_score.p, (size_t)nd * 8, hipMemcpyDeviceToDevice, c0.own_stream));
                HIPCHK(hipMemcpyAsync((int64_t *)c0.gat_off.p + j.p0, c0.res_off.p, (size_t)(nd + 1) * 8, hipMemcpyDeviceToDevice, c0.own_stream));
            }
            if (j.total_ops > 0) HIPCHK(hipMemcpyAsync(c0.gat_ops.p, c0.res_ops.p, (size_t)j.total_ops * sizeof(gnx_cigar), hipMemcpyDeviceToDevice, c0.own_stream));
            return GNX_OK;
        };
        auto gather_rccl = [&]() -> int {
            int64_t obase = jobs[0].total_ops;
            const int grc = rccl_grouped([&]() -> int {
                for (int d = 1; d < nc; d++) {
                    HostJob &j = jobs[(size_t)d];
                    Ctx &c = *j.c;
                    const int64_t nd = j.p1 - j.p0;
                    HIPCHK(hipSetDevice(c.device));
                    if (nd > 0) {
                        RCCLCHK(g_rccl.Send(c.res_score.p, (size_t)nd, ncclInt64, 0, g_rccl.comms[(size_t)d], c.own_stream));
                        RCCLCHK(g_rccl.Send(c.res_off.p, (size_t)nd + 1, ncclInt64, 0, g_rccl.comms[(size_t)d], c.own_stream));
                    }
                    if (d == 1 && rccl_injected_failure(2)) return GNX_EDEVICE; // (a send is enqueued, its receive is not)
                    if (j.total_ops > 0) RCCLCHK(g_rccl.Send(c.res_ops.p, (size_t)j.total_ops * 2, ncclInt64, 0, g_rccl.comms[(size_t)d], c.own_stream));
                    HIPCHK(hipSetDevice(c0.device));
                    if (nd > 0) {
                        RCCLCHK(g_rccl.Recv((int64_t *)c0.gat_score.p + j.p0, (size_t)nd, ncclInt64, d, g_rccl.comms[0], c0.own_stream));
                        RCCLCHK(g_rccl.Recv((int64_t *)c0.gat_off.p + j.p0 + d, (size_t)nd + 1, ncclInt64, d, g_rccl.comms[0], c0.own_stream));
                    }
                    if (j.total_ops > 0) RCCLCHK(g_rccl.Recv((gnx_cigar *)c0.gat_ops.p + obase, (size_t)j.total_ops * 2, ncclInt64, d, g_rccl.comms[0], c0.own_stream));
                    obase += j.total_ops;
                }
                return GNX_OK;
            });
            if (grc) return grc;
            for (int d = 0; d < nc; d++) { Ctx &c = *jobs[(size_t)d].c; HIPCHK(hipSetDevice(c.device)); HIPCHK(hipStreamSynchronize(c.own_stream)); }
            HIPCHK(hipSetDevice(c0.device));
            return GNX_OK;
        };
        auto gather_copies = [&]() -> int {
            int64_t obase = jobs[0].total_ops;
            for (int d = 1; d < nc; d++) {
                HostJob &j = jobs[(size_t)d];
                Ctx &c = *j.c;
                const int64_t nd = j.p1 - j.p0;
                if (nd > 0) {
                    HIPCHK(hipMemcpyPeer((int64_t *)c0.gat_score.p + j.p0, c0.device, c.res_score.p, c.device, (size_t)nd * 8));
                    HIPCHK(hipMemcpyPeer((int64_t *)c0.gat_off.p + j.p0 + d, c0.device, c.res_off.p, c.device, (size_t)(nd + 1) * 8));
                }
                if (j.total_ops > 0) HIPCHK(hipMemcpyPeer((gnx_cigar *)c0.gat_ops.p + obase, c0.device, c.res_ops.p, c.device, (size_t)j.total_ops * sizeof(gnx_cigar)));
                obase += j.total_ops;
            }
            HIPCHK(hipSetDevice(c0.device));
            return GNX_OK;
        };
        auto gather = [&]() -> int {
            int r = gather_local();
            if (r) return r;
            // every rank of the communicator must take part in a grouped exchange: only when the call uses all contexts
            bool via_rccl = rccl_active() && (int)g_rccl.comms.size() == nc;
            if (via_rccl && gather_rccl() != GNX_OK) { rccl_give_up(); via_rccl = false; if ((r = gather_local())) return r; }
            if (!via_rccl && (r = gather_copies())) return r;
            HIPCHK(hipStreamSynchronize(c0.own_stream));
            g_transport = via_rccl ? 1 : (g_rccl_broken ? 3 : 2);
            return GNX_OK;
        };
        const auto t_gather = std::chrono::steady_clock::now();
        if ((rc = gather())) return fail(rc);
        gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_gather).count();
        d_score = (const int64_t *)c0.gat_score.p; d_ops = (const gnx_cigar *)c0.gat_ops.p;
    }
    auto fetch = [&]() -> int {
        hipStream_t st = c0.own_stream;
        if (n_pairs) HIPCHK(hipMemcpyAsync(out_score, d_score, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st));
        if (total) HIPCHK(hipMemcpyAsync(ops, d_ops, (size_t)total * sizeof(gnx_cigar), hipMemcpyDeviceToHost, st));
        if (nc == 1) HIPCHK(hipMemcpyAsync(off, d_off, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, st));
        else {
            // each context's offsets (n_d + 1 of them, starting at 0) sit at [p0 + d ..]: rebase on the host while copying
            std::vector<int64_t> tmp((size_t)(n_pairs + nc));
            HIPCHK(hipMemcpyAsync(tmp.data(), c0.gat_off.p, (size_t)(n_pairs + nc) * 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            int64_t base = 0;
            for (int d = 0; d < nc; d++) {
                const HostJob &j = jobs[(size_t)d];
                for (int64_t q = j.p0; q < j.p1; q++) off[q] = tmp[(size_t)(q + d)] + base;
                base += j.total_ops;
            }
            off[n_pairs] = base;
        }
        HIPCHK(hipStreamSynchronize(st));
        return GNX_OK;
    };
    if ((rc = fetch())) return fail(rc);
    // per-call timing = the slowest context (they run side by side); cells and bytes add up
    gnx_timing t = jobs[0].timing;
    for (int d = 1; d < nc; d++) {
        const gnx_timing &u = jobs[(size_t)d].timing;
        t.fill_ms = std::max(t.fill_ms, u.fill_ms); t.traceback_ms = std::max(t.traceback_ms, u.traceback_ms); t.total_ms = std::max(t.total_ms, u.total_ms);
        t.dominant_ms = std::max(t.dominant_ms, u.dominant_ms); t.cells += u.cells; t.trace_bytes += u.trace_bytes; t.n_launches += u.n_launches; t.dominant_launches += u.dominant_launches;
    }
    t.fetch_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fetch).count();
    t.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count();
    t.transport = g_transport; t.n_contexts = nc; t.gather_ms = gather_ms; t.bcast_ms = g_bcast_ms;
    c0.timing = t;
    if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx host] %lld pairs on %d context(s): call %.3f ms = first upload %.3f + kernels %.3f (device, slowest context) + gather / D2H %.3f (gather %.3f, transport %d) + broadcast %.3f + rest\n",
                                     (long long)n_pairs, nc, t.host_ms, t.stage0_ms, t.total_ms, t.fetch_ms, t.gather_ms, t.transport, t.bcast_ms);
    *out_ops = ops; *out_ops_off = off;
    return GNX_OK;
}

} // namespace
