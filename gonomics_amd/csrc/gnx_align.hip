// gnx_align.hip -- MI355X (gfx950 / CDNA4) implementation of the gonomics `align` pairwise DP hot path.
//
// What is computed (reference semantics, bit-exact):
//   align/affineGap.go:59-344, align/constGap.go:13-311, align/affineGap_highMem.go:57-223,
//   align/constGap_highMem.go:11-67, align/align.go:76-90 (tripleMaxTrace tie order M >= I >= D).
//
// How (MI355X-first, nothing here is a translation of the Go loops; details in DESIGN.md section 4 and in the kernel headers):
//   * Cells are int32 "keys" = 4*score + tag (tag 3/2/1 = came-from M/I/D), so v_max3_i32 returns value AND argmax with the
//     reference's tie order, and the 2-bit direction is the low bits of the winner.  With h = max3(M,I,D) and gapOpen <= 0 the
//     Gotoh recurrences collapse to  rt = max(h+oe, I+e), dn = max(h+oe, D+e)  with identical values and tags ("h-form").
//   * GENERAL path (any shape): fill kernels with one 16-lane DPP row per pair (4 pairs per wave64), each lane owns R = 10
//     consecutive DP rows (alpha), the wave sweeps the columns (beta) as an anti-diagonal wavefront; cross-lane traffic is
//     three `row_shr:1` DPP moves per step.  Direction bits are shifted into accumulators with v_alignbit and flushed every 16
//     steps as coalesced 16-byte stores (6 bits/cell affine, 2 bits/cell const).  Longer alpha = 160-row strips with a row buffer
//     in HBM; small launches run the strips as pipelined workgroups.  A separate traceback kernel walks the packed matrix
//     (one lane per pair, or one wave per pair for long pairs), emulating the reference's checkerboard quirks, two passes
//     (count, exclusive scan, write) so CIGARs come out densely in input order.
//   * FAST path (a read of up to 20 480 bases x a window of >= 768, affine; the headline workload; AffineGapLocal runs it
//     transposed): fp_sweep_kernel computes scores only -- 8 lanes x 19/20 rows per pair, rebased keys (5 VALU instructions per
//     cell), column checkpoints every 128 columns, the I-planes of the last four rows; longer reads as row blocks of 160 rows in
//     ONE launch whose levels follow each other through a row buffer (fp_sweep_levels_kernel) -- and fp_walk_kernel finishes a
//     read whose traceback is a plain diagonal on the spot, or re-fills just the <= 319 columns x <= 160 rows around the path
//     with the general kernel (WIN) to read its direction bits, block by block.
//   * No MFMA: this is an integer max-plus recurrence.  No CPU fallback: every entry point needs the GPU.
//
// Source layout (one translation unit):
//   gnx_common.hip.h   constants, PairPlan / KParams, helpers
//   fill_affine.hip.h  fill_affine_kernel        fp_sweep.hip.h   fp_sweep_kernel (dominant kernel of the headline workload)
//   fill_const.hip.h   fill_const_kernel (+GSW)   traceback.hip.h  traceback_kernel, gsw_traceback_kernel
//   fp_walk.hip.h      fp_walk_kernel & co        aux_kernels.hip.h score_matrix / scale_runs / scan kernels
//   const_long.hip.h   cl_sweep_kernel / cl_walk_kernel: constant-gap pairs without a stored direction matrix (config C5)
//   affine_long.hip.h  the same scheme for affine pairs whose matrix does not fit     seed_kernels.hip.h  the graph aligner's index / seed search
//   lat_fill.hip.h     lat_fill_kernel: one pair per wave (64 lanes x 2 rows), launches of few long pairs
//   gnx_host.hip.h     host-buffer entry points: pinned staging, sub-batches, resident reference, contexts per GPU, RCCL
//   gnx_align.hip      host orchestration + C ABI (this file)
#include "gnx_common.hip.h"
#include "fill_affine.hip.h"
#include "fp_sweep.hip.h"
#include "fill_const.hip.h"
#include "traceback.hip.h"
#include "fp_walk.hip.h"
#include "aux_kernels.hip.h"
#include "const_long.hip.h"
#include "const_long_wg.hip.h"
#include "const_long_walk.hip.h"
#include "affine_long.hip.h"
#include "affine_long64.hip.h"
#include "const_long64.hip.h"
#include "farm64.hip.h"
#include "lat_fill.hip.h"
#include "lat_wide.hip.h"
#include "seed_kernels.hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------------
// Error text: thread local (an error belongs to the call that failed), plus a process-wide copy of the most recent one for callers
// whose runtime may fetch it on another OS thread (cgo without runtime.LockOSThread).
thread_local char g_err[512] = "";
std::mutex g_lasterr_mu;
char g_lasterr[512] = "";
void publish_err() { std::lock_guard<std::mutex> lk(g_lasterr_mu); memcpy(g_lasterr, g_err, sizeof(g_lasterr)); }
void set_err(const char *fmt, const char *a = "", long long b = 0) { snprintf(g_err, sizeof(g_err), fmt, a, b); publish_err(); }
void set_err_hip(hipError_t e, const char *file, int line) { snprintf(g_err, sizeof(g_err), "HIP error %s at %s:%d", hipGetErrorString(e), file, line); publish_err(); }

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) { set_err_hip(e_, __FILE__, __LINE__); return GNX_EDEVICE; }             \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // What an allocation costs on this system (tools/alloc_probe.hip, profiles/r6_alloc_probe.txt): ~0.02 ms per GB when the driver hands out memory that has been
    // free for a while, 20 - 55 ms per GB when it hands out memory that was freed a moment ago (by this process or the one before it) -- freed memory is cleared in
    // the background and an allocation waits for the clearing of what it gets.  So a buffer that grows is allocated BEFORE the old one is freed (the new one comes
    // from memory that has long been clean; free-then-allocate got the just-freed pages back and waited: the "27 ms per GB" of round 5), and big buffers take no
    // 1/8 of slack along.
    int ensure(size_t bytes) {
        if (bytes <= cap) return GNX_OK;
        size_t want = bytes + (bytes < ((size_t)1 << 30) ? bytes / 8 + 256 : bytes / 64);
        void *q = nullptr;
        if (hipMalloc(&q, want) == hipSuccess) { // beside the old buffer
            if (p) (void)hipFree(p);
            p = q; cap = want;
            return GNX_OK;
        }
        (void)hipGetLastError(); // (the failed attempt must not stay behind as the "last error" a later launch check would read)
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; set_err("device allocation of %s%lld bytes failed", "", (long long)bytes); return GNX_ENOMEM; }
            want = bytes;
        }
        cap = want;
        return GNX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
// pinned host staging (hipHostMalloc): copies to / from it are asynchronous DMA, and the stager thread fills it while kernels run
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return GNX_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; set_err("pinned host allocation of %s%lld bytes failed", "", (long long)want); return GNX_ENOMEM; }
        cap = want;
        return GNX_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// One context per device (gnx_init binds context 0; gnx_init_devices creates one per GPU of the node).  A context owns its streams,
// events and every device buffer; entry points lock the contexts they use, and a context's work always runs on one host thread at a
// time (the calling thread, or one worker thread per context for the multi-device entry points): t_ctx is that thread's context.
struct Ctx {
    std::mutex mu;
    bool inited = false;
    int index = 0;
    int device = -1;
    int64_t ws_limit = 0;
    int n_cu = 256; // compute units of the device
    hipStream_t own_stream = nullptr, s_in = nullptr;
    DevBuf trace, hcol, rowbuf, dcol, plans, nops, misc;
    DevBuf strip_map, tb_scr, tb_scr_off, scan_tmp, fp_tail, fp_rowi, fp_ckpt, fp_states, fp_stage, fp_wplans[2], fp_active[2], fp_thcol, fp_ttrace, fp_redo, fp_strag, mx_idx, mx_tab, mx_score, mx_off, mx_ops, fp_prog;
    DevBuf in_a, in_b, in_as, in_al, in_bs, in_bl, out_score, out_off, out_ops, out_end, sc_pairs, sc_mat, sc_err;
    // pipelined host entry (gnx_host.hip.h): double-buffered inputs, results accumulated on the device, the resident reference
    DevBuf pin_a[2], pin_as[2], pin_b[2], pin_bs[2], res_score, res_off, res_ops, ref, gat_score, gat_off, gat_ops;
    // resident seed index of the graph aligner (gnx_seed_index_set)
    DevBuf sd_keys, sd_locs, sd_nodes, sd_node_off, sd_word_off, sd_words, sd_tmp[8];
    int64_t sd_n = -1, sd_nodes_n = 0; int sd_seed_len = 0;
    PinBuf h_plans; // host-side plans of the general path
    PinBuf st_a[2], st_as[2], st_b[2], st_bs[2];
    // the resident reference, PACKED (gnx_host.hip.h: pack_reference): `ref` = 2 bits per base, ref_flag / ref_rank / ref_exc = the
    // sparse list of bases that are not A C G T (KParams::b2 / bflag / brank / bexc)
    DevBuf ref_flag, ref_rank, ref_exc, unpk_b, unpk_off, cl_bases, sc_prof_a, sc_prof_b, mega_rows, mega_state, farm, mega_arena;
    int64_t ref_len = -1;  // >= 0: a reference of that many bases is resident
    int64_t ref_nexc = 0;  // 64-base blocks with an exception
    int64_t ref_epoch = 0; // which gnx_set_reference call filled it (contexts created later are brought up to date on first use)
    bool beta_packed = false; // set by the host flow around run_device: beta windows index the packed reference of this context
    hipEvent_t ev_in[2] = {nullptr, nullptr};
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    gnx_timing timing = {};
    // the fast path's plans of the previous call, still in `plans` on the device: batches of a stream of reads keep the same
    // lengths (C2-C4: every pair 150 x 10 000), and building + uploading 100 k plans (9.6 MB from pageable memory) costs ~0.6 ms
    std::vector<int64_t> fpc_alen, fpc_blen;
    int64_t fpc_roff = 0, fpc_coff = 0, fpc_cells = 0, fpc_mmax = 1, fpc_rboff = 0;
    int fpc_strips = 1;
    const void *fpc_ptr = nullptr; // == plans.p while the cached plans are what the device holds
};
std::mutex g_ctxs_mu;          // guards the list itself
std::vector<Ctx *> g_ctxs;     // [0] = the default context; never shrinks while the library is loaded
thread_local Ctx *t_ctx = nullptr;
// set while a call is re-run with sequential strips after a pipelined launch timed out waiting for a row buffer (a bug trap: nobody
// waits for an item that has not been claimed, claim_items; should the trap ever fire, the call still returns right results)
thread_local bool t_no_pipe = false;
thread_local int t_w64_ck = 128; // snapshot spacing of the 64-lane affine sweep chosen by the routing for this call (farm64.hip.h: 128 .. 512)
std::atomic<int64_t> g_last_w64_r{0}, g_last_w64_ck{0}; // what the last 64-lane affine sweep ran with (gnx_debug_counter(5 / 6): bench.py prices its bytes with them)
thread_local bool t_scored_wide = false; // run_host_scored: this call's score matrices are plain 4 * s entries for the int64 kernel (a pair beyond the int32 range)
thread_local int t_w64_rc = R;   // ... and of the 64-lane CONSTANT-GAP sweep + farm (4 / 10)
thread_local int t_w64_r = R;    // rows per lane of the 64-lane AFFINE sweep + farm chosen by the routing for this call (w64_pick_rows: 6 / 8 / 10 / 16)
thread_local bool t_no_lat = false; // set while a call is re-run without the latency geometry (its bug trap fired)
thread_local int64_t t_min_cols = 0; // != 0: the fast path takes windows of at least this many columns (set by a mixed batch for its groups, see run_device)
bool no_pipe() { return t_no_pipe || getenv("GNX_NO_PIPE") != nullptr; }
bool w64_two_waves() { const char *e = getenv("GNX_W64_SPEC"); return !(e && e[0] == '0'); }
Ctx &ctx_at(int k) {
    std::lock_guard<std::mutex> lk(g_ctxs_mu);
    while ((int)g_ctxs.size() <= k) { Ctx *c = new Ctx; c->index = (int)g_ctxs.size(); g_ctxs.push_back(c); }
    return *g_ctxs[(size_t)k];
}
int ctx_count() { std::lock_guard<std::mutex> lk(g_ctxs_mu); int n = 0; for (Ctx *c : g_ctxs) if (c->inited) n++; return n; }
#define g_ctx (*t_ctx)
// binds a context to the calling thread for the duration of an entry point (locks it, makes its device current)
struct CtxScope {
    Ctx *prev;
    Ctx &c;
    std::unique_lock<std::mutex> lk;
    explicit CtxScope(Ctx &cc) : prev(t_ctx), c(cc), lk(cc.mu) { t_ctx = &cc; }
    ~CtxScope() { t_ctx = prev; }
};

int init_ctx(Ctx &c, int device, int64_t workspace_bytes) {
    HIPCHK(hipSetDevice(device));
    c.device = device;
    HIPCHK(hipStreamCreate(&c.own_stream));
    HIPCHK(hipStreamCreateWithFlags(&c.s_in, hipStreamNonBlocking));
    for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&c.ev[i]));
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&c.ev_in[i], hipEventDisableTiming));
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c.n_cu = v; }
    if (workspace_bytes > 0) c.ws_limit = workspace_bytes;
    if (c.ws_limit <= 0) {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        c.ws_limit = (int64_t)std::min<size_t>(fr / 2, (size_t)64 << 30);
    }
    c.inited = true;
    return GNX_OK;
}

// the current context, initialised on first use: context 0 binds device LOCAL_RANK (one process per GPU under torch.distributed.run)
int ensure_init() {
    Ctx &c = g_ctx;
    if (c.inited) { HIPCHK(hipSetDevice(c.device)); return GNX_OK; }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { set_err("no HIP device available (this library has no CPU fallback)%s", ""); return GNX_EDEVICE; }
    int dev = 0;
    const char *lr = getenv("LOCAL_RANK");
    if (lr && *lr) dev = atoi(lr) % cnt;
    return init_ctx(c, dev, 0);
}

int check_params(const gnx_params *p, KParams &kp, TbParams &tp, bool &affine, bool &local, bool &lowmem) {
    if (!p) { set_err("null params%s", ""); return GNX_EINVAL; }
    affine = (p->mode == GNX_AFFINE_GAP || p->mode == GNX_AFFINE_GAP_HIGHMEM || p->mode == GNX_AFFINE_GAP_LOCAL);
    const bool cst = (p->mode == GNX_CONST_GAP || p->mode == GNX_CONST_GAP_HIGHMEM);
    if (!affine && !cst) { set_err("unknown mode %s%lld", "", (long long)p->mode); return GNX_EINVAL; }
    local = (p->mode == GNX_AFFINE_GAP_LOCAL);
    lowmem = (p->mode == GNX_AFFINE_GAP || p->mode == GNX_CONST_GAP);
    if (lowmem && (p->checkersize_i < 1 || p->checkersize_j < 1)) { set_err("checkersize must be >= 1%s", ""); return GNX_EINVAL; }
    const int64_t lim = (int64_t)1 << 26;
    for (int x = 0; x < 25; x++) if (p->scores[x] > lim || p->scores[x] < -lim) { set_err("score out of int32 kernel range%s", ""); return GNX_ERANGE; }
    if (p->gap_open > lim || p->gap_open < -lim || (affine && (p->gap_extend > lim || p->gap_extend < -lim))) { set_err("gap penalty out of int32 kernel range%s", ""); return GNX_ERANGE; }
    kp.b2 = nullptr; kp.bflag = nullptr; kp.brank = nullptr; kp.bexc = nullptr;
    for (int x = 0; x < 25; x++) kp.sc4[x] = (int)(4 * p->scores[x]);
    kp.o4 = (int)(4 * p->gap_open);
    kp.e4 = affine ? (int)(4 * p->gap_extend) : 0;
    kp.oe4 = kp.o4 + kp.e4;
    kp.d00_4 = local ? 0 : kp.o4;
    kp.ecol4 = local ? 0 : kp.e4;
    kp.g4 = kp.o4;
    kp.rb_pub = RB_PUB;
    kp.ckc = CKC_SMALL;
    tp.ci = lowmem ? p->checkersize_i : ((int64_t)1 << 62);
    tp.cj = lowmem ? p->checkersize_j : ((int64_t)1 << 62);
    tp.d00 = local ? 0 : p->gap_open;
    tp.ecol = local ? 0 : p->gap_extend;
    tp.gap_open = p->gap_open;
    tp.gap_extend = affine ? p->gap_extend : 0;
    tp.affine = affine ? 1 : 0;
    return GNX_OK;
}

int64_t max_abs_pen(const gnx_params *p, bool affine) {
    int64_t mx = 0;
    for (int x = 0; x < 25; x++) mx = std::max<int64_t>(mx, llabs((long long)p->scores[x]));
    if (affine) mx = std::max<int64_t>(mx, llabs((long long)(p->gap_open + p->gap_extend)));
    mx = std::max<int64_t>(mx, llabs((long long)p->gap_open));
    if (affine) mx = std::max<int64_t>(mx, llabs((long long)p->gap_extend));
    return mx;
}

// Fast path for batches of short-alpha global affine alignments (see fp_walk_kernel).  Returns GNX_OK, an error,
// or -1 when the batch should go through the general path after all (workspace too small / staging overflow).
// `first` = first sub-batch of a call: later sub-batches keep the error flags and the CIGAR offset carry.
int run_device(const gnx_params *prm, int64_t n_pairs, const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
               const int64_t *h_alen, const int64_t *h_blen, int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off,
               int64_t *out_total, hipStream_t stream, const int *d_smat, const int64_t *h_soff, int gsw, int2 *d_endpos, bool no_fast_path, bool smat16);

// holds a CU's whole LDS and spins until `ticks` of the 100 MHz wall clock have passed (gnx_debug_occupy)
__global__ __launch_bounds__(64) void occupy_kernel(long long ticks) {
    extern __shared__ int occupy_lds[];
    occupy_lds[threadIdx.x] = (int)ticks;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// GNX_TICKET_DELAY=k (tests): the lower half of a piped grid sleeps k x 127 x 64 cycles before it claims its items (claim_items), so the
// upper half finds its predecessors unclaimed and runs them itself.  sw = the switch word behind the claim words (already zeroed).
int claim_test_switch(int *sw, hipStream_t st) {
    const char *e = getenv("GNX_TICKET_DELAY"), *g = getenv("GNX_CLAIM_GRACE_US"); // (grace: only together with a delay, see claim_items)
    if (!e || !*e) return GNX_OK;
    const int v = (std::min(std::max(atoi(e), 0), 0xffff)) | ((g && *g) ? (std::min(std::max(atoi(g), 1), 0x7fff) << 16) : 0);
    HIPCHK(hipMemcpyAsync(sw, &v, 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return GNX_OK;
}

// exclusive scan of the run counts: off[0..n], carry[0] in / out (see scan_kernel); three launches when the array is long
int launch_scan(const int64_t *d_nops, int n, int64_t *d_off, int64_t *d_carry, hipStream_t stream) {
    Ctx &c = g_ctx;
    if (n <= 8192) { hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, d_nops, n, d_off, d_carry); return GNX_OK; }
    const int nb = (n + 1023) / 1024;
    int rc;
    if ((rc = c.scan_tmp.ensure((size_t)(2 * nb + 2) * 8))) return rc;
    int64_t *sums = reinterpret_cast<int64_t *>(c.scan_tmp.p), *boff = sums + nb;
    hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)nb), dim3(1024), 0, stream, d_nops, n, sums);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, sums, nb, boff, d_carry); // boff[nb] = new carry
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(1024), 0, stream, d_nops, n, boff, d_off);
    return GNX_OK;
}

int run_device_fp(const gnx_params *prm, const KParams &kp, const TbParams &tp, int64_t n_pairs,
                  const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
                  const int64_t *h_alen, const int64_t *h_blen, int rows_per_lane,
                  int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off,
                  int64_t *out_total, hipStream_t stream, bool first, bool xp, int S) {
    // rows_per_lane: 19 (every n <= 152) or 20 (n <= 160) -> fp_sweep_kernel<19 / 20>
    // S >= 2: every read has 160 (S - 1) + 1 .. 160 S bases: swept as S row blocks in one launch (fp_sweep_levels_kernel: the levels
    //     follow each other through the row buffer); the walk re-fills one row block per window / tile
    // xp: AffineGapLocal, transposed -- the caller passes the query as "a" (rows) and the target as "b" (columns), and kp holds the
    //     transposed score table with the column-0 boundary of a global alignment (fp_sweep_kernel<.., true> and friends)
    Ctx &c = g_ctx;
    int rc;
    const int np = (int)n_pairs;
    const bool two = S >= 2; // (several row blocks)
    const int CAP = fp_cap(S);
    int64_t top_hi = 1; // S >= 2: the longest top block of the batch (rows above the last 160 (S - 1))
    for (int64_t p = 0; two && p < n_pairs; p++) top_hi = std::max<int64_t>(top_hi, h_alen[p] - (int64_t)H * (S - 1));
    const bool cached = c.fpc_ptr && c.fpc_ptr == c.plans.p && (int64_t)c.fpc_alen.size() == n_pairs && c.fpc_strips == S &&
                        memcmp(c.fpc_alen.data(), h_alen, (size_t)n_pairs * 8) == 0 && memcmp(c.fpc_blen.data(), h_blen, (size_t)n_pairs * 8) == 0;
    std::vector<PairPlan> plans(cached ? 0 : (size_t)n_pairs);
    int64_t roff = 0, coff = 0, cells = 0, m_maxb = 1, rboff = 0;
    if (cached) { roff = c.fpc_roff; coff = c.fpc_coff; cells = c.fpc_cells; m_maxb = c.fpc_mmax; rboff = c.fpc_rboff; }
    for (int64_t p = 0; !cached && p < n_pairs; p++) {
        PairPlan &pl = plans[(size_t)p];
        const int64_t n = h_alen[p], m = h_blen[p];
        pl.n = (int32_t)n; pl.m = (int32_t)m; pl.words = (int32_t)((m + 15 + 15) / 16); pl.strips = S;
        pl.trace_off = 0; pl.hcol_off = p; pl.rowbuf_off = rboff; pl.dcol_off = 0;
        if (two) rboff += (int64_t)(S - 1) * (m + 1); // the bottom rows of the S - 1 blocks that hand one down
        pl.src = (int32_t)p; pl.col_off = 0; pl.ckpt_off = coff; pl.rowi_off = roff; pl.s_off = 0; pl.s_pitch = 0;
        roff += (int64_t)FP_PLANES * pl.words; coff += ((m - 1) / CKW) * n; cells += n * m;
        m_maxb = std::max(m_maxb, m);
    }
    // windows per request (fp_walk_kernel): reads of several row blocks get speculative windows for the next blocks up as well
    const char *spenv = getenv("GNX_FP_SPEC");
    // (batches of up to 32 768 such reads, whose window walk runs one wave per pair: with lanes walking on their own the rounds are
    // not what costs -- 100 000 x (1000 x 1200): 2.97e12 cells/s with lanes and single windows, 2.71e12 with waves and 4 windows)
    const bool wide_walk = S > 1 && np <= 32768 && !getenv("GNX_WALK_LANE");
    const int K = wide_walk ? std::max(1, std::min(std::min(FP_SPEC, S), spenv ? atoi(spenv) : FP_SPEC)) : 1;
    const int WW = fp_spec_wwords(K);
    const size_t wtrace_b = (size_t)np * K * WW * QA * G * 16;
    const size_t need = wtrace_b + (size_t)coff * 8 + (size_t)roff * 4 + (size_t)rboff * 8 +
                        (size_t)np * (CAP * sizeof(gnx_cigar) + sizeof(FpState) + (1 + 2 * K) * sizeof(PairPlan) + (size_t)K * (H * 4 + G * 4) + 40);
    if ((int64_t)need > c.ws_limit) { if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] working set %zu B exceeds the workspace limit -> general path\n", need); return -1; }
    if ((rc = c.trace.ensure(wtrace_b))) return rc;
    if ((rc = c.hcol.ensure((size_t)np * ((size_t)K * H + 1) * 4))) return rc;   // [0,np) h(n,m) of the forward sweep, then the window hcol slots
    if ((rc = c.dcol.ensure((size_t)np * K * G * 4))) return rc;
    if ((rc = c.fp_strag.ensure((size_t)np * 2 * 4 + 64))) return rc;   // stragglers of this round / of the next one
    if ((rc = c.rowbuf.ensure((size_t)std::max<int64_t>(rboff, 1) * 8))) return rc;   // what each row block hands to the one below it
    if (two && (rc = c.fp_prog.ensure((size_t)S * ((np + G8 - 1) / G8) * 8 + 64))) return rc; // progress words and claim words of the levels' waves, test switch
    if ((rc = c.plans.ensure((size_t)np * sizeof(PairPlan)))) return rc;
    if ((rc = c.nops.ensure((size_t)np * 8))) return rc;
    if ((rc = c.misc.ensure(64))) return rc;
    if ((rc = c.fp_rowi.ensure((size_t)std::max<int64_t>(roff, 1) * 4))) return rc;
    if ((rc = c.fp_tail.ensure((size_t)np * 4 * FP_TAILW))) return rc;
    if ((rc = c.fp_ckpt.ensure((size_t)std::max<int64_t>(coff, 1) * 8))) return rc;
    if ((rc = c.fp_states.ensure((size_t)np * sizeof(FpState)))) return rc;
    if ((rc = c.fp_stage.ensure((size_t)np * CAP * sizeof(gnx_cigar)))) return rc;
    for (int x = 0; x < 2; x++) {
        if ((rc = c.fp_wplans[x].ensure((size_t)np * K * sizeof(PairPlan)))) return rc;
        if ((rc = c.fp_active[x].ensure((size_t)np * 4))) return rc;
    }
    int *d_err = reinterpret_cast<int *>(c.misc.p);
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    int *d_cnt = reinterpret_cast<int *>(reinterpret_cast<char *>(c.misc.p) + 32); // window-request counters (ping-pong)
    if (first) HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, stream));
    else HIPCHK(hipMemsetAsync(d_cnt, 0, 16, stream));
    if (!cached || c.fpc_ptr != c.plans.p) { // (ensure() above may have moved the buffer)
        if (cached) { set_err("internal: plan cache lost its buffer%s", ""); return GNX_EINVAL; }
        HIPCHK(hipMemcpyAsync(c.plans.p, plans.data(), (size_t)np * sizeof(PairPlan), hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream)); // `plans` is pageable: the copy is staged, but do not rely on it
        c.fpc_alen.assign(h_alen, h_alen + n_pairs); c.fpc_blen.assign(h_blen, h_blen + n_pairs);
        c.fpc_roff = roff; c.fpc_coff = coff; c.fpc_cells = cells; c.fpc_mmax = m_maxb; c.fpc_ptr = c.plans.p; c.fpc_rboff = rboff; c.fpc_strips = S;
    }
    const PairPlan *dpl = reinterpret_cast<const PairPlan *>(c.plans.p);
    int *d_hfwd = reinterpret_cast<int *>(c.hcol.p);
    int *d_whcol = d_hfwd + np;
    unsigned *d_rowi = reinterpret_cast<unsigned *>(c.fp_rowi.p);
    unsigned *d_tail = reinterpret_cast<unsigned *>(c.fp_tail.p);
    int2 *d_ckpt = reinterpret_cast<int2 *>(c.fp_ckpt.p);
    FpState *d_st = reinterpret_cast<FpState *>(c.fp_states.p);
    gnx_cigar *d_stage = reinterpret_cast<gnx_cigar *>(c.fp_stage.p);
    int64_t *d_nops = reinterpret_cast<int64_t *>(c.nops.p);
    PairPlan *d_wpl[2] = {reinterpret_cast<PairPlan *>(c.fp_wplans[0].p), reinterpret_cast<PairPlan *>(c.fp_wplans[1].p)};
    int *d_act[2] = {reinterpret_cast<int *>(c.fp_active[0].p), reinterpret_cast<int *>(c.fp_active[1].p)};
    const dim3 blockF(64), blockT(64);
    // window rounds continue while they pay: a fixed number with GNX_FP_MAXIT, else while more than 4 % of the pairs are waiting
    // (a round has a fixed latency of ~0.2 ms; the few pairs left over go to the tile re-fill, whose cost is per pair)
    const int max_it = getenv("GNX_FP_MAXIT") ? atoi(getenv("GNX_FP_MAXIT")) : -1;
    const int tiles_per = (int)((m_maxb + FP_TILE - 1) / FP_TILE);
    double refill_ms = 0;

    auto forward = [&](int p0, int cnt, hipStream_t st) -> int {
        const dim3 grid8((unsigned)((cnt + G8 - 1) / G8));
        if (two) { // S row blocks ("levels"): one launch, each level following the one above it through the row buffer (GNX_NO_PIPE: a launch per level)
            int2 *rb = reinterpret_cast<int2 *>(c.rowbuf.p);
            const int top_rows = (int)(top_hi + G8 - 1) / G8; // slots per lane of the top block: as few as hold the longest read's rows above the full blocks
            const bool pk = kp.b2 != nullptr; // beta = the packed resident reference (never with xp: AffineGapLocal gets byte windows)
            auto klev = xp ? (top_rows <= 8 ? fp_sweep_levels_kernel<8, true> : (top_rows <= 12 ? fp_sweep_levels_kernel<12, true> : (top_rows <= 16 ? fp_sweep_levels_kernel<16, true> : fp_sweep_levels_kernel<20, true>)))
                      : pk ? (top_rows <= 8 ? fp_sweep_levels_kernel<8, false, true> : (top_rows <= 12 ? fp_sweep_levels_kernel<12, false, true> : (top_rows <= 16 ? fp_sweep_levels_kernel<16, false, true> : fp_sweep_levels_kernel<20, false, true>)))
                           : (top_rows <= 8 ? fp_sweep_levels_kernel<8> : (top_rows <= 12 ? fp_sweep_levels_kernel<12> : (top_rows <= 16 ? fp_sweep_levels_kernel<16> : fp_sweep_levels_kernel<20>)));
            const int W = (int)grid8.x;
            int *prog = reinterpret_cast<int *>(c.fp_prog.p);
            if (!no_pipe()) {
                HIPCHK(hipMemsetAsync(prog, 0, ((size_t)S * W * 2 + 2) * 4, st)); // progress words, claim words, test switch
                if ((rc = claim_test_switch(prog + (size_t)S * W * 2, st))) return rc;
                hipLaunchKernelGGL(klev, dim3((unsigned)(S * W)), blockF, 0, st, dpl + p0, cnt, d_a, d_as, d_b, d_bs, kp, d_hfwd, d_ckpt, d_rowi, d_tail, d_err, rb, S, W, 0, 1, prog);
                HIPCHK(hipGetLastError());
                int e = 0; // a level that waited 5 s for the one above it (a bug trap: the level above is always claimed by a running workgroup): sweep again, level by level
                HIPCHK(hipMemcpyAsync(&e, d_err, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (!(e & 16)) return GNX_OK;
                e &= ~16;
                HIPCHK(hipMemcpyAsync(d_err, &e, 4, hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
                if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] a row block timed out waiting for the one above it -> one launch per level\n");
            }
            for (int level = 0; level < S; level++)
                hipLaunchKernelGGL(klev, grid8, blockF, 0, st, dpl + p0, cnt, d_a, d_as, d_b, d_bs, kp, d_hfwd, d_ckpt, d_rowi, d_tail, d_err, rb, S, W, level, 0, prog);
            HIPCHK(hipGetLastError());
            return GNX_OK;
        }
        auto k = xp ? (rows_per_lane == 19 ? fp_sweep_kernel<19, true> : fp_sweep_kernel<20, true>)
                 : kp.b2 ? (rows_per_lane == 19 ? fp_sweep_kernel<19, false, true> : fp_sweep_kernel<20, false, true>)
                         : (rows_per_lane == 19 ? fp_sweep_kernel<19, false> : fp_sweep_kernel<20, false>);
        hipLaunchKernelGGL(k, grid8, blockF, 0, st, dpl + p0, cnt, d_a, d_as, d_b, d_bs, kp, d_hfwd, d_ckpt, d_rowi, d_tail, d_err);
        HIPCHK(hipGetLastError());
        return GNX_OK;
    };
    // walk / window stages of the pairs [p0, p0+cnt) on stream `st`; their window slots are [p0, p0+cnt) as well.
    // Rounds: re-fill the windows the walks asked for (one row block of one pair each), walk on; a walk that used up its window inside
    // the same row block is a straggler (a long gap on a row without a stored plane) and gets all remaining columns of that block as
    // tiles in the same round; walks that leave a row block through its top ask for a window of the block above.  A read of S row
    // blocks takes S rounds (plus those of its stragglers).  cnt2: [0], [1] = window requests (ping-pong), [2] = stragglers.
    auto post = [&](int p0, int cnt, hipStream_t st, int *cnt2, hipEvent_t e1, hipEvent_t e2) -> int {
        uint4 *wtr = reinterpret_cast<uint4 *>(c.trace.p) + (int64_t)p0 * K * WW * QA * G;
        int *whc = d_whcol + (int64_t)p0 * K * H;
        unsigned *wdc = reinterpret_cast<unsigned *>(c.dcol.p) + (int64_t)p0 * K * G;
        PairPlan *wpl[2] = {d_wpl[0] + (int64_t)p0 * K, d_wpl[1] + (int64_t)p0 * K};
        int2 *srb = reinterpret_cast<int2 *>(c.rowbuf.p); // the sweep's row buffer: the rows the blocks handed down
        int *d_strag = reinterpret_cast<int *>(c.fp_strag.p) + p0;
        int cur = 0, n_act = 0, it = 0;
        float f = 0;
        // The stragglers' walk through their tiles runs one WAVE per pair (lanes that walk alone diverge: 0.80 -> 0.5 ms for 1 700
        // stragglers, and a long gap is taken 64 tile words per look); the first walk and the window walk stay one lane per pair
        // (100 000 waves of redundant scalar work cost more than they save: 0.45 -> 1.14 ms).  GNX_WALK_LANE=1: lanes everywhere;
        // GNX_WALK_CW=1: waves for the first walk as well (A/B runs).
        const bool cw = !getenv("GNX_WALK_LANE");
        const bool cw_first = cw && getenv("GNX_WALK_CW");
        auto k_first = cw_first ? (xp ? fp_walk_kernel<true, false, true, true> : fp_walk_kernel<true, false, false, true>) : (xp ? fp_walk_kernel<true, false, true> : fp_walk_kernel<true, false, false>);
        // reads of several row blocks: the window walk with one wave per pair as well -- a long read's walk is thousands of dependent
        // loads for a lane on its own, and its diagonal runs are taken 64 cells per look (GNX_WALK_LANE=1: lanes)
        // (round 4: whatever the batch size -- with the block shortcut a window walk is mostly sums along diagonals, which a wave takes 64
        // cells at a time from coalesced loads; the speculative windows stay with the batches of <= 32 768 reads.  GNX_WALK_WIDE=0: as before)
        const bool cw_next = cw && (wide_walk || (S > 1 && !(getenv("GNX_WALK_WIDE") && atoi(getenv("GNX_WALK_WIDE")) == 0)));
        auto k_next = cw_next ? (xp ? fp_walk_kernel<false, false, true, true> : fp_walk_kernel<false, false, false, true>) : (xp ? fp_walk_kernel<false, false, true> : fp_walk_kernel<false, false, false>);
        auto k_tiled = cw ? (xp ? fp_walk_kernel<false, true, true, true> : fp_walk_kernel<false, true, false, true>) : (xp ? fp_walk_kernel<false, true, true> : fp_walk_kernel<false, true, false>);
        auto wgrid = [&](int n) { return dim3((unsigned)(cw ? n : (n + 63) / 64)); };
        auto k_win = xp ? fill_affine_kernel<false, false, false, true, true, false, true> : fill_affine_kernel<false, false, false, true, true, false, false>;
        // GNX_FP_MAXIT = k: after k window rounds every walk that asks for more is treated as a straggler (tests: the tile path for everyone)
        auto force = [&](int round) { return (max_it >= 0 && round >= max_it) ? 1 : 0; };
        int h_cnt[4] = {0, 0, 0, 0};
        HIPCHK(hipMemsetAsync(cnt2, 0, 16, st));
        hipLaunchKernelGGL(k_first, dim3((unsigned)(cw_first ? cnt : (cnt + 63) / 64)), blockT, 0, st, dpl, (const int *)nullptr, cnt, d_st, d_hfwd, d_rowi, d_tail,
                           (const PairPlan *)nullptr, wtr, whc, tp, d_stage, d_score, d_nops, d_act[0] + p0, cnt2, wpl[0], d_err, p0, d_strag, cnt2 + 2, force(0), srb, K, WW, d_a, d_as, d_b, d_bs, kp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_cnt, cnt2, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        n_act = h_cnt[0];
        int n_strag = h_cnt[2];
        const int max_rounds = 6 * S + 32;
        while (n_act > 0 || n_strag > 0) {
            if (it >= max_rounds) { if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] %d pairs still walking after %d rounds -> general path\n", n_act + n_strag, it); return -1; }
            // n_act window requests in list `cur`, n_strag stragglers; what they ask for next goes to list `nxt`
            const int nxt = cur ^ 1;
            double round_ms = 0;
            HIPCHK(hipMemsetAsync(cnt2 + nxt, 0, 4, st));
            HIPCHK(hipMemsetAsync(cnt2 + 2, 0, 4, st));
            int n_strag2 = 0;
            if (n_act > 0) {
                HIPCHK(hipEventRecord(e1, st));
                hipLaunchKernelGGL(k_win, dim3((unsigned)(((int64_t)n_act * K + 3) / 4)), blockF, 0, st, wpl[cur], n_act * K, d_a, d_as, d_b, d_bs, kp,
                                   wtr, whc, srb, wdc, d_ckpt, d_err, (const int *)nullptr, (const int2 *)nullptr, (int *)nullptr);
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(e2, st));
                // (its stragglers go to a second list: d_strag holds this round's while they are being served)
                hipLaunchKernelGGL(k_next, dim3((unsigned)(cw_next ? n_act : (n_act + 63) / 64)), blockT, 0, st, dpl, d_act[cur] + p0, n_act, d_st, d_hfwd, d_rowi, d_tail,
                                   wpl[cur], wtr, whc, tp, d_stage, d_score, d_nops, d_act[nxt] + p0, cnt2 + nxt, wpl[nxt], d_err, 0,
                                   d_strag + cnt, cnt2 + 2, force(it + 1), srb, K, WW, d_a, d_as, d_b, d_bs, kp);
                HIPCHK(hipGetLastError());
            }
            if (n_strag > 0 && getenv("GNX_DEBUG") && atoi(getenv("GNX_DEBUG")) >= 2) { // census: where the stragglers stand (row distance from n, state, column)
                std::vector<int> hs((size_t)n_strag);
                HIPCHK(hipStreamSynchronize(st));
                HIPCHK(hipMemcpy(hs.data(), d_strag, (size_t)n_strag * 4, hipMemcpyDeviceToHost));
                std::vector<PairPlan> hp((size_t)cnt);
                HIPCHK(hipMemcpy(hp.data(), dpl, (size_t)cnt * sizeof(PairPlan), hipMemcpyDeviceToHost));
                int hist_d[8] = {0}, hist_k[3] = {0}, far = 0;
                for (int x = 0; x < n_strag; x++) {
                    FpState fs;
                    HIPCHK(hipMemcpy(&fs, d_st + hs[(size_t)x], sizeof(fs), hipMemcpyDeviceToHost));
                    const int dd = hp[(size_t)hs[(size_t)x]].n - fs.i;
                    hist_d[dd < 7 ? dd : 7]++; hist_k[fs.k < 3 ? fs.k : 2]++;
                    if (fs.j > 600) far++;
                    if (x < 12) fprintf(stderr, "[gnx fp]   straggler pair %d: i %d (n - i = %d) j %d state %d last_op %d runs so far %d\n", hs[(size_t)x], fs.i, dd, fs.j, fs.k, fs.last_op, fs.cnt);
                }
                fprintf(stderr, "[gnx fp] straggler census: n - i = 0..6, >= 7: %d %d %d %d %d %d %d %d; state M / I / D: %d %d %d; with more than 600 columns left: %d of %d\n",
                        hist_d[0], hist_d[1], hist_d[2], hist_d[3], hist_d[4], hist_d[5], hist_d[6], hist_d[7], hist_k[0], hist_k[1], hist_k[2], far, n_strag);
            }
            if (n_strag > 0) { // all remaining columns of the stragglers' row blocks as independent tiles, one launch
                const int64_t n_tiles = (int64_t)n_strag * tiles_per;
                const size_t tb = (size_t)n_tiles * FP_TWORDS * QA * G * 16;
                if ((int64_t)tb > c.ws_limit || n_tiles > 0x3fffffff) { if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] %lld straggler tiles exceed the workspace limit -> general path\n", (long long)n_tiles); return -1; }
                int rc2;
                if ((rc2 = c.fp_redo.ensure((size_t)n_tiles * sizeof(PairPlan)))) return rc2; // tile plans (in a buffer this path does not use otherwise)
                if ((rc2 = c.fp_thcol.ensure((size_t)n_tiles * (H + G) * 4))) return rc2;
                if ((rc2 = c.fp_ttrace.ensure(tb))) return rc2;
                PairPlan *tpl = reinterpret_cast<PairPlan *>(c.fp_redo.p);
                int *thc = reinterpret_cast<int *>(c.fp_thcol.p);
                unsigned *tdc = reinterpret_cast<unsigned *>(thc + n_tiles * H);
                uint4 *ttr = reinterpret_cast<uint4 *>(c.fp_ttrace.p);
                hipLaunchKernelGGL(fp_straggler_plans_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, st, dpl, d_strag, n_strag, tiles_per, d_st, tpl);
                HIPCHK(hipEventRecord(c.ev[6], st));
                hipLaunchKernelGGL(k_win, dim3((unsigned)((n_tiles + 3) / 4)), blockF, 0, st, tpl, (int)n_tiles, d_a, d_as, d_b, d_bs, kp,
                                   ttr, thc, srb, tdc, d_ckpt, d_err, (const int *)nullptr, (const int2 *)nullptr, (int *)nullptr);
                HIPCHK(hipEventRecord(c.ev[7], st));
                hipLaunchKernelGGL(k_tiled, wgrid(n_strag), blockT, 0, st, dpl, d_strag, n_strag, d_st, d_hfwd, d_rowi, d_tail,
                                   tpl, ttr, thc, tp, d_stage, d_score, d_nops, d_act[nxt] + p0, cnt2 + nxt, wpl[nxt], d_err, 0,
                                   d_strag + cnt, cnt2 + 2, 0, srb, K, WW, d_a, d_as, d_b, d_bs, kp);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipMemcpyAsync(h_cnt, cnt2, 16, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (n_act > 0) { HIPCHK(hipEventElapsedTime(&f, e1, e2)); round_ms += f; }
            if (n_strag > 0) { HIPCHK(hipEventElapsedTime(&f, c.ev[6], c.ev[7])); round_ms += f; }
            refill_ms += round_ms;
            n_strag2 = h_cnt[2];
            it++;
            if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] pairs [%d,%d) round %d: %d windows, %d stragglers (%d tiles of %d columns) re-filled in %.3f ms; next: %d windows, %d stragglers\n",
                                             p0, p0 + cnt, it, n_act, n_strag, n_strag * tiles_per, FP_TILE, round_ms, h_cnt[nxt], n_strag2);
            if (n_strag2 > 0) HIPCHK(hipMemcpyAsync(d_strag, d_strag + cnt, (size_t)n_strag2 * 4, hipMemcpyDeviceToDevice, st)); // the next round's stragglers
            n_act = h_cnt[nxt];
            n_strag = n_strag2;
            cur = nxt;
        }
        return GNX_OK;
    };

    HIPCHK(hipEventRecord(c.ev[0], stream));
    if ((rc = forward(0, np, stream))) return rc;
    HIPCHK(hipEventRecord(c.ev[1], stream));
    if ((rc = post(0, np, stream, d_cnt, c.ev[4], c.ev[5]))) return rc;
    if ((rc = launch_scan(d_nops, np, d_ops_off, d_carry, stream))) return rc;
    hipLaunchKernelGGL(fp_compact_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, np, d_st, d_stage, d_nops, d_ops_off, d_ops, ops_capacity, d_err, CAP);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c.ev[3], stream));
    int h_misc[16];
    HIPCHK(hipMemcpyAsync(h_misc, c.misc.p, 64, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float tot = 0, fa = 0;
    HIPCHK(hipEventElapsedTime(&tot, c.ev[0], c.ev[3]));
    HIPCHK(hipEventElapsedTime(&fa, c.ev[0], c.ev[1]));
    const double forward_ms = (double)fa;
    if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] forward sweep (%d row block%s, %d slots per lane%s): %d pairs %.3f ms\n", S, S > 1 ? "s in one launch" : "", rows_per_lane, xp ? ", transposed" : "", np, fa);
    if (first) c.timing = gnx_timing{};
    c.timing.fill_ms += forward_ms + refill_ms; c.timing.traceback_ms += std::max(0.0, (double)tot - fa - refill_ms); c.timing.total_ms += tot;
    c.timing.cells += cells; c.timing.n_launches += 1; c.timing.trace_bytes += (int64_t)coff * 8 + (int64_t)roff * 4;
    c.timing.dominant_ms += forward_ms; c.timing.dominant_launches += 1; c.timing.fast_path = 1;
    int64_t total;
    memcpy(&total, reinterpret_cast<char *>(h_misc) + 16, 8);
    if (out_total) *out_total = total;
    const int ef = h_misc[0];
    if (ef & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (ef & 2) { set_err("unexpected traceback%s", ""); return GNX_ETRACE; }
    if (ef & 4) { set_err("CIGAR buffer too small: need %s%lld elements", "", (long long)total); return GNX_ECAPACITY; }
    if (ef & 8) {
        // some CIGARs have more than fp_cap(S) runs: their run counts (and so every offset) are right, their runs were not staged.
        // Align those pairs again on the general path and put the results in place; the batch goes back only if they are many.
        std::vector<int64_t> hn((size_t)np);
        HIPCHK(hipMemcpy(hn.data(), d_nops, (size_t)np * 8, hipMemcpyDeviceToHost));
        std::vector<int> idx;
        std::vector<int64_t> sal, sbl;
        int64_t sub_total = 0;
        for (int p = 0; p < np; p++) if (hn[(size_t)p] > CAP) { idx.push_back(p); sal.push_back(h_alen[p]); sbl.push_back(h_blen[p]); sub_total += hn[(size_t)p]; }
        const int ns = (int)idx.size();
        if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx fp] %d CIGARs have more than %d runs -> those pairs again on the general path\n", ns, CAP);
        if (ns > np / 4) return -1;
        const size_t o_idx = 0, o_as = (size_t)ns * 4 + 64, o_bs = o_as + (size_t)ns * 8, o_sc = o_bs + (size_t)ns * 8, o_off = o_sc + (size_t)ns * 8,
                     o_ops = ((o_off + (size_t)(ns + 1) * 8 + 63) & ~(size_t)63), bytes = o_ops + (size_t)(sub_total + 1) * sizeof(gnx_cigar);
        if ((rc = c.fp_redo.ensure(bytes))) return rc;
        char *rb = reinterpret_cast<char *>(c.fp_redo.p);
        int *d_idx = reinterpret_cast<int *>(rb + o_idx);
        int64_t *s_as = reinterpret_cast<int64_t *>(rb + o_as), *s_bs = reinterpret_cast<int64_t *>(rb + o_bs), *s_sc = reinterpret_cast<int64_t *>(rb + o_sc),
                *s_off = reinterpret_cast<int64_t *>(rb + o_off);
        gnx_cigar *s_ops = reinterpret_cast<gnx_cigar *>(rb + o_ops);
        HIPCHK(hipMemcpyAsync(d_idx, idx.data(), (size_t)ns * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(fp_redo_gather_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, d_idx, ns, d_as, d_bs, s_as, s_bs);
        HIPCHK(hipGetLastError());
        const gnx_timing saved = c.timing;
        int64_t tot2 = 0;
        if (xp) rc = run_device(prm, ns, d_b, s_bs, d_a, s_as, sbl.data(), sal.data(), s_sc, s_ops, sub_total + 1, s_off, &tot2, stream, nullptr, nullptr, 0, nullptr, true, false);
        else rc = run_device(prm, ns, d_a, s_as, d_b, s_bs, sal.data(), sbl.data(), s_sc, s_ops, sub_total + 1, s_off, &tot2, stream, nullptr, nullptr, 0, nullptr, true, false);
        const double redo_ms = c.timing.total_ms;
        c.timing = saved;
        c.timing.traceback_ms += redo_ms; c.timing.total_ms += redo_ms;
        if (rc) return rc;
        // the general path used the shared error / carry words: put this call's back (minus the overflow flag)
        h_misc[0] = ef & ~8;
        HIPCHK(hipMemcpyAsync(c.misc.p, h_misc, 64, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(fp_redo_scatter_kernel, dim3((unsigned)ns), dim3(256), 0, stream, d_idx, ns, s_sc, s_off, s_ops, d_score, d_ops_off, d_ops, d_err);
        HIPCHK(hipGetLastError());
        int e2 = 0;
        HIPCHK(hipMemcpyAsync(&e2, d_err, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (e2 & 2) { set_err("unexpected traceback%s", ""); return GNX_ETRACE; }
    }
    return GNX_OK;
}

int run_device_mega(const gnx_params *prm, const KParams &kp, const TbParams &tp, bool affine, int64_t n_pairs,
                    const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs, const int64_t *h_alen, const int64_t *h_blen,
                    int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off, int64_t *out_total, hipStream_t stream, bool w64);

// tiles per round of the walk farm (farm64.hip.h); 0: the one-workgroup walks
// snapshot spacing of the 64-lane affine sweep when the farm walks (farm64.hip.h): 512 steps (GNX_W64_CK = 128 / 256 / 512); the one-workgroup walks need CKA
int w64_farm_ck() { const char *e = getenv("GNX_W64_CK"); const int v = e ? atoi(e) : 512; return (v == 128 || v == 256 || v == 512) ? v : 512; }
// (affine, pairs of the launch given: the tiles a round of that launch re-fills per pair.  24 for one or two AffineGap pairs -- at 512-step tiles a round is bound by its re-fills, more of them side by
// side pay: 1 Mb x 1 Mb walk 37 -> 30 ms --, 16 otherwise: with four pairs and more, or the constant-gap 224-step tiles, the extra re-fills of a wrong guess cost more than they buy
// (profiles/r6_few_long_pairs_24_tiles_everywhere.jsonl: 16 / 64 ConstGap pairs of 20 kb x 100 kb walked 28 / 31 ms at 24 tiles a round against 22 / 23 at 16, 4 AffineGap pairs of 200 kb 14.0 against 11.8))
int w64_farm_tiles(bool affine = false, int64_t n_pairs = 1 << 20) {
    const char *e = getenv("GNX_W64_FARM");
    if (!e) return (affine && n_pairs <= 2) ? 24 : 16;
    const int v = atoi(e);
    return v <= 0 ? 0 : std::min(v, (int)FARM_MAX);
}

// Rows per lane of the affine 64-lane sweep (affine_long64.hip.h): call f with the compile-time constant of the instantiation
template <typename F>
void w64_rows_dispatch(int rw, F &&f) {
    switch (rw) {
    case 6: f(std::integral_constant<int, 6>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    case 16: f(std::integral_constant<int, 16>{}); break;
    default: f(std::integral_constant<int, R>{}); break;
    }
}
template <typename F>
void w64c_rows_dispatch(int rw, F &&f) { // (the constant-gap twins: const_long64.hip.h)
    if (rw == 4) f(std::integral_constant<int, 4>{});
    else f(std::integral_constant<int, R>{});
}
constexpr int W64_ROWS[4] = {6, 8, R, 16};
constexpr int W64C_ROWS[2] = {4, R};
// One long pair is strips(RW) = n / (64 RW) waves piped through the row buffer, each ~lag steps behind the one above it; a SIMD that holds w of
// them issues w x (5 RW + ~18) instructions per step of the pipeline, and the pipeline moves at the pace of the fullest SIMD.  1 Mb x 1 Mb at RW = 10:
// 1 563 waves on 1 024 SIMDs -- half of the SIMDs hold two, the others wait for them (valu_busy 0.56, profiles/r5_pmc_long_pair.txt); at RW = 8: 1 954,
// two on (nearly) every SIMD, 52 instead of 62 instructions per wave and step.  Per instruction: ~2.6 ns for a wave alone on its SIMD, ~2.05 ns each
// for two, ~1.75 for three, ~1.65 from four on (fitted to profiles/r6_rows_per_lane_after.jsonl: 2 Mb x 2 Mb is fastest at RW = 8 -- 3 907 strips, four per SIMD: 850 ms against 915 at RW = 16).  The model below is that arithmetic;
// GNX_W64_R = 6 / 8 / 10 / 16 overrides it (tests, A/B runs).  `strips_cap` > 0: row panels -- at most that many strips are in flight.
int w64_pick_rows(const Ctx &c, bool affine, int64_t n_pairs, const int64_t *h_alen, const int64_t *h_blen, int64_t step4, int ck, int64_t strips_cap = 0) {
    // (constant gap: 2 RW + ~9 instructions per step; rows per lane 4 / 10, GNX_W64_RC)
    if (const char *e = getenv(affine ? "GNX_W64_R" : "GNX_W64_RC")) { const int v = atoi(e); if (affine) { for (int x : W64_ROWS) if (x == v) return v; } else { for (int x : W64C_ROWS) if (x == v) return v; } }
    // constant gap: the short strips only for the single-pair callers (cmd/globalAlignment: one ConstGap call on two whole sequences).  A tile of the farm is 64 RW rows x 224 steps, so a
    // path crosses n / 256 + m / 224 tiles at RW = 4 instead of n / 640 + m / 224: 16 ... 64 pairs of 20 kb x 100 kb sweep 2 ms faster and walk 10 ms longer (profiles/r6_few_long_pairs.jsonl of the
    // first round-end run: 38 -> 48 ms, 49 -> 60 ms per call), while ONE pair of 150 kb ... 2 Mb sweeps 1 - 30 ms faster and walks 0.5 - 3 ms longer
    if (!affine && n_pairs > 3) return R;
    const double simds = 4.0 * c.n_cu, lag = 110.0;
    double best = 0;
    int best_rw = R;
    for (int rw : {4, 6, 8, (int)R, 16}) {
        if (affine ? rw == 4 : (rw != 4 && rw != R)) continue;
        if ((int64_t)(G64 * rw + G64 + ck + 64) * step4 >= ((int64_t)1 << 28)) continue; // (the keys' spread around a strip's moving base)
        double strips = 0, steps = 0, rows = 0;
        for (int64_t p = 0; p < n_pairs; p++) {
            const double sp = (double)((h_alen[p] + G64 * rw - 1) / (G64 * rw));
            strips += sp; rows += (double)h_alen[p];
            steps = std::max(steps, (double)h_blen[p] + std::min(sp, strips_cap > 0 ? (double)strips_cap : sp) * lag);
        }
        double passes = 1.0;
        if (strips_cap > 0 && strips > (double)strips_cap) { passes = strips / (double)strips_cap; strips = (double)strips_cap; }
        const double w = std::ceil(strips / simds);
        // ns per instruction and SIMD, fitted to profiles/r6_rows_per_lane_after.jsonl (340 kb / 1 Mb / 2 Mb at RW = 6 / 8 / 10 / 16 after the hand-over lost its progress word):
        // a wave alone 2.6, two waves 1.9 + 0.02 RW each (the longer step of a taller strip overlaps worse), three 1.75, four or more 1.65
        const double ns = w <= 1.0 ? 2.6 : (w <= 2.0 ? 1.9 + 0.02 * rw : (w <= 3.0 ? 1.75 : 1.65));
        const double cost = passes * w * (affine ? 5.0 * rw + 12.4 : 2.0 * rw + 9.0) * ns * steps;
        if (best == 0 || cost < best) { best = cost; best_rw = rw; }
    }
    return best_rw;
}

// The walk of the 64-lane snapshot path as rounds of {re-fill the tiles ahead of the walk on many CUs, walk them} (farm64.hip.h).
// d_st: np MegaStates the caller has prepared (zeroed for a whole pair; the panel's state for row panels).  Launches rounds until every
// pair's walk is over (or has left its panel): the first batch sized by the path's length, no host round trip inside a batch.
int run_walk_farm(Ctx &c, bool affine, bool p16, int np, int nt, const PairPlan *dpl, const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
                  const KParams &kp, const TbParams &tp, const void *drb, const int *dsn, const int64_t *dhf, int64_t *d_score, int64_t *dn, const int64_t *d_so,
                  gnx_cigar *d_scr, int *d_err, const long long *dbs, MegaState *d_st, int64_t path_cells, hipStream_t stream) {
    int rc;
    const int ckr = affine ? kp.ckc : 0; // (the affine sweep's snapshot spacing of this call, w64_farm_ck; the constant-gap tiles are CKC64 steps)
    const int rw = affine ? t_w64_r : t_w64_rc; // rows per lane of the sweep that wrote the snapshots (w64_pick_rows)
    size_t tile_dw = 0;
    if (affine) w64_rows_dispatch(rw, [&](auto rwc) { tile_dw = (size_t)FarmGeo<true, decltype(rwc)::value>(ckr).tile_dw(); });
    else w64c_rows_dispatch(rw, [&](auto rwc) { tile_dw = (size_t)FarmGeo<false, decltype(rwc)::value>(0).tile_dw(); });
    const size_t planes_bytes = (size_t)np * 2 * FARM_MAX * tile_dw * 4;
    if ((rc = c.farm.ensure(planes_bytes + (size_t)np * sizeof(FarmCtl)))) return rc;
    unsigned *d_planes = reinterpret_cast<unsigned *>(c.farm.p);
    FarmCtl *d_ctl = reinterpret_cast<FarmCtl *>(reinterpret_cast<char *>(c.farm.p) + planes_bytes);
    const bool pipe = !(getenv("GNX_W64_FARM_PIPE") && getenv("GNX_W64_FARM_PIPE")[0] == '0'); // overlapped rounds (one launch each); 0: {fill, walk} launches
    if (affine) w64_rows_dispatch(rw, [&](auto rwc) { hipLaunchKernelGGL((farm_init_kernel<true, decltype(rwc)::value>), dim3((unsigned)((np + 63) / 64)), dim3(64), 0, stream, dpl, np, tp, d_st, d_ctl, nt, ckr); });
    else w64c_rows_dispatch(rw, [&](auto rwc) { hipLaunchKernelGGL((farm_init_kernel<false, decltype(rwc)::value>), dim3((unsigned)((np + 63) / 64)), dim3(64), 0, stream, dpl, np, tp, d_st, d_ctl, nt, 0); });
    const dim3 gf((unsigned)nt, (unsigned)np), gw((unsigned)np), gr((unsigned)nt + 1, (unsigned)np);
    const int2 *drb2 = reinterpret_cast<const int2 *>(drb);
    const int *drb1 = reinterpret_cast<const int *>(drb);
    auto launch_fill = [&](int par) {
        if (affine) {
            w64_rows_dispatch(rw, [&](auto rwc) {
                constexpr int RW = decltype(rwc)::value;
                if (p16) hipLaunchKernelGGL((al64_farm_fill_kernel<RW, true>), gf, dim3(64), 0, stream, dpl, d_a, d_as, d_b, d_bs, kp, drb2, dsn, d_err, dbs, d_ctl, d_planes, par);
                else hipLaunchKernelGGL((al64_farm_fill_kernel<RW, false>), gf, dim3(64), 0, stream, dpl, d_a, d_as, d_b, d_bs, kp, drb2, dsn, d_err, dbs, d_ctl, d_planes, par);
            });
        } else {
            w64c_rows_dispatch(rw, [&](auto rwc) {
                constexpr int RW = decltype(rwc)::value;
                if (p16) hipLaunchKernelGGL((cl64_farm_fill_kernel<RW, true>), gf, dim3(64), 0, stream, dpl, d_a, d_as, d_b, d_bs, kp, drb1, dsn, d_err, dbs, d_ctl, d_planes, par);
                else hipLaunchKernelGGL((cl64_farm_fill_kernel<RW, false>), gf, dim3(64), 0, stream, dpl, d_a, d_as, d_b, d_bs, kp, drb1, dsn, d_err, dbs, d_ctl, d_planes, par);
            });
        }
    };
    auto launch_round = [&](int par) {
        if (affine) {
            w64_rows_dispatch(rw, [&](auto rwc) {
                constexpr int RW = decltype(rwc)::value;
                if (p16) hipLaunchKernelGGL((al64_farm_round_kernel<RW, true>), gr, dim3(256), 0, stream, dpl, d_a, d_as, d_b, d_bs, kp, tp, drb2, dsn, dhf, d_score, dn, d_so, d_scr, d_err, dbs, d_st, d_ctl, d_planes, nt, par);
                else hipLaunchKernelGGL((al64_farm_round_kernel<RW, false>), gr, dim3(256), 0, stream, dpl, d_a, d_as, d_b, d_bs, kp, tp, drb2, dsn, dhf, d_score, dn, d_so, d_scr, d_err, dbs, d_st, d_ctl, d_planes, nt, par);
            });
        } else {
            w64c_rows_dispatch(rw, [&](auto rwc) {
                constexpr int RW = decltype(rwc)::value;
                if (p16) hipLaunchKernelGGL((cl64_farm_round_kernel<RW, true>), gr, dim3(256), 0, stream, dpl, d_a, d_as, d_b, d_bs, kp, tp, drb1, dsn, dhf, d_score, dn, d_so, d_scr, d_err, dbs, d_st, d_ctl, d_planes, nt, par);
                else hipLaunchKernelGGL((cl64_farm_round_kernel<RW, false>), gr, dim3(256), 0, stream, dpl, d_a, d_as, d_b, d_bs, kp, tp, drb1, dsn, dhf, d_score, dn, d_so, d_scr, d_err, dbs, d_st, d_ctl, d_planes, nt, par);
            });
        }
    };
    int64_t batch = path_cells / ((int64_t)(affine ? ckr * 3 / 4 : 100) * nt) + 8; // (a diagonal crosses ~0.9 of a tile's steps; a round that finds the walk over costs a few us)
    std::vector<FarmCtl> h_ctl((size_t)np);
    int64_t rounds = 0;
    if (pipe) launch_fill(0);
    while (true) {
        for (int64_t r = 0; r < batch; r++) {
            if (pipe) launch_round((int)((rounds + r) & 1));
            else {
                launch_fill(0);
                if (affine) w64_rows_dispatch(rw, [&](auto rwc) { hipLaunchKernelGGL((farm_walk_kernel<true, decltype(rwc)::value>), gw, dim3(256), 0, stream, dpl, tp, dhf, d_score, dn, d_so, d_scr, d_err, d_st, d_ctl, d_planes, nt, ckr); });
                else w64c_rows_dispatch(rw, [&](auto rwc) { hipLaunchKernelGGL((farm_walk_kernel<false, decltype(rwc)::value>), gw, dim3(256), 0, stream, dpl, tp, dhf, d_score, dn, d_so, d_scr, d_err, d_st, d_ctl, d_planes, nt, 0); });
            }
        }
        rounds += batch;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_ctl.data(), d_ctl, (size_t)np * sizeof(FarmCtl), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        bool all = true;
        for (int p = 0; p < np; p++) all = all && h_ctl[(size_t)p].fin;
        if (all) break;
        if (rounds > 3 * path_cells + 64) { set_err("internal: the walk farm does not end%s", ""); return GNX_ETRACE; }
        batch = 16;
    }
    if (getenv("GNX_DEBUG")) for (int p = 0; p < np; p++) fprintf(stderr, "[gnx] walk farm%s: pair %d, %d tiles per round, %d rounds walked %d tiles (%lld launched)\n", pipe ? " (overlapped)" : "", p, nt, h_ctl[(size_t)p].rounds, h_ctl[(size_t)p].hits, (long long)rounds);
    return GNX_OK;
}

// Pairs without a stored direction matrix (const_long.hip.h; affine: affine_long.hip.h): score-only sweep that keeps the strips' bottom
// rows and a snapshot of the wavefront every CKC / CKA steps, then one fused re-fill + walk kernel.  Every n, m >= 1 (validated by the caller).
// Returns GNX_OK, an error, or -1 when the batch should take the general path (a single pair exceeds the workspace).
// rebase: the REBASE instantiations (const_long.hip.h): keys relative to a base every strip moves along -- pairs of any length
// w64 (affine, implies rebase): the whole wave on one pair, strips of 640 rows (affine_long64.hip.h) -- launches of very few pairs
int run_device_clong(const gnx_params *prm, const KParams &kp, const TbParams &tp, bool affine, int64_t n_pairs,
                     const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
                     const int64_t *h_alen, const int64_t *h_blen,
                     int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off, int64_t *out_total, hipStream_t stream, bool rebase, bool w64 = false) {
    Ctx &c = g_ctx;
    int rc;
    if (w64 && !rebase) w64 = false;
    const int rw64 = (w64 && w64_farm_tiles() > 0) ? (affine ? t_w64_r : t_w64_rc) : R; // rows per lane of the 64-lane kernels (the one-workgroup walks: R)
    const int64_t HS = w64 ? (int64_t)G64 * rw64 : H, GS = w64 ? G64 : G; // rows per strip, lanes per pair
    // snapshot spacing of the constant-gap form (const_long.hip.h): the wide tiles only when the walk will have the GPU full of long chains
    int64_t ckc = CKC_SMALL;
    {
        int64_t strips_sum = 0;
        for (int64_t p = 0; p < n_pairs; p++) strips_sum += (h_alen[p] + HS - 1) / HS;
        if (n_pairs > 1536 && strips_sum >= 64 * n_pairs) ckc = CKC;
        if (const char *e = getenv("GNX_CL_CKC")) { const int v = atoi(e); if (v == CKC || v == CKC_SMALL) ckc = v; }
        if (w64) ckc = CKC64;
    }
    // affine snapshot spacing: CKA; the 64-lane sweep under the walk farm: wider (farm64.hip.h) -- a quarter of the snapshots to allocate and to write
    const int64_t ck_aff = (w64 && affine && w64_farm_tiles() > 0) ? t_w64_ck : CKA;
    std::vector<PairPlan> plans((size_t)n_pairs);
    std::vector<int64_t> so((size_t)n_pairs + 1, 0); // staging offsets (runs), chunk-relative; so[chunk end] is unused
    std::vector<int64_t> chunk_begin{0};
    int64_t cells = 0, max_rb = 1, max_sn = 1, max_sc = 1, max_bs = 1;
    {
        int64_t rb = 0, sn = 0, sc = 0, bs = 0;
        int64_t budget = c.ws_limit - c.ws_limit / 16;
        const int64_t rbw = affine ? 8 : 4, ck = affine ? ck_aff : ckc, snw = affine ? (w64 ? al64_snapw(rw64) : AL_SNAPW) : (w64 ? cl64_snapw(rw64) : SNAPW); // row-buffer entry bytes, snapshot spacing / dwords
        auto bytes_of = [&](int64_t rb2, int64_t sn2, int64_t sc2, int64_t bs2) { return rbw * rb2 + 4 * sn2 + (int64_t)sizeof(gnx_cigar) * sc2 + 8 * bs2; };
        auto nq_of = [&](int64_t m) { return rebase ? (((m + GS + 14) & ~(int64_t)15) / ck + 2) : 0; }; // K-step blocks of a strip (REBASE: one int64 base each)
        // One pair that needs more than the workspace limit (a 1 Mb x 1 Mb pair: 50 GB of bottom rows + 43 GB of snapshots) is given what
        // the device has free, plus what these buffers already hold: the limit is there to leave room for other contexts' batches, and such a
        // pair cannot run any other way (the stored matrix would be 750 GB).
        int64_t one_max = 0;
        for (int64_t p = 0; p < n_pairs; p++) {
            const int64_t n = h_alen[p], m = h_blen[p], strips = (n + HS - 1) / HS;
            one_max = std::max(one_max, bytes_of((strips - 1) * (m + 1), (m + GS - 1) / ck * strips * GS * snw, n + m + 2, strips * nq_of(m)));
        }
        if (one_max > budget) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = 0;
            const int64_t avail = (int64_t)fr + (int64_t)(c.rowbuf.cap + c.fp_ckpt.cap + c.tb_scr.cap + c.cl_bases.cap);
            if (one_max + (one_max >> 3) + ((int64_t)1 << 28) > avail) return -1;
            budget = one_max;
            // (the buffers grow to exactly what is asked for: DevBuf::ensure falls back to the plain size when size + 1/8 does not fit)
        }
        for (int64_t p = 0; p < n_pairs; p++) {
            const int64_t n = h_alen[p], m = h_blen[p];
            const int64_t strips = (n + HS - 1) / HS, ncp = (m + GS - 1) / ck;
            const int64_t prb = (strips - 1) * (m + 1), psn = ncp * strips * GS * snw, psc = n + m + 2, pbs = strips * nq_of(m);
            if (bytes_of(prb, psn, psc, pbs) > budget) return -1;
            if (bytes_of(rb + prb, sn + psn, sc + psc, bs + pbs) > budget) { // new chunk, 4-aligned so that waves stay whole
                int64_t cb = p & ~(int64_t)3;
                if (cb <= chunk_begin.back()) cb = p;
                rb = sn = sc = bs = 0;
                for (int64_t q2 = cb; q2 < p; q2++) {
                    PairPlan &pq = plans[(size_t)q2];
                    pq.rowbuf_off = rb; pq.ckpt_off = sn; so[(size_t)q2] = sc; pq.hcol_off = q2 - cb; pq.src = (int32_t)(q2 - cb); pq.rowi_off = bs;
                    rb += (int64_t)(pq.strips - 1) * (pq.m + 1); sn += (int64_t)((pq.m + GS - 1) / ck) * pq.strips * GS * snw; sc += (int64_t)pq.n + pq.m + 2; bs += (int64_t)pq.strips * pq.s_pitch;
                }
                chunk_begin.push_back(cb);
            }
            PairPlan &pl = plans[(size_t)p];
            pl.n = (int32_t)n; pl.m = (int32_t)m; pl.words = 0; pl.strips = (int32_t)strips;
            pl.trace_off = 0; pl.dcol_off = 0; pl.col_off = 0; pl.rowi_off = bs; pl.s_off = 0; pl.s_pitch = nq_of(m);
            pl.rowbuf_off = rb; pl.ckpt_off = sn; so[(size_t)p] = sc;
            pl.hcol_off = p - chunk_begin.back(); pl.src = (int32_t)(p - chunk_begin.back());
            rb += prb; sn += psn; sc += psc; bs += pbs;
            max_rb = std::max(max_rb, rb); max_sn = std::max(max_sn, sn); max_sc = std::max(max_sc, sc); max_bs = std::max(max_bs, bs);
            cells += n * m;
        }
        chunk_begin.push_back(n_pairs);
    }
    int64_t max_np = 1;
    for (size_t ch = 0; ch + 1 < chunk_begin.size(); ch++) max_np = std::max(max_np, chunk_begin[ch + 1] - chunk_begin[ch]);
    if ((rc = c.rowbuf.ensure((size_t)max_rb * (affine ? 8 : 4)))) return rc;
    if ((rc = c.fp_ckpt.ensure((size_t)max_sn * 4))) return rc;
    if ((rc = c.tb_scr.ensure((size_t)max_sc * sizeof(gnx_cigar)))) return rc;
    if ((rc = c.tb_scr_off.ensure(((size_t)n_pairs + 1) * 8))) return rc;
    if ((rc = c.hcol.ensure((size_t)max_np * 8))) return rc;
    if (rebase && (rc = c.cl_bases.ensure((size_t)max_bs * 8))) return rc;
    c.fpc_ptr = nullptr;
    if ((rc = c.plans.ensure((size_t)n_pairs * sizeof(PairPlan)))) return rc;
    if ((rc = c.nops.ensure((size_t)n_pairs * 8))) return rc;
    if ((rc = c.misc.ensure(64))) return rc;
    int *d_err = reinterpret_cast<int *>(c.misc.p);
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, stream));
    HIPCHK(hipMemcpyAsync(c.plans.p, plans.data(), (size_t)n_pairs * sizeof(PairPlan), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(c.tb_scr_off.p, so.data(), ((size_t)n_pairs + 1) * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream)); // plans / so are locals
    // int16 profile when every entry 4*(s - 2g) + 1 fits (half the LDS reads per step); GNX_CL_P16=0/1 overrides for A/B runs
    bool p16 = true;
    for (int x = 0; x < 25; x++) { const int64_t v = 4 * (prm->scores[x] - 2 * (affine ? prm->gap_extend : prm->gap_open)) + 1; if (v > 32767 || v < -32768) p16 = false; }
    if (getenv("GNX_CL_P16")) p16 = p16 && atoi(getenv("GNX_CL_P16")) != 0;
    double fill_ms = 0, tb_ms = 0;
    int64_t trace_bytes = 0;
    HIPCHK(hipEventRecord(c.ev[0], stream));
    const size_t nchunks = chunk_begin.size() - 1;
    for (size_t ch = 0; ch < nchunks; ch++) {
        const int64_t b = chunk_begin[ch], e = chunk_begin[ch + 1];
        const int np = (int)(e - b);
        if (np <= 0) continue;
        const PairPlan *dpl = reinterpret_cast<const PairPlan *>(c.plans.p) + b;
        int *drb = reinterpret_cast<int *>(c.rowbuf.p), *dsn = reinterpret_cast<int *>(c.fp_ckpt.p);
        int64_t *dhf = reinterpret_cast<int64_t *>(c.hcol.p);
        long long *dbs = rebase ? reinterpret_cast<long long *>(c.cl_bases.p) : nullptr;
        int64_t *dn = reinterpret_cast<int64_t *>(c.nops.p) + b;
        const int64_t *d_so = reinterpret_cast<const int64_t *>(c.tb_scr_off.p) + b;
        gnx_cigar *d_scr = reinterpret_cast<gnx_cigar *>(c.tb_scr.p);
        bool multi = false;
        int64_t m_maxc = 0;
        for (int64_t q2 = b; q2 < e; q2++) { if (plans[(size_t)q2].strips > 1) multi = true; m_maxc = std::max<int64_t>(m_maxc, plans[(size_t)q2].m); }
        int64_t n_blocks = (np + 3) / 4;
        const bool piped = w64 || (multi && n_blocks < 3072 && m_maxc >= 8 * RB_PUB && !no_pipe());
        // constant gap, int16 profile: several strips per workgroup, rows handed over through LDS (cl_sweep_wg_kernel; GNX_CL_WG=0: one strip per workgroup)
        constexpr int CLW_NW = 4;
        const bool wg = piped && !affine && p16 && !rebase && !(getenv("GNX_CL_WG") && atoi(getenv("GNX_CL_WG")) == 0);
        const int per_item = wg ? CLW_NW : 1;
        const int2 *d_smap = nullptr;
        int *d_sprog = nullptr;
        if (piped) {
            std::vector<int2> smap; // (group, item): an item = one strip, or CLW_NW consecutive strips; w64: (pair, strip)
            if (w64) for (int q3 = 0; q3 < np; q3++) for (int st2 = 0; st2 < plans[(size_t)(b + q3)].strips; st2++) smap.push_back(make_int2(q3, st2));
            for (int gq = 0; !w64 && gq < (np + 3) / 4; gq++) {
                int smax = 0;
                for (int q3 = 0; q3 < 4 && gq * 4 + q3 < np; q3++) smax = std::max(smax, (int)plans[(size_t)(b + gq * 4 + q3)].strips);
                for (int st2 = 0; st2 < (smax + per_item - 1) / per_item; st2++) smap.push_back(make_int2(gq, st2));
            }
            n_blocks = (int64_t)smap.size();
            if (n_blocks > 0x7fffffff) { set_err("too many strips in one chunk%s", ""); return GNX_ENOMEM; }
            if ((rc = c.strip_map.ensure((size_t)std::max<int64_t>(n_blocks, 1) * 16 + 8))) return rc; // map, progress words, claim words, test switch
            d_smap = reinterpret_cast<const int2 *>(c.strip_map.p);
            d_sprog = reinterpret_cast<int *>(reinterpret_cast<char *>(c.strip_map.p) + (size_t)std::max<int64_t>(n_blocks, 1) * 8);
            HIPCHK(hipMemcpyAsync(c.strip_map.p, smap.data(), (size_t)n_blocks * 8, hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemsetAsync(d_sprog, 0, (size_t)std::max<int64_t>(n_blocks, 1) * 8 + 8, stream));
            if ((rc = claim_test_switch(d_sprog + 2 * n_blocks, stream))) return rc;
            HIPCHK(hipStreamSynchronize(stream)); // smap is a local
        }
        HIPCHK(hipEventRecord(c.ev[1], stream));
        // How often a piped strip publishes its bottom row.  A strip cannot start before the one above it has published its first columns,
        // so a launch ramps up over (strips per pair) x (publish interval + look-ahead) columns: with the GPU full of waves (C5: 64 000
        // workgroups on 5 120 wave slots) a publish every block instead of every fourth hides its cost (the wait for the stores'
        // acknowledgement) behind the other waves and shortens the ramp -- 2048 pairs 437 -> 400 ms; a launch that does not fill the GPU
        // pays for every publish with its own latency (64 pairs of C5: 17.4 -> 19.2 ms) and keeps the coarse interval.
        KParams kps = kp, kpa = kp;
        kps.ckc = (int)ckc;
        kpa.ckc = (int)ck_aff; // (al64_sweep_kernel, the farm's re-fills)
        if (w64) { g_last_w64_r = rw64; g_last_w64_ck = affine ? ck_aff : ckc; }
        kps.rb_pub = n_blocks * per_item >= (int64_t)40 * c.n_cu ? 16 : RB_PUB; // (twice the wave slots of the piped sweep: 20 per CU)
        if (const char *e = getenv("GNX_CL_PUB")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64 || v == 128) kps.rb_pub = v; }
        int2 *drb2 = reinterpret_cast<int2 *>(c.rowbuf.p);
        const dim3 gridS((unsigned)n_blocks);
#define GNX_AL_SWEEP(P_, RBS_) hipLaunchKernelGGL((al_sweep_kernel<P_, RBS_>), gridS, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, drb2, dsn, dhf, d_err, d_smap, d_sprog, dbs)
#define GNX_CL_SWEEP(P_, RBS_) hipLaunchKernelGGL((cl_sweep_kernel<P_, RBS_>), gridS, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kps, drb, dsn, dhf, d_err, d_smap, d_sprog, dbs)
#define GNX_CL_FLAT(P_, RBS_) hipLaunchKernelGGL((cl_sweep_flat_kernel<P_, RBS_>), gridS, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kps, drb, dsn, dhf, d_err, dbs)
        if (rebase && !w64) HIPCHK(hipMemsetAsync(dbs, 0, (size_t)max_bs * 8, stream));
        if (w64) { // the 64-lane sweeps hand rows over without a progress word (affine_long64.hip.h, W64_SENT): row buffer and bases start as "not written yet", block 0 of every strip's bases as 0
            int64_t rb_used = 0, bs_used = 0;
            for (int64_t q2 = b; q2 < e; q2++) { const PairPlan &pq = plans[(size_t)q2]; rb_used = std::max(rb_used, pq.rowbuf_off + (int64_t)(pq.strips - 1) * (pq.m + 1)); bs_used = std::max(bs_used, pq.rowi_off + (int64_t)pq.strips * pq.s_pitch); }
            if (rb_used > 0) HIPCHK(hipMemsetAsync(c.rowbuf.p, 0x80, (size_t)rb_used * (affine ? 8 : 4), stream));
            if (bs_used > 0) HIPCHK(hipMemsetAsync(dbs, 0x80, (size_t)bs_used * 8, stream));
            for (int64_t q2 = b; q2 < e; q2++) { const PairPlan &pq = plans[(size_t)q2]; HIPCHK(hipMemset2DAsync(dbs + pq.rowi_off, (size_t)pq.s_pitch * 8, 0, 8, (size_t)pq.strips, stream)); }
        }
        if (w64 && !affine) {
            w64c_rows_dispatch(rw64, [&](auto rwc) {
                constexpr int RW = decltype(rwc)::value;
                if (p16) hipLaunchKernelGGL((cl64_sweep_kernel<RW, true>), gridS, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kps, drb, dsn, dhf, d_err, d_smap, d_sprog, dbs);
                else hipLaunchKernelGGL((cl64_sweep_kernel<RW, false>), gridS, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kps, drb, dsn, dhf, d_err, d_smap, d_sprog, dbs);
            });
        } else if (w64) {
            w64_rows_dispatch(rw64, [&](auto rwc) {
                constexpr int RW = decltype(rwc)::value;
                if (p16) hipLaunchKernelGGL((al64_sweep_kernel<RW, true>), gridS, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kpa, drb2, dsn, dhf, d_err, d_smap, d_sprog, dbs);
                else hipLaunchKernelGGL((al64_sweep_kernel<RW, false>), gridS, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kpa, drb2, dsn, dhf, d_err, d_smap, d_sprog, dbs);
            });
        } else if (affine) { if (rebase) { if (p16) GNX_AL_SWEEP(true, true); else GNX_AL_SWEEP(false, true); } else { if (p16) GNX_AL_SWEEP(true, false); else GNX_AL_SWEEP(false, false); } }
        else if (wg) {
            if (GNX_CLW_SENT) { // the item boundaries' rows start as "not written yet" (const_long_wg.hip.h)
                int smax_all = 0;
                for (int64_t q2 = b; q2 < e; q2++) smax_all = std::max(smax_all, (int)plans[(size_t)q2].strips);
                const dim3 gfs((unsigned)std::max((smax_all + CLW_NW - 1) / CLW_NW, 1), (unsigned)np);
                hipLaunchKernelGGL(clw_fill_sentinel_kernel, gfs, dim3(256), 0, stream, dpl, np, CLW_NW, drb);
            }
            hipLaunchKernelGGL(cl_sweep_wg_kernel<CLW_NW>, gridS, dim3(64 * CLW_NW), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kps, drb, dsn, dhf, d_err, d_smap, d_sprog);
        }
        else if (piped) { if (rebase) { if (p16) GNX_CL_SWEEP(true, true); else GNX_CL_SWEEP(false, true); } else { if (p16) GNX_CL_SWEEP(true, false); else GNX_CL_SWEEP(false, false); } }
        else { if (rebase) { if (p16) GNX_CL_FLAT(true, true); else GNX_CL_FLAT(false, true); } else { if (p16) GNX_CL_FLAT(true, false); else GNX_CL_FLAT(false, false); } }
#undef GNX_AL_SWEEP
#undef GNX_CL_SWEEP
#undef GNX_CL_FLAT
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c.ev[2], stream));
        const dim3 gridW((unsigned)((np + 3) / 4));
#define GNX_AL_WALK(P_, RBS_) hipLaunchKernelGGL((al_walk_kernel<P_, RBS_>), gridW, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb2, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr)
        const int farm_nt = w64 ? w64_farm_tiles(affine, np) : 0;
        if (farm_nt > 0) { // the walk as rounds of tiles re-filled ahead of it on many CUs (farm64.hip.h)
            int64_t path = 0;
            for (int64_t p = b; p < e; p++) path = std::max(path, (int64_t)plans[(size_t)p].n + plans[(size_t)p].m);
            if ((rc = c.mega_state.ensure(256 + (size_t)np * sizeof(MegaState)))) return rc;
            MegaState *d_fst = reinterpret_cast<MegaState *>(reinterpret_cast<char *>(c.mega_state.p) + 256);
            HIPCHK(hipMemsetAsync(d_fst, 0, (size_t)np * sizeof(MegaState), stream));
            if ((rc = run_walk_farm(c, affine, p16, (int)np, farm_nt, dpl, d_a, d_as + b, d_b, d_bs + b, kpa, tp, affine ? (const void *)drb2 : (const void *)drb, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, d_fst, path / 2, stream))) return rc;
        } else if (w64 && !affine) { // one pair per workgroup
            const dim3 gw((unsigned)np);
            if (w64_two_waves()) {
                if (p16) hipLaunchKernelGGL((cl64_walk2_kernel<true>), gw, dim3(128), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr);
                else hipLaunchKernelGGL((cl64_walk2_kernel<false>), gw, dim3(128), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr);
            }
            else if (p16) hipLaunchKernelGGL((cl64_walk_kernel<true>), gw, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr);
            else hipLaunchKernelGGL((cl64_walk_kernel<false>), gw, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr);
        } else if (w64) {
            const dim3 gw((unsigned)np);
            if (w64_two_waves()) { // (the tile to the left re-filled by a second wave while the first re-fills the walk's: GNX_W64_SPEC=0 switches it off)
                if (p16) hipLaunchKernelGGL((al64_walk2_kernel<true>), gw, dim3(128), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb2, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr);
                else hipLaunchKernelGGL((al64_walk2_kernel<false>), gw, dim3(128), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb2, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr);
            }
            else if (p16) hipLaunchKernelGGL((al64_walk_kernel<true>), gw, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb2, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr);
            else hipLaunchKernelGGL((al64_walk_kernel<false>), gw, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb2, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr);
        } else if (affine) { if (rebase) { if (p16) GNX_AL_WALK(true, true); else GNX_AL_WALK(false, true); } else { if (p16) GNX_AL_WALK(true, false); else GNX_AL_WALK(false, false); } }
#undef GNX_AL_WALK
        else if (rebase) { // (one pair per workgroup, plain walk)
            const dim3 gw((unsigned)np);
#define GNX_CL_WALKR(P_, CK_) hipLaunchKernelGGL((cl_walk_kernel<P_, 1, CK_, true>), gw, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, dbs, (MegaState *)nullptr)
            if (ckc == CKC) { if (p16) GNX_CL_WALKR(true, CKC); else GNX_CL_WALKR(false, CKC); }
            else { if (p16) GNX_CL_WALKR(true, CKC_SMALL); else GNX_CL_WALKR(false, CKC_SMALL); }
#undef GNX_CL_WALKR
        } else {
            // pairs per walk workgroup: GNX_CL_WALK_NP = 1 / 2 / 4 (default 1, see cl_walk_kernel)
            const char *npe = getenv("GNX_CL_WALK_NP");
            const int wnp = (npe && (npe[0] == '2' || npe[0] == '4')) ? npe[0] - '0' : 1;
            const dim3 gw((unsigned)((np + wnp - 1) / wnp));
            // (padding the workgroups' LDS so that a launch smaller than the GPU spreads over all CUs changes nothing: the dispatcher already does)
            auto launch_walk = [&](auto kern) {
                hipLaunchKernelGGL(kern, gw, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err, (const long long *)nullptr, (MegaState *)nullptr);
            };
            const bool wide = ckc == CKC;
            // speculative re-fills of the next tiles by the wave's other lane groups (cl_walk_spec_kernel): GNX_CL_WALK_SPEC = 0 / 3 / 4 tiles per round
            // Three tiles per round while every pair of the launch is resident at once (33.6 KB of LDS: four workgroups per CU) and the walk is a
            // latency chain: 1024 pairs of C5 37.9 -> 28.6 ms.  With more pairs than that the plain walk's eight workgroups per CU win
            // (2048 pairs, 448-step tiles: 42 ms plain, 111 ms speculative), and four tiles per round (three workgroups per CU) never pay.
            int spec = (np <= 4 * c.n_cu && !wide) ? 3 : 0;
            if (const char *se = getenv("GNX_CL_WALK_SPEC")) { const int v = atoi(se); spec = (v == 2 || v == 3 || v == 4) ? v : 0; }
            if (npe) spec = 0; // (an explicit GNX_CL_WALK_NP asks for the plain walk)
            const dim3 gs((unsigned)np);
            auto launch_spec = [&](auto kern) {
                hipLaunchKernelGGL(kern, gs, dim3(64), 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, tp, drb, dsn, dhf, d_score + b, dn, d_so, d_scr, d_err);
            };
            if (spec == 2) { // two tiles per round: 24.6 / 42.5 KB of LDS, six / three workgroups per CU
                if (p16) { if (wide) launch_spec(cl_walk_spec_kernel<true, CKC, 2>); else launch_spec(cl_walk_spec_kernel<true, CKC_SMALL, 2>); }
                else { if (wide) launch_spec(cl_walk_spec_kernel<false, CKC, 2>); else launch_spec(cl_walk_spec_kernel<false, CKC_SMALL, 2>); }
            } else if (spec && p16) {
                if (spec == 3) { if (wide) launch_spec(cl_walk_spec_kernel<true, CKC, 3>); else launch_spec(cl_walk_spec_kernel<true, CKC_SMALL, 3>); }
                else { if (wide) launch_spec(cl_walk_spec_kernel<true, CKC, 4>); else launch_spec(cl_walk_spec_kernel<true, CKC_SMALL, 4>); }
            } else if (spec) {
                if (spec == 3) { if (wide) launch_spec(cl_walk_spec_kernel<false, CKC, 3>); else launch_spec(cl_walk_spec_kernel<false, CKC_SMALL, 3>); }
                else { if (wide) launch_spec(cl_walk_spec_kernel<false, CKC, 4>); else launch_spec(cl_walk_spec_kernel<false, CKC_SMALL, 4>); }
            } else if (p16) {
                if (wnp == 1) { if (wide) launch_walk(cl_walk_kernel<true, 1, CKC>); else launch_walk(cl_walk_kernel<true, 1, CKC_SMALL>); }
                else if (wnp == 2) { if (wide) launch_walk(cl_walk_kernel<true, 2, CKC>); else launch_walk(cl_walk_kernel<true, 2, CKC_SMALL>); }
                else { if (wide) launch_walk(cl_walk_kernel<true, 4, CKC>); else launch_walk(cl_walk_kernel<true, 4, CKC_SMALL>); }
            } else {
                if (wnp == 1) { if (wide) launch_walk(cl_walk_kernel<false, 1, CKC>); else launch_walk(cl_walk_kernel<false, 1, CKC_SMALL>); }
                else if (wnp == 2) { if (wide) launch_walk(cl_walk_kernel<false, 2, CKC>); else launch_walk(cl_walk_kernel<false, 2, CKC_SMALL>); }
                else { if (wide) launch_walk(cl_walk_kernel<false, 4, CKC>); else launch_walk(cl_walk_kernel<false, 4, CKC_SMALL>); }
            }
        }
        HIPCHK(hipGetLastError());
        if ((rc = launch_scan(dn, np, d_ops_off + b, d_carry, stream))) return rc;
        hipLaunchKernelGGL(reverse_runs_kernel, dim3((unsigned)np), dim3(256), 0, stream, dpl, np, d_scr, d_so, dn, d_ops_off + b, d_ops, ops_capacity, d_err);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c.ev[3], stream));
        HIPCHK(hipEventSynchronize(c.ev[3])); // the chunks reuse the workspace
        float f1 = 0, f2 = 0;
        HIPCHK(hipEventElapsedTime(&f1, c.ev[1], c.ev[2]));
        HIPCHK(hipEventElapsedTime(&f2, c.ev[2], c.ev[3]));
        fill_ms += f1; tb_ms += f2;
        for (int64_t p = b; p < e; p++) {
            const PairPlan &pl = plans[(size_t)p];
            trace_bytes += (affine ? 8 : 4) * (int64_t)(pl.strips - 1) * (pl.m + 1) + 4 * (int64_t)((pl.m + GS - 1) / (affine ? ck_aff : ckc)) * pl.strips * GS * (affine ? (w64 ? al64_snapw(rw64) : AL_SNAPW) : (w64 ? cl64_snapw(rw64) : SNAPW));
        }
    }
    HIPCHK(hipEventRecord(c.ev[2], stream));
    int h_misc[16];
    HIPCHK(hipMemcpyAsync(h_misc, c.misc.p, 64, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float tot = 0;
    HIPCHK(hipEventElapsedTime(&tot, c.ev[0], c.ev[2]));
    c.timing.fill_ms = fill_ms; c.timing.traceback_ms = tb_ms; c.timing.total_ms = tot;
    c.timing.cells = cells; c.timing.n_launches = (int64_t)nchunks; c.timing.trace_bytes = trace_bytes;
    c.timing.dominant_ms = fill_ms; c.timing.dominant_launches = (int64_t)nchunks; c.timing.fast_path = w64 ? 6 : 2;
    int64_t total;
    memcpy(&total, reinterpret_cast<char *>(h_misc) + 16, 8);
    if (out_total) *out_total = total;
    const int ef = h_misc[0];
    if (ef & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (ef & 16) {
        if (t_no_pipe) { set_err("a strip waited more than 5 s for the strip above it%s", ""); return GNX_EDEVICE; }
        if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx] a pipelined strip timed out: the call runs again with sequential strips\n");
        t_no_pipe = true;
        rc = run_device_clong(prm, kp, tp, affine, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, rebase, false);
        t_no_pipe = false;
        return rc;
    }
    if (ef & 2) { set_err("unexpected traceback%s", ""); return GNX_ETRACE; }
    if (ef & 4) { set_err("CIGAR buffer too small: need %s%lld elements", "", (long long)total); return GNX_ECAPACITY; }
    return GNX_OK;
}

// Pairs whose snapshot working set (bottom rows of all strips + snapshots: 20 B per column and strip for AffineGap) exceeds the workspace:
// ROW PANELS.  Forward, panel by panel: the REBASE sweep of const_long / affine_long on as many strips as fit, the panel's top boundary = the
// bottom row of the panel above (kept with the bases it is relative to: 8 B per column and panel).  The kernels need no change for that: a
// panel below the first gets a stand-in strip 0 whose bottom row and bases the host copies into the row buffer, whose workgroup finds its
// item claimed and its progress word at "done".  Backward, from the last panel: (re-)sweep the panel up to the column the walk has reached,
// walk until it leaves the panel upwards (MegaState), continue in the panel above.  Cost: the forward sweep + about half of it again.
// One pair at a time (such pairs fill the device on their own).  5 Mb x 5 Mb: 79 panels of 400 strips, 40 GB.
int run_device_mega(const gnx_params *prm, const KParams &kp, const TbParams &tp, bool affine, int64_t n_pairs,
                    const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
                    const int64_t *h_alen, const int64_t *h_blen,
                    int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off, int64_t *out_total, hipStream_t stream, bool w64) {
    Ctx &c = g_ctx;
    int rc;
    const int rw64 = (w64 && w64_farm_tiles() > 0) ? (affine ? t_w64_r : t_w64_rc) : R; // rows per lane of the 64-lane kernels (the one-workgroup walks: R)
    const int64_t HS = w64 ? (int64_t)G64 * rw64 : H, GS = w64 ? G64 : G; // rows per strip, lanes per pair (w64: affine_long64.hip.h / const_long64.hip.h)
    const int np = (int)n_pairs;
    // Snapshot spacing of the affine 64-lane sweep in row panels: what a backward panel holds per strip is its bottom row (8 B per column) + its snapshots
    // (4 (2 RW + 2) / K B per column and lane), and the strips that fit the budget are the waves that sweep: 5 Mb x 5 Mb at RW = 16, K = 512: 814 strips a
    // panel (0.8 waves per SIMD), K = 2 048: 1 690.  The farm's re-fills run beside the walk, so a tile of 2 048 steps costs the walk nothing per cell of path
    // (its planes: 1.6 MB, 64 of them).  The widest power of two the keys' spread around a strip's base admits, GNX_W64_CK_MEGA = 128 .. 2048 (default 2 048).
    int64_t ck_mega = t_w64_ck;
    if (affine && w64 && w64_farm_tiles() > 0) {
        const int64_t step4 = 4 * (max_abs_pen(prm, false) + 2 * llabs((long long)prm->gap_open) + 2 * llabs((long long)prm->gap_extend));
        int want = 2048;
        if (const char *e = getenv("GNX_W64_CK_MEGA")) { const int v = atoi(e); if (v == 128 || v == 256 || v == 512 || v == 1024 || v == 2048) want = v; }
        for (int v = want; v >= CKA; v >>= 1) if ((int64_t)((int64_t)G64 * rw64 + G64 + v + 64) * step4 < ((int64_t)1 << 28)) { ck_mega = v; break; }
    }
    const int64_t ck = affine ? ((w64 && w64_farm_tiles() > 0) ? ck_mega : CKA) : CKC_SMALL, snw = affine ? (w64 ? al64_snapw(rw64) : AL_SNAPW) : (w64 ? cl64_snapw(rw64) : SNAPW), rbw = affine ? 8 : 4;
    bool p16 = true;
    for (int x = 0; x < 25; x++) { const int64_t v = 4 * (prm->scores[x] - 2 * (affine ? prm->gap_extend : prm->gap_open)) + 1; if (v > 32767 || v < -32768) p16 = false; }
    std::vector<int64_t> so((size_t)np + 1, 0), h_start((size_t)np * 2);
    int64_t m_hi = 1;
    for (int64_t p = 0; p < n_pairs; p++) {
        if (h_alen[p] < 1 || h_blen[p] < 1) return -1;
        so[(size_t)p + 1] = so[(size_t)p] + h_alen[p] + h_blen[p] + 2;
        m_hi = std::max(m_hi, h_blen[p]);
    }
    const int64_t nq_hi = ((m_hi + GS + 14) & ~(int64_t)15) / ck + 2;
    const int64_t strip_fwd_hi = (m_hi + 1) * rbw + nq_hi * 8;                                   // forward pass: bottom row + bases
    const int64_t strip_bwd_hi = strip_fwd_hi + ((m_hi + GS - 1) / ck) * GS * snw * 4;                 // backward pass: + snapshots
    const int64_t scr_b = so[(size_t)np] * (int64_t)sizeof(gnx_cigar);
    // such a pair cannot run any other way: it is given what the device has free (plus what these buffers hold already), not the workspace
    // limit that leaves room for other contexts' batches -- strips in flight are what keeps the device busy (400 strips: a quarter of it)
    int64_t budget = c.ws_limit - c.ws_limit / 8;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            // (the snapshot path's buffers of earlier calls -- up to what the device has -- give way to a pair that needs row panels)
            if (c.rowbuf.cap + c.fp_ckpt.cap + c.cl_bases.cap > ((size_t)1 << 30)) { c.rowbuf.release(); c.fp_ckpt.release(); c.cl_bases.release(); (void)hipMemGetInfo(&fr, &tot); }
            const int64_t avail = (int64_t)fr + (int64_t)(c.mega_arena.cap + c.tb_scr.cap + c.mega_rows.cap);
            // half of what is there: a fresh device allocation costs ~27 ms per GB (MI355X, measured: tools/memprobe.py), which a one-call process --
            // cmd/cigarToBed -- pays in full; panels of three quarters of the device sweep ~10 % faster and allocate 2 s longer
            budget = std::max(budget, avail / 2);
        }
    }
    budget -= scr_b;
    // the saved panel boundaries (mega_rows: one row + its bases per backward panel) come out of the same budget: size the panels, count them, size again
    int64_t Sb = 0, Sf = 0, rows_hi = 0;
    {
        int64_t n_hi = 1;
        for (int64_t p = 0; p < n_pairs; p++) n_hi = std::max(n_hi, h_alen[p]);
        const int64_t strips_hi = (n_hi + HS - 1) / HS, top_hi = (m_hi + 1) * rbw + nq_hi * 8;
        int64_t tops = 16 * top_hi;
        for (int pass = 0; pass < 3; pass++) {
            const int64_t b2 = budget - tops - tops / 8;
            Sb = b2 / (strip_bwd_hi + strip_bwd_hi / 8) - 3; Sf = b2 / (strip_fwd_hi + strip_fwd_hi / 8) - 3; // strips per backward / forward panel (/ 8: DevBuf::ensure's slack)
            if (Sb < 2) break;
            const int64_t need = ((strips_hi + Sb - 1) / Sb) * top_hi;
            if (need <= tops) break;
            tops = need;
        }
        rows_hi = tops;
    }
    (void)rows_hi;
    if (const char *e = getenv("GNX_MEGA_STRIPS")) { Sb = atoll(e); Sf = 2 * Sb; } // (tests: panels of a few strips, forward panels of two backward ones)
    if (Sb < 2) { set_err("a single strip of pair %s%lld does not fit the workspace", "", 0); return GNX_ENOMEM; }
    const bool forced_panels = getenv("GNX_MEGA_STRIPS") != nullptr;
    const int64_t Sb_max = Sb, Sf_max = Sf;
    if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx] row panels: budget %.1f GB, %lld strips per backward panel (%.1f MB each), %lld per forward panel (%.1f MB each)\n", budget / 1e9, (long long)Sb, strip_bwd_hi / 1e6, (long long)Sf, strip_fwd_hi / 1e6);
    if ((rc = c.tb_scr.ensure((size_t)scr_b))) return rc;
    if ((rc = c.tb_scr_off.ensure(((size_t)np + 1) * 8))) return rc;
    if ((rc = c.nops.ensure((size_t)np * 8))) return rc;
    if ((rc = c.misc.ensure(64))) return rc;
    if ((rc = c.hcol.ensure(64))) return rc;
    if ((rc = c.mega_state.ensure(256))) return rc;
    c.fpc_ptr = nullptr;
    if ((rc = c.plans.ensure(sizeof(PairPlan)))) return rc;
    int *d_err = reinterpret_cast<int *>(c.misc.p);
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    int64_t *dn = reinterpret_cast<int64_t *>(c.nops.p);
    int64_t *dhf = reinterpret_cast<int64_t *>(c.hcol.p);
    gnx_cigar *d_scr = reinterpret_cast<gnx_cigar *>(c.tb_scr.p);
    const int64_t *d_so = reinterpret_cast<const int64_t *>(c.tb_scr_off.p);
    MegaState *d_st = reinterpret_cast<MegaState *>(c.mega_state.p);
    int64_t *d_starts = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.mega_state.p) + 128); // {alpha start, beta start} of the panel
    HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, stream));
    HIPCHK(hipMemcpyAsync(c.tb_scr_off.p, so.data(), ((size_t)np + 1) * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(h_start.data(), d_as, (size_t)np * 8, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(h_start.data() + np, d_bs, (size_t)np * 8, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    double fill_ms = 0, tb_ms = 0;
    int64_t cells = 0, launches = 0, ws_bytes = 0;
    for (int64_t p = 0; p < n_pairs; p++) {
        const int64_t n = h_alen[p], m = h_blen[p];
        // backward panels of equal size (a remainder panel of a few strips would sweep the whole width at a fraction of the device); forward
        // panels of whole backward panels -- their top rows are what a forward panel saves -- up to 9 strips per CU (5 Mb x 5 Mb: 13 per CU sweep
        // 0.5 s faster and take 1.4 s longer to allocate in a process's first call)
        // ... unless whole rounds of the SIMDs fit: a launch is paced by the SIMD that holds most strips (one wave each), so 2 048 strips cost what 1 628 do -- panels of
        // k x 1 024 strips and one remainder panel (5 Mb x 5 Mb at 1 024-row strips: 2 048 + 2 048 + 787 instead of 3 x 1 628), when the remainder is not a sliver
        const int64_t total_strips = (n + HS - 1) / HS, simds = 4 * (int64_t)c.n_cu;
        int64_t n_bwd = (total_strips + Sb_max - 1) / Sb_max, Sb_pick = (total_strips + n_bwd - 1) / n_bwd;
        if (w64 && Sb_max >= simds && total_strips > Sb_max) {
            const int64_t Sq = Sb_max / simds * simds, rem = total_strips % Sq;
            if (rem == 0 || rem >= simds / 4) { Sb_pick = Sq; n_bwd = (total_strips + Sq - 1) / Sq; }
        }
        const int64_t Sb = forced_panels ? Sb_max : Sb_pick;
        if (forced_panels) n_bwd = (total_strips + Sb_max - 1) / Sb_max;
        const int64_t Sf = forced_panels ? Sf_max : std::max(Sb, std::min<int64_t>(Sf_max, std::max<int64_t>(Sb, 9 * c.n_cu)) / Sb * Sb);
        const int64_t nq = ((m + GS + 14) & ~(int64_t)15) / ck + 2; // K-step blocks of a strip: the pitch of the bases, the same in every panel
        const int64_t top_b = (m + 1) * rbw + nq * 8;           // one saved boundary: the row + its bases
        if ((rc = c.mega_rows.ensure((size_t)(n_bwd * top_b)))) return rc;
        char *tops = reinterpret_cast<char *>(c.mega_rows.p); // tops[b]: the row above backward panel b (b >= 1)
        cells += n * m;
        PairPlan pl;
        char *ar_rows = nullptr, *ar_bases = nullptr, *ar_snap = nullptr; // this launch's slices of the arena
        // sweep of the strips [s0, s0 + cnt) of the pair over columns 1 .. mcols; forward: no snapshots, the bottom rows of every Sb-th strip saved
        auto sweep_rows = [&](int64_t s0, int64_t cnt, int64_t mcols, bool forward) -> int {
            const int64_t r0 = s0 * HS, rows = std::min(n, (s0 + cnt) * HS) - r0, virt = s0 > 0 ? HS : 0;
            const bool last = s0 + cnt >= total_strips;
            const int64_t local = cnt + (s0 > 0 ? 1 : 0), planned = local + (last ? 0 : 1); // (+1: the last real strip hands its row down)
            pl.n = (int32_t)(rows + virt); pl.m = (int32_t)mcols; pl.words = 0; pl.strips = (int32_t)planned;
            pl.trace_off = 0; pl.dcol_off = 0; pl.col_off = 0; pl.s_off = 0; pl.rowbuf_off = 0; pl.ckpt_off = 0;
            pl.hcol_off = (forward && last) ? 0 : 1; pl.src = 0; pl.rowi_off = 0; pl.s_pitch = nq;
            const int64_t rb_e = (planned - 1) * (mcols + 1), sn_e = forward ? 0 : ((mcols + GS - 1) / ck) * planned * GS * snw, bs_e = planned * nq;
            int r2;
            if (getenv("GNX_DEBUG")) {
                size_t fr = 0, tot = 0;
                (void)hipMemGetInfo(&fr, &tot);
                fprintf(stderr, "[gnx] row panels: %s sweep of strips %lld + %lld: rows %.1f GB (held %.1f), snapshots %.1f GB (held %.1f), free %.1f GB\n", forward ? "forward" : "backward", (long long)s0, (long long)cnt,
                        rb_e * rbw / 1e9, c.mega_arena.cap / 1e9, sn_e * 4 / 1e9, c.mega_arena.cap / 1e9, fr / 1e9);
            }
            // rows | bases | snapshots of this launch, carved from the call's ONE arena (sized below for the biggest launch of either pass): no buffer is freed
            // and allocated again between the passes (round 5 released the forward pass's rows to make room for the snapshots in EVERY call, and waited for
            // the driver to clear what it got back: seconds per call at 5 Mb x 5 Mb)
            const size_t o_bases = ((size_t)std::max<int64_t>(rb_e, 1) * rbw + 255) & ~(size_t)255, o_snap = (o_bases + (size_t)bs_e * 8 + 255) & ~(size_t)255;
            if (o_snap + (size_t)std::max<int64_t>(sn_e, 1) * 4 > c.mega_arena.cap) { set_err("internal: a row panel needs %s%lld bytes, more than its arena", "", (long long)(o_snap + sn_e * 4)); return GNX_ENOMEM; }
            ar_rows = reinterpret_cast<char *>(c.mega_arena.p); ar_bases = ar_rows + o_bases; ar_snap = ar_rows + o_snap;
            if ((r2 = c.strip_map.ensure((size_t)local * 16 + 8))) return r2;
            ws_bytes = std::max(ws_bytes, rb_e * rbw + sn_e * 4 + bs_e * 8);
            std::vector<int2> smap((size_t)local);
            for (int64_t st = 0; st < local; st++) smap[(size_t)st] = make_int2(0, (int)st);
            int *d_sprog = reinterpret_cast<int *>(reinterpret_cast<char *>(c.strip_map.p) + (size_t)local * 8);
            const int64_t starts[2] = {h_start[(size_t)p] + r0 - virt, h_start[(size_t)(np + p)]};
            HIPCHK(hipMemcpyAsync(c.plans.p, &pl, sizeof(PairPlan), hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(c.strip_map.p, smap.data(), (size_t)local * 8, hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(d_starts, starts, 16, hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemsetAsync(d_sprog, 0, (size_t)local * 8 + 8, stream));
            if (w64) { // (the hand-over without a progress word: everything starts as "not written yet", block 0 of every strip's bases as 0; the stand-in strip's row and bases come below)
                if (rb_e > 0) HIPCHK(hipMemsetAsync(ar_rows, 0x80, (size_t)rb_e * rbw, stream));
                HIPCHK(hipMemsetAsync(ar_bases, 0x80, (size_t)bs_e * 8, stream));
                HIPCHK(hipMemset2DAsync(ar_bases, (size_t)nq * 8, 0, 8, (size_t)planned, stream));
            } else HIPCHK(hipMemsetAsync(ar_bases, 0, (size_t)bs_e * 8, stream));
            if (s0 > 0) { // the stand-in strip 0: done and claimed; its bottom row and bases = what the strip above handed down
                // (no conversion: the keys are V' = V - e (i + j) with the PAIR's row i in every panel -- the recurrences never look at i, and
                // column 0 of a global alignment is the same constant in every row; only the final un-rebasing used the panel's row count, below)
                const int pre[1] = {0x7fffffff}, one[1] = {1};
                const char *top = tops + (s0 / Sb) * top_b;
                HIPCHK(hipMemcpyAsync(d_sprog, pre, 4, hipMemcpyHostToDevice, stream));
                HIPCHK(hipMemcpyAsync(d_sprog + local, one, 4, hipMemcpyHostToDevice, stream));
                HIPCHK(hipMemcpyAsync(ar_rows, top, (size_t)(mcols + 1) * rbw, hipMemcpyDeviceToDevice, stream));
                HIPCHK(hipMemcpyAsync(ar_bases, top + (m + 1) * rbw, (size_t)nq * 8, hipMemcpyDeviceToDevice, stream));
            }
            HIPCHK(hipStreamSynchronize(stream)); // (pl, smap, starts are locals)
            const PairPlan *dpl = reinterpret_cast<const PairPlan *>(c.plans.p);
            const int2 *d_smap = reinterpret_cast<const int2 *>(c.strip_map.p);
            long long *dbs = reinterpret_cast<long long *>(ar_bases);
            int *dsn = forward ? nullptr : reinterpret_cast<int *>(ar_snap); // (null: the sweep keeps no snapshots)
            KParams kps = kp;
            kps.ckc = (int)ck; kps.rb_pub = RB_PUB;
            if (w64) { g_last_w64_r = rw64; g_last_w64_ck = ck; }
            const dim3 gridS((unsigned)local);
            HIPCHK(hipEventRecord(c.ev[1], stream));
            if (w64 && !affine) {
                int *drb = reinterpret_cast<int *>(ar_rows);
                w64c_rows_dispatch(rw64, [&](auto rwc) {
                    constexpr int RW = decltype(rwc)::value;
                    if (p16) hipLaunchKernelGGL((cl64_sweep_kernel<RW, true>), gridS, dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kps, drb, dsn, dhf, d_err, d_smap, d_sprog, dbs);
                    else hipLaunchKernelGGL((cl64_sweep_kernel<RW, false>), gridS, dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kps, drb, dsn, dhf, d_err, d_smap, d_sprog, dbs);
                });
            } else if (w64) {
                int2 *drb2 = reinterpret_cast<int2 *>(ar_rows);
                w64_rows_dispatch(rw64, [&](auto rwc) {
                    constexpr int RW = decltype(rwc)::value;
                    if (p16) hipLaunchKernelGGL((al64_sweep_kernel<RW, true>), gridS, dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kps, drb2, dsn, dhf, d_err, d_smap, d_sprog, dbs);
                    else hipLaunchKernelGGL((al64_sweep_kernel<RW, false>), gridS, dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kps, drb2, dsn, dhf, d_err, d_smap, d_sprog, dbs);
                });
            } else if (affine) {
                int2 *drb2 = reinterpret_cast<int2 *>(ar_rows);
                if (p16) hipLaunchKernelGGL((al_sweep_kernel<true, true>), gridS, dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kps, drb2, dsn, dhf, d_err, d_smap, d_sprog, dbs);
                else hipLaunchKernelGGL((al_sweep_kernel<false, true>), gridS, dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kps, drb2, dsn, dhf, d_err, d_smap, d_sprog, dbs);
            } else {
                int *drb = reinterpret_cast<int *>(ar_rows);
                if (p16) hipLaunchKernelGGL((cl_sweep_kernel<true, true>), gridS, dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kps, drb, dsn, dhf, d_err, d_smap, d_sprog, dbs);
                else hipLaunchKernelGGL((cl_sweep_kernel<false, true>), gridS, dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kps, drb, dsn, dhf, d_err, d_smap, d_sprog, dbs);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c.ev[2], stream));
            launches++;
            if (forward) { // the bottom rows (and bases) of the strips that end a backward panel: the top boundaries of the panels below them
                for (int64_t e = s0 + Sb; e <= s0 + cnt && e < total_strips; e += Sb) {
                    const int64_t sl = (e - 1 - s0) + (s0 > 0 ? 1 : 0); // that strip's slot in this launch
                    char *top = tops + (e / Sb) * top_b;
                    HIPCHK(hipMemcpyAsync(top, reinterpret_cast<char *>(ar_rows) + (size_t)(sl * (m + 1)) * rbw, (size_t)(m + 1) * rbw, hipMemcpyDeviceToDevice, stream));
                    HIPCHK(hipMemcpyAsync(top + (m + 1) * rbw, reinterpret_cast<char *>(ar_bases) + (size_t)(sl * nq) * 8, (size_t)nq * 8, hipMemcpyDeviceToDevice, stream));
                }
            }
            HIPCHK(hipEventSynchronize(c.ev[2]));
            float f = 0;
            HIPCHK(hipEventElapsedTime(&f, c.ev[1], c.ev[2]));
            fill_ms += f;
            if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx] row panels: %s sweep of %lld strips x %lld columns: %.1f ms (%.3g cells/s)\n", forward ? "forward" : "backward", (long long)cnt, (long long)mcols, f, (double)rows * (double)mcols / (f * 1e-3));
            return GNX_OK;
        };
        { // the arena: the biggest launch of the forward pass (rows + bases of Sf + 2 strips) or of the backward pass (+ snapshots, Sb + 2 strips)
            const int64_t lf = std::min(Sf, total_strips) + 2, lb = std::min(Sb, total_strips) + 2;
            const int64_t need_f = lf * (m + 1) * rbw + lf * nq * 8 + 1024, need_b = lb * (m + 1) * rbw + lb * nq * 8 + ((m + GS - 1) / ck) * lb * GS * snw * 4 + 1024;
            if ((rc = c.mega_arena.ensure((size_t)std::max(need_f, need_b)))) return rc;
        }
        for (int64_t s0 = 0; s0 < total_strips; s0 += Sf) if ((rc = sweep_rows(s0, std::min(Sf, total_strips - s0), m, true))) return rc; // forward
        int64_t n_local_last = 0;
        { const int64_t s0l = ((total_strips - 1) / Sf) * Sf; n_local_last = (n - s0l * HS) + (s0l > 0 ? HS : 0); } // rows of the launch that wrote h(n, m)
        HIPCHK(hipMemcpyAsync(d_score + p, dhf, 8, hipMemcpyDeviceToDevice, stream)); // h(n, m) of the last forward launch ...
        { // ... un-rebased by the kernel with that launch's row count: the rows above it are still owed
            const long long owed = (long long)(affine ? prm->gap_extend : prm->gap_open) * (n - n_local_last);
            if (owed) hipLaunchKernelGGL(add_i64_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<long long *>(d_score + p), (int64_t)1, owed);
        }
        MegaState st;
        memset(&st, 0, sizeof(st));
        int64_t jcur = m;
        for (int64_t k = n_bwd - 1; k >= 0; k--) {
            if ((rc = sweep_rows(k * Sb, std::min(Sb, total_strips - k * Sb), jcur, false))) return rc;
            const int64_t r0 = k * Sb * HS, virt = k > 0 ? HS : 0;
            st.virt = (int32_t)virt; st.row_off = r0 - virt;
            if (st.resume) { st.wi = pl.n; st.wj = (int32_t)jcur; }
            HIPCHK(hipMemcpyAsync(d_st, &st, sizeof(st), hipMemcpyHostToDevice, stream));
            const PairPlan *dpl = reinterpret_cast<const PairPlan *>(c.plans.p);
            const long long *dbs = reinterpret_cast<const long long *>(ar_bases);
            const int *dsn = reinterpret_cast<const int *>(ar_snap);
            int64_t *d_tmp_score = dhf + 2; // (the walk writes hfin[pl.hcol_off] here when it ends: not the pair's score, see above)
            HIPCHK(hipEventRecord(c.ev[1], stream));
            const int farm_nt = w64 ? w64_farm_tiles(affine, 1) : 0;
            if (farm_nt > 0) {
                KParams kpf = kp;
                kpf.ckc = (int)ck;
                if ((rc = run_walk_farm(c, affine, p16, 1, farm_nt, dpl, d_a, d_starts, d_b, d_starts + 1, kpf, tp, ar_rows, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st,
                                        std::min<int64_t>((int64_t)pl.n + jcur, 2 * (int64_t)pl.n) / 2, stream))) return rc;
            } else if (w64 && !affine) {
                const int *drb = reinterpret_cast<const int *>(ar_rows);
                if (w64_two_waves()) {
                    if (p16) hipLaunchKernelGGL((cl64_walk2_kernel<true>), dim3(1), dim3(128), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
                    else hipLaunchKernelGGL((cl64_walk2_kernel<false>), dim3(1), dim3(128), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
                }
                else if (p16) hipLaunchKernelGGL((cl64_walk_kernel<true>), dim3(1), dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
                else hipLaunchKernelGGL((cl64_walk_kernel<false>), dim3(1), dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
            } else if (w64) {
                const int2 *drb2 = reinterpret_cast<const int2 *>(ar_rows);
                if (w64_two_waves()) {
                    if (p16) hipLaunchKernelGGL((al64_walk2_kernel<true>), dim3(1), dim3(128), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb2, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
                    else hipLaunchKernelGGL((al64_walk2_kernel<false>), dim3(1), dim3(128), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb2, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
                }
                else if (p16) hipLaunchKernelGGL((al64_walk_kernel<true>), dim3(1), dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb2, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
                else hipLaunchKernelGGL((al64_walk_kernel<false>), dim3(1), dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb2, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
            } else if (affine) {
                const int2 *drb2 = reinterpret_cast<const int2 *>(ar_rows);
                if (p16) hipLaunchKernelGGL((al_walk_kernel<true, true>), dim3(1), dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb2, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
                else hipLaunchKernelGGL((al_walk_kernel<false, true>), dim3(1), dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb2, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
            } else {
                const int *drb = reinterpret_cast<const int *>(ar_rows);
                if (p16) hipLaunchKernelGGL((cl_walk_kernel<true, 1, CKC_SMALL, true>), dim3(1), dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
                else hipLaunchKernelGGL((cl_walk_kernel<false, 1, CKC_SMALL, true>), dim3(1), dim3(64), 0, stream, dpl, 1, d_a, d_starts, d_b, d_starts + 1, kp, tp, drb, dsn, dhf, d_tmp_score, dn + p, d_so + p, d_scr, d_err, dbs, d_st);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c.ev[2], stream));
            HIPCHK(hipMemcpyAsync(&st, d_st, sizeof(st), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            float f = 0;
            HIPCHK(hipEventElapsedTime(&f, c.ev[1], c.ev[2]));
            tb_ms += f;
            if (st.done) break;
            if (k == 0) { set_err("internal: the walk left the first panel upwards%s", ""); return GNX_ETRACE; }
            st.resume = 1;
            jcur = st.wj;
        }
    }
    if ((rc = launch_scan(dn, np, d_ops_off, d_carry, stream))) return rc;
    {
        // reverse_runs_kernel reads plans[p].src: one plan per pair whose src is the pair's index
        std::vector<PairPlan> pls((size_t)np);
        memset(pls.data(), 0, (size_t)np * sizeof(PairPlan));
        for (int p = 0; p < np; p++) pls[(size_t)p].src = p;
        if ((rc = c.plans.ensure((size_t)np * sizeof(PairPlan)))) return rc;
        HIPCHK(hipMemcpyAsync(c.plans.p, pls.data(), (size_t)np * sizeof(PairPlan), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(reverse_runs_kernel, dim3((unsigned)np), dim3(256), 0, stream, reinterpret_cast<const PairPlan *>(c.plans.p), np, d_scr, d_so, dn, d_ops_off, d_ops, ops_capacity, d_err);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream)); // (pls is a local)
    }
    int h_misc[16];
    HIPCHK(hipMemcpyAsync(h_misc, c.misc.p, 64, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    c.timing.fill_ms = fill_ms; c.timing.traceback_ms = tb_ms; c.timing.total_ms = fill_ms + tb_ms;
    c.timing.cells = cells; c.timing.n_launches = launches; c.timing.trace_bytes = ws_bytes;
    c.timing.dominant_ms = fill_ms; c.timing.dominant_launches = launches; c.timing.fast_path = 5;
    int64_t total;
    memcpy(&total, reinterpret_cast<char *>(h_misc) + 16, 8);
    if (out_total) *out_total = total;
    const int ef = h_misc[0];
    if (ef & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (ef & 16) { set_err("a strip waited more than 5 s for the strip above it%s", ""); return GNX_EDEVICE; }
    if (ef & 2) { set_err("unexpected traceback%s", ""); return GNX_ETRACE; }
    if (ef & 4) { set_err("CIGAR buffer too small: need %s%lld elements", "", (long long)total); return GNX_ECAPACITY; }
    return GNX_OK;
}

// The latency geometry (lat_fill.hip.h): one pair per wave, 64 lanes x 2 rows, the strips of a pair as piped workgroups; one wave per pair
// walks the stored matrix.  For launches of few pairs (the single align.AffineGap / ConstGap call and small loops of them).
// Returns GNX_OK, an error, or -1 when the batch should take the general path (empty sequences, workspace).
int run_device_lat(const gnx_params *prm, const KParams &kp, const TbParams &tp, bool affine, bool local, int64_t n_pairs,
                   const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
                   const int64_t *h_alen, const int64_t *h_blen,
                   int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off, int64_t *out_total, hipStream_t stream,
                   const int *d_smat = nullptr, const int64_t *h_soff = nullptr, bool smat16 = false, bool wide = false) {
    // d_smat / h_soff: explicit score matrices (the chunk / multiple-alignment variants): lat_fill_kernel<.., SCORED>
    // wide: int64 keys, literal recurrences (lat_wide.hip.h): pairs no int32 kernel can hold
    Ctx &c = g_ctx;
    int rc;
    const int np = (int)n_pairs;
    const int Q = affine ? LQA : LQC;
    int64_t n_blocks = 0;
    // (the scored int64 kernel takes empty sequences along -- AffineGap_highMem semantics: no strip to fill, traceback_kernel's closed forms -- it has no other route)
    const bool empties_ok = wide && d_smat != nullptr;
    for (int64_t p = 0; p < n_pairs; p++) {
        if (h_alen[p] < 1 || h_blen[p] < 1) { if (!empties_ok) return -1; continue; }
        n_blocks += (h_alen[p] + LH - 1) / LH;
    }
    if (n_blocks > 0x3fffffff) return -1;
    // plans, staging offsets and the strip map are built in ONE pinned host block and go up in one copy (a single pair per call is the
    // common case here: every separate copy or synchronisation is ~10 us of a ~200 us call); the claim words follow them on the device
    const size_t o_so = (size_t)np * sizeof(PairPlan), o_map = o_so + ((size_t)np + 1) * 8, o_claims = o_map + (size_t)n_blocks * 8, meta_b = o_claims + (size_t)n_blocks * 4 + 8;
    if ((rc = c.h_plans.ensure(o_claims))) return rc;
    char *hm = reinterpret_cast<char *>(c.h_plans.p);
    PairPlan *plans = reinterpret_cast<PairPlan *>(hm);
    int64_t *so = reinterpret_cast<int64_t *>(hm + o_so);
    int2 *smap = reinterpret_cast<int2 *>(hm + o_map);
    int64_t toff = 0, hoff = 0, roff = 0, doff = 0, cells = 0, nb = 0;
    const int64_t rbw = (wide && affine) ? 2 : 1; // row-buffer entries per column (8 bytes each: int2 {dn, h}, or int64 keys)
    so[0] = 0;
    for (int64_t p = 0; p < n_pairs; p++) {
        const int64_t n = h_alen[p], m = h_blen[p];
        PairPlan &pl = plans[(size_t)p];
        pl.n = (int32_t)n; pl.m = (int32_t)m;
        pl.strips = (n < 1 || m < 1) ? 0 : (int32_t)((n + LH - 1) / LH);
        pl.words = (int32_t)((m + (LG - 1) + 15) / 16);
        pl.trace_off = toff; pl.hcol_off = hoff; pl.rowbuf_off = roff; pl.dcol_off = doff;
        pl.src = (int32_t)p; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0;
        if (h_soff) { pl.s_off = h_soff[p]; pl.s_pitch = (int64_t)((n + H - 1) / H) * H; } // (the matrices are laid out for the general path's 160-row strips)
        toff += (int64_t)pl.strips * pl.words * Q * LG; hoff += n; roff += (int64_t)std::max(pl.strips - 1, 0) * (m + 1) * rbw; doff += (int64_t)pl.strips * LG;
        so[(size_t)p + 1] = so[(size_t)p] + n + m + 2;
        cells += n * m;
        for (int st = 0; st < pl.strips; st++) smap[(size_t)nb++] = make_int2((int)p, st);
    }
    const int64_t n_scr = so[(size_t)np];
    const size_t need = (size_t)toff * 16 + (size_t)hoff * 4 + (size_t)roff * 8 + (size_t)doff * 4 + (size_t)n_scr * sizeof(gnx_cigar);
    if ((int64_t)need > c.ws_limit - c.ws_limit / 8) return -1;
    if ((rc = c.trace.ensure((size_t)std::max<int64_t>(toff, 1) * 16))) return rc;
    if ((rc = c.hcol.ensure((size_t)std::max<int64_t>(hoff, 1) * 4))) return rc;
    if ((rc = c.rowbuf.ensure((size_t)std::max<int64_t>(roff, 1) * 8))) return rc;
    if ((rc = c.dcol.ensure((size_t)std::max<int64_t>(doff, 1) * 4))) return rc;
    if ((rc = c.tb_scr.ensure((size_t)n_scr * sizeof(gnx_cigar)))) return rc;
    c.fpc_ptr = nullptr;
    if ((rc = c.plans.ensure(meta_b))) return rc;
    if ((rc = c.nops.ensure((size_t)np * 8))) return rc;
    if ((rc = c.misc.ensure(64))) return rc;
    if (wide && (rc = c.mx_score.ensure((size_t)np * 8))) return rc; // (the int64 scores of the wide kernel)
    char *dm = reinterpret_cast<char *>(c.plans.p);
    int *d_err = reinterpret_cast<int *>(c.misc.p);
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    const int2 *d_smap = reinterpret_cast<const int2 *>(dm + o_map);
    int *d_claims = reinterpret_cast<int *>(dm + o_claims);
    HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, stream));
    HIPCHK(hipMemcpyAsync(dm, hm, o_claims, hipMemcpyHostToDevice, stream)); // (pinned: no synchronisation needed; every call ends with one)
    HIPCHK(hipMemsetAsync(d_claims, 0, (size_t)n_blocks * 4 + 4, stream));
    if (roff > 0) HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c.rowbuf.p), LAT_SENT, (size_t)roff * 2, stream)); // "not written yet"
    if ((rc = claim_test_switch(d_claims + n_blocks, stream))) return rc;
    const PairPlan *dpl = reinterpret_cast<const PairPlan *>(c.plans.p);
    uint4 *dtrace = reinterpret_cast<uint4 *>(c.trace.p);
    int *dh = reinterpret_cast<int *>(c.hcol.p);
    int2 *drb = reinterpret_cast<int2 *>(c.rowbuf.p);
    unsigned *ddc = reinterpret_cast<unsigned *>(c.dcol.p);
    int64_t *dn = reinterpret_cast<int64_t *>(c.nops.p);
    gnx_cigar *d_scr = reinterpret_cast<gnx_cigar *>(c.tb_scr.p);
    const int64_t *d_so = reinterpret_cast<const int64_t *>(dm + o_so);
    HIPCHK(hipEventRecord(c.ev[0], stream));
    const dim3 gridF((unsigned)n_blocks), gridP((unsigned)np), blk(64);
#define GNX_LAT(A_, L_) hipLaunchKernelGGL((lat_fill_kernel<A_, L_>), gridF, blk, 0, stream, dpl, np, d_a, d_as, d_b, d_bs, kp, dtrace, dh, drb, ddc, d_err, d_smap, d_claims)
    if (wide) {
        int64_t *d_s64 = reinterpret_cast<int64_t *>(c.mx_score.p);
        k64 *drw = reinterpret_cast<k64 *>(c.rowbuf.p);
        const long long o4w = 4 * (long long)prm->gap_open, e4w = affine ? 4 * (long long)prm->gap_extend : 0;
        const long long d00w = local ? 0 : o4w, ecolw = local ? 0 : e4w;
#define GNX_WIDE(A_, L_) hipLaunchKernelGGL((lat_wide_kernel<A_, L_>), gridF, blk, 0, stream, dpl, np, d_a, d_as, d_b, d_bs, kp, o4w, e4w, d00w, ecolw, dtrace, dh, d_s64, drw, ddc, d_err, d_smap, d_claims)
        if (d_smat) hipLaunchKernelGGL((lat_wide_kernel<true, false, true>), gridF, blk, 0, stream, dpl, np, d_a, d_as, d_b, d_bs, kp, o4w, e4w, d00w, ecolw, dtrace, dh, d_s64, drw, ddc, d_err, d_smap, d_claims, d_smat);
        else if (affine) { if (local) GNX_WIDE(true, true); else GNX_WIDE(true, false); }
        else GNX_WIDE(false, false);
#undef GNX_WIDE
    }
    else if (d_smat) {
        if (smat16) hipLaunchKernelGGL((lat_fill_kernel<true, false, true, true>), gridF, blk, 0, stream, dpl, np, d_a, d_as, d_b, d_bs, kp, dtrace, dh, drb, ddc, d_err, d_smap, d_claims, d_smat);
        else hipLaunchKernelGGL((lat_fill_kernel<true, false, true, false>), gridF, blk, 0, stream, dpl, np, d_a, d_as, d_b, d_bs, kp, dtrace, dh, drb, ddc, d_err, d_smap, d_claims, d_smat);
    }
    else if (affine) { if (local) GNX_LAT(true, true); else GNX_LAT(true, false); }
    else GNX_LAT(false, false);
#undef GNX_LAT
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c.ev[1], stream));
    if (affine) hipLaunchKernelGGL((traceback_kernel<true, false, true, true, LG, LR>), gridP, blk, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score, dn, d_so, d_scr, (int64_t)0, d_err);
    else hipLaunchKernelGGL((traceback_kernel<false, false, true, true, LG, LR>), gridP, blk, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score, dn, d_so, d_scr, (int64_t)0, d_err);
    HIPCHK(hipGetLastError());
    if (wide) hipLaunchKernelGGL(wide_scores_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, stream, dpl, reinterpret_cast<const int64_t *>(c.mx_score.p), d_score, np);
    if ((rc = launch_scan(dn, np, d_ops_off, d_carry, stream))) return rc;
    hipLaunchKernelGGL(reverse_runs_kernel, gridP, dim3(256), 0, stream, dpl, np, d_scr, d_so, dn, d_ops_off, d_ops, ops_capacity, d_err);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c.ev[2], stream));
    int h_misc[16];
    HIPCHK(hipMemcpyAsync(h_misc, c.misc.p, 64, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float f1 = 0, f2 = 0;
    HIPCHK(hipEventElapsedTime(&f1, c.ev[0], c.ev[1]));
    HIPCHK(hipEventElapsedTime(&f2, c.ev[1], c.ev[2]));
    c.timing.fill_ms = f1; c.timing.traceback_ms = f2; c.timing.total_ms = f1 + f2;
    c.timing.cells = cells; c.timing.n_launches = 1; c.timing.trace_bytes = toff * 16;
    c.timing.dominant_ms = f1; c.timing.dominant_launches = 1; c.timing.fast_path = wide ? 4 : 3;
    int64_t total;
    memcpy(&total, reinterpret_cast<char *>(h_misc) + 16, 8);
    if (out_total) *out_total = total;
    const int ef = h_misc[0];
    if (ef & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (ef & 16) { if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx] a strip of the latency geometry timed out: the call runs again on the general path\n"); return -2; }
    if (ef & 2) { set_err("unexpected traceback%s", ""); return GNX_ETRACE; }
    if (ef & 4) { set_err("CIGAR buffer too small: need %s%lld elements", "", (long long)total); return GNX_ECAPACITY; }
    return GNX_OK;
}

// The device flow shared by all entry points.  All pointers are device pointers except h_*.
int run_device(const gnx_params *prm, int64_t n_pairs,
               const uint8_t *d_a, const int64_t *d_as, const uint8_t *d_b, const int64_t *d_bs,
               const int64_t *h_alen, const int64_t *h_blen,
               int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off,
               int64_t *out_total, hipStream_t stream, const int *d_smat = nullptr, const int64_t *h_soff = nullptr,
               int gsw = 0, int2 *d_endpos = nullptr, bool no_fast_path = false, bool smat16 = false) {
    // gsw: 1 / 2 = LeftDynamicAln / RightDynamicAln of the graph aligner (constant-gap kernels with GSW = 1 / 2 and their own
    // traceback; d_endpos receives the (i, j) the reference returns); prm->mode must be GNX_CONST_GAP_HIGHMEM
    // d_smat / h_soff: explicit per-cell score matrices (SCORED kernels, N1 variants); the sequences are then unused
    Ctx &c = g_ctx;
    KParams kp; TbParams tp; bool affine, local, lowmem;
    int rc = check_params(prm, kp, tp, affine, local, lowmem);
    if (rc) return rc;
    if (c.beta_packed) { // beta windows of the packed resident reference (by_offset flow); d_b is then unused
        if (local || gsw || d_smat) { set_err("internal: packed reference on a path that reads beta as bytes%s", ""); return GNX_EINVAL; }
        kp.b2 = reinterpret_cast<const unsigned *>(c.ref.p); kp.bflag = reinterpret_cast<const unsigned long long *>(c.ref_flag.p);
        kp.brank = reinterpret_cast<const unsigned *>(c.ref_rank.p); kp.bexc = reinterpret_cast<const unsigned long long *>(c.ref_exc.p);
    }
    if (d_smat && (!affine || local || lowmem)) { set_err("scored mode needs AffineGap_highMem semantics%s", ""); return GNX_EINVAL; }
    if (gsw && (affine || lowmem || !d_endpos || prm->gap_open > 0)) { set_err("gsw extension needs ConstGap_highMem parameters with gapPen <= 0%s", ""); return GNX_EINVAL; }
    if (n_pairs < 0 || n_pairs > 0x7ffffff0) { set_err("bad n_pairs%s", ""); return GNX_EINVAL; }
    c.timing = gnx_timing{};
    if (n_pairs == 0) {
        int64_t z = 0;
        HIPCHK(hipMemcpyAsync(d_ops_off, &z, 8, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (out_total) *out_total = 0;
        return GNX_OK;
    }
    const int64_t maxpen = max_abs_pen(prm, affine);
    int64_t first_oor = -1; // first pair beyond the static key range
    // ---- validation, the same for every path below ----
    for (int64_t p = 0; p < n_pairs; p++) {
        const int64_t n = h_alen[p], m = h_blen[p];
        if (n < 0 || m < 0 || n > 0x3fffffff || m > 0x3fffffff) { set_err("bad sequence length at pair %s%lld", "", (long long)p); return GNX_EINVAL; }
        if (lowmem && prm->checkersize_i != prm->checkersize_j && n > prm->checkersize_i) {
            // the reference indexes its saved columns with checkersize_j where checkersize_i is meant
            // (affineGap.go:252-254, constGap.go:207): undefined for non-square tiles once n > checkersize_i
            set_err("non-square checkerboards with n > checkersize_i are undefined in the reference (pair %s%lld)", "", (long long)p); return GNX_EINVAL;
        }
        if (lowmem && (n < 1 || m < 1)) { set_err("empty sequence at pair %s%lld: the reference never terminates on it", "", (long long)p); return GNX_EEMPTY; }
        // beyond the STATIC int32 range of the kernels' keys (4 * score, absolute): such pairs take the snapshot path with moving bases
        // (REBASE, const_long.hip.h), which has no length limit -- see the clong block below; the reference is int64 throughout (align/align.go:8)
        if ((n + m + 2) * std::max<int64_t>(maxpen, 1) >= ((int64_t)1 << 27) && first_oor < 0) first_oor = p;
    }
    // ---- the chunk / multiple-alignment variants with a pair beyond the int32 range (run_host_scored built plain 4 * s matrices): the int64 kernel ----
    if (!gsw && d_smat && t_scored_wide) {
        rc = run_device_lat(prm, kp, tp, affine, local, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, d_smat, h_soff, false, true);
        if (rc >= 0) return rc;
        if (rc == -2) { set_err("a strip of the int64 kernel waited more than 5 s for the strip above it%s", ""); return GNX_EDEVICE; }
        set_err("the direction matrices of a chunk / multiple-alignment call beyond the int32 range do not fit the workspace%s", ""); return GNX_ENOMEM;
    }
    // ---- GNX_WIDE=2 (tests): everything through the int64 kernel (lat_wide.hip.h) ----
    if (!gsw && !d_smat && !c.beta_packed) {
        const char *we = getenv("GNX_WIDE");
        if (we && we[0] == '2') {
            rc = run_device_lat(prm, kp, tp, affine, local, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, nullptr, nullptr, false, true);
            if (rc >= 0) return rc; // (-1: an empty sequence / the workspace -> the ordinary routes; -2: its bug trap)
        }
    }
    // ---- latency geometry: few pairs (lat_fill.hip.h).  A lone wave is paced by its own instruction stream, so a launch that cannot fill
    // the device runs one pair per wave on 64 lanes x 2 rows instead of four pairs per wave on 16 x 10: at most LAT_MAX 128-row strips in
    // all (~3.5 waves per SIMD: profiles/r5_lat_crossover.jsonl).  GNX_LAT=0 / 2: never / whenever the mode allows; the switches that force another route for the tests
    // (GNX_FASTPATH=0 / 2, GNX_CLONG=2, GNX_FP_SMALL=1, GNX_NO_HFORM) switch the automatic choice off.
    if (!gsw && first_oor < 0 && !c.beta_packed && !t_no_lat && (!affine || prm->gap_open <= 0) && !(d_smat && getenv("GNX_NO_HFORM"))) {
        const char *le = getenv("GNX_LAT");
        const char *fpe = getenv("GNX_FASTPATH"), *cle = getenv("GNX_CLONG"), *fse = getenv("GNX_FP_SMALL");
        const bool forced = le && le[0] == '2';
        bool use = !(le && le[0] == '0');
        if (!forced && ((fpe && (fpe[0] == '2' || fpe[0] == '0')) || (cle && cle[0] == '2') || (fse && fse[0] == '1') || getenv("GNX_NO_HFORM") || no_fast_path)) use = false;
        int64_t strips = 0;
        for (int64_t p = 0; use && p < n_pairs; p++) { if (h_alen[p] < 1 || h_blen[p] < 1) use = false; strips += (h_alen[p] + LH - 1) / LH; }
        const int64_t lat_max = getenv("GNX_LAT_MAX") ? atoll(getenv("GNX_LAT_MAX")) : (int64_t)14 * c.n_cu; // (measured crossovers, tools/lat_crossover.py: 150 x 10 000 ~3 600 strips, 10 kb x 10 kb ~4 000, 1 kb x 1 kb and 3 kb x 3 kb beyond 8 000)
        if (use && (forced || strips <= lat_max)) {
            rc = run_device_lat(prm, kp, tp, affine, local, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, d_smat, h_soff, smat16);
            if (rc == -2) { // the bug trap of its hand-over fired: once more without it
                t_no_lat = true;
                rc = run_device(prm, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, d_smat, h_soff, gsw, d_endpos, no_fast_path, smat16);
                t_no_lat = false;
                return rc;
            }
            if (rc != -1) return rc;
        }
    }
    // ---- fast path: every alpha fits one strip, long beta, global affine with gapOpen <= 0 ----
    {
        const char *fpenv = getenv("GNX_FASTPATH");
        bool fp = affine && !d_smat && !no_fast_path && prm->gap_open <= 0 && !(fpenv && fpenv[0] == '0') && first_oor < 0;
        // AffineGapLocal(target, query): the same sweep on the transposed problem (rows = query), see fp_sweep_kernel<.., XP>
        const bool xp = local;
        if (xp && (prm->gap_extend >= 0 || prm->gap_extend <= -8000)) fp = false;
        const int64_t *h_rows = xp ? h_blen : h_alen, *h_cols = xp ? h_alen : h_blen;
        // fp_sweep_kernel keeps an int16 profile of 4*(s - 2e); its padding rows need 4*|gapOpen| well inside int16
        if (prm->gap_open <= -8000) fp = false;
        for (int x = 0; x < 25; x++) { const int64_t v = 4 * (prm->scores[x] - 2 * prm->gap_extend); if (v > 32767 || v < -32000) fp = false; }
        // Row blocks of 160 rows per pair (fp_sweep_kernel's ROLE): 1 = the read fits one block; 2 .. FP_MAXS = swept as that many blocks
        // (AffineGapLocal: of its query, the rows of the transposed problem); 0 = not for the fast path.  The walk re-fills ~256 columns per row block whatever the window
        // length, so the path pays from ~768 columns on (a third of the cells again at half the sweep's rate).
        const bool forced = fpenv && fpenv[0] == '2'; // no shape rules (tests)
        // Round 3: for big batches the path pays at ANY window length -- its plans are built on the device, the general path's on the host
        // (~35 ns per pair: 400 000 pairs of 150 x 216 take 23.8 ms on the general path, 6.9 ms here; 50 x 128: 18.8 / 4.5 ms); small
        // batches of short windows stay on the general path (4 000 pairs of 150 x 256: 0.41 / 0.47 ms).
        const int64_t min_cols = t_min_cols ? t_min_cols : (n_pairs >= 8192 ? 32 : 768); // (t_min_cols: the rule of the whole batch, for the groups of a mixed one)
        auto key_of = [&](int64_t n, int64_t m) -> int {
            if (n < 1 || m < 1 || m > 0x3fffffff || (n + m + 2) * std::max<int64_t>(maxpen, 1) >= ((int64_t)1 << 27)) return 0;
            const int64_t Sp = (n + H - 1) / H;
            if (Sp > FP_MAXS) return 0;
            if (!forced && m < min_cols) return 0;
            return (int)Sp;
        };
        int64_t n_hi = 0, cntk[FP_MAXS + 1] = {0};
        for (int64_t p = 0; fp && p < n_pairs; p++) { cntk[key_of(h_rows[p], h_cols[p])]++; n_hi = std::max(n_hi, h_rows[p]); }
        int S = 0, distinct = 0;
        for (int k = 0; k <= FP_MAXS; k++) if (cntk[k]) { distinct++; S = k; }
        if (distinct == 1 && S == 0) fp = false;
        // 8 pairs per wave and row block: a small batch of long reads leaves the GPU half empty (the general path runs 4 pairs per wave and strip)
        if (distinct == 1 && S >= 3 && n_pairs * S < 8192 && !forced) fp = false;
        // Round 4: a small batch of one-block reads is a handful of waves whose time is ONE wave's 10 000 dependent steps on either path; the
        // general path then has its directions, the fast path still owes its walk / re-fill rounds (150 x 10 000: 1 pair 3.3 / 5.2 ms,
        // 128 pairs 2.6 / 3.4 ms, 2 048 pairs 2.7 / 3.4 ms, 8 192 pairs 6.5 / 3.7 ms; 150 x 3 000 x 1 024: 0.9 / 1.7 ms -- what a loop of
        // single align.AffineGap calls sees).  While the stored directions (0.75 B per cell) stay small.
        // GNX_FP_SMALL=1: no such rule (the test suites: their small batches are there to exercise the fast path)
        const char *fps = getenv("GNX_FP_SMALL");
        if (fp && !xp && distinct == 1 && S == 1 && n_pairs < 3072 && !forced && !(fps && fps[0] == '1')) {
            int64_t cells = 0;
            for (int64_t p = 0; p < n_pairs; p++) cells += h_rows[p] * h_cols[p];
            if (cells < ((int64_t)1 << 32)) fp = false;
        }
        if (fp && distinct > 1 && n_pairs < 256 && !forced) fp = false;
        if (fp && distinct > 1) {
            // Mixed batch: one uniform sub-batch per number of row blocks, each on its fast path (those not for it: general path),
            // the results put back into input order
            std::vector<int> idx[FP_MAXS + 1];
            std::vector<int64_t> hal[FP_MAXS + 1], hbl[FP_MAXS + 1];
            for (int64_t p = 0; p < n_pairs; p++) { const int gq = key_of(h_rows[p], h_cols[p]); idx[gq].push_back((int)p); hal[gq].push_back(h_alen[p]); hbl[gq].push_back(h_blen[p]); }
            if ((rc = c.mx_idx.ensure((size_t)n_pairs * 4))) return rc;
            if ((rc = c.mx_tab.ensure((size_t)n_pairs * 16))) return rc;
            if ((rc = c.mx_score.ensure((size_t)n_pairs * 8))) return rc;
            if ((rc = c.mx_off.ensure((size_t)(n_pairs + FP_MAXS + 1) * 8))) return rc;
            if ((rc = c.mx_ops.ensure((size_t)std::max<int64_t>(ops_capacity, 1) * sizeof(gnx_cigar)))) return rc;
            if ((rc = c.nops.ensure((size_t)n_pairs * 8))) return rc;
            if ((rc = c.misc.ensure(64))) return rc;
            int *d_idx = reinterpret_cast<int *>(c.mx_idx.p);
            int64_t g0[FP_MAXS + 2] = {0}; // first pair of each group in the gathered order
            for (int k = 0; k <= FP_MAXS; k++) {
                g0[k + 1] = g0[k] + (int64_t)idx[k].size();
                if (!idx[k].empty()) HIPCHK(hipMemcpyAsync(d_idx + g0[k], idx[k].data(), idx[k].size() * 4, hipMemcpyHostToDevice, stream));
            }
            HIPCHK(hipStreamSynchronize(stream));
            gnx_timing tsum = {};
            int64_t tot_all = 0, obase[FP_MAXS + 1] = {0};
            bool capfail = false;
            for (int k = 0; k <= FP_MAXS; k++) {
                const int ng = (int)idx[k].size();
                if (!ng) continue;
                const int *gi = d_idx + g0[k];
                int64_t *gas = reinterpret_cast<int64_t *>(c.mx_tab.p) + 2 * g0[k], *gbs = gas + ng;
                int64_t *gsc = reinterpret_cast<int64_t *>(c.mx_score.p) + g0[k], *goff = reinterpret_cast<int64_t *>(c.mx_off.p) + g0[k] + k;
                hipLaunchKernelGGL(fp_redo_gather_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, stream, gi, ng, d_as, d_bs, gas, gbs);
                HIPCHK(hipGetLastError());
                obase[k] = std::min(tot_all, ops_capacity);
                int64_t tg = 0;
                const int64_t outer_min_cols = t_min_cols;
                t_min_cols = min_cols;
                rc = run_device(prm, ng, d_a, gas, d_b, gbs, hal[k].data(), hbl[k].data(), gsc, reinterpret_cast<gnx_cigar *>(c.mx_ops.p) + obase[k], ops_capacity - obase[k], goff, &tg, stream,
                                nullptr, nullptr, 0, nullptr, k == 0, false);
                t_min_cols = outer_min_cols;
                if (rc == GNX_ECAPACITY) capfail = true;
                else if (rc) return rc;
                tot_all += tg;
                tsum.fill_ms += c.timing.fill_ms; tsum.traceback_ms += c.timing.traceback_ms; tsum.total_ms += c.timing.total_ms; tsum.cells += c.timing.cells;
                tsum.n_launches += c.timing.n_launches; tsum.trace_bytes += c.timing.trace_bytes; tsum.dominant_ms += c.timing.dominant_ms;
                tsum.dominant_launches += c.timing.dominant_launches; tsum.fast_path = std::max(tsum.fast_path, c.timing.fast_path);
            }
            c.timing = tsum;
            if (out_total) *out_total = tot_all;
            if (capfail || tot_all > ops_capacity) { set_err("CIGAR buffer too small: need %s%lld elements", "", (long long)tot_all); return GNX_ECAPACITY; }
            int *d_err = reinterpret_cast<int *>(c.misc.p);
            int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
            int64_t *d_cnt = reinterpret_cast<int64_t *>(c.nops.p);
            HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, stream));
            for (int k = 0; k <= FP_MAXS; k++) {
                const int ng = (int)idx[k].size();
                if (!ng) continue;
                hipLaunchKernelGGL(mix_counts_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, stream, d_idx + g0[k], ng,
                                   reinterpret_cast<const int64_t *>(c.mx_off.p) + g0[k] + k, reinterpret_cast<const int64_t *>(c.mx_score.p) + g0[k], d_cnt, d_score);
            }
            if ((rc = launch_scan(d_cnt, (int)n_pairs, d_ops_off, d_carry, stream))) return rc;
            for (int k = 0; k <= FP_MAXS; k++) {
                const int ng = (int)idx[k].size();
                if (!ng) continue;
                hipLaunchKernelGGL(mix_copy_kernel, dim3((unsigned)ng), dim3(256), 0, stream, d_idx + g0[k], ng, reinterpret_cast<const int64_t *>(c.mx_off.p) + g0[k] + k,
                                   reinterpret_cast<const gnx_cigar *>(c.mx_ops.p) + obase[k], d_ops_off, d_ops, ops_capacity, d_err);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));
            return GNX_OK;
        }
        const bool two = S >= 2;
        const int rows_per_lane = (two || n_hi > 19 * G8) ? 20 : 19;
        if (fp) {
            // sub-batches whose fast-path working set (checkpoints, planes, window slots, staging) fits the workspace
            const int Kw = S > 1 ? std::min(FP_SPEC, S) : 1; // windows per request (at most)
            const size_t fixed = (size_t)Kw * ((size_t)fp_spec_wwords(Kw) * QA * G * 16 + H * 4 + G * 4 + 2 * sizeof(PairPlan)) + fp_cap(S) * sizeof(gnx_cigar) + sizeof(FpState) + sizeof(PairPlan) + 72;
            // ... of about equal size (a small last sub-batch would leave most of the GPU idle for the length of a sweep wave)
            std::vector<int64_t> cb{0};
            size_t acc_b = 0, total_b = 0;
            const size_t budget = (size_t)(c.ws_limit - c.ws_limit / 8);
            auto pair_bytes = [&](int64_t p) { return fixed + (size_t)((h_cols[p] - 1) / CKW) * h_rows[p] * 8 + (size_t)FP_PLANES * ((h_cols[p] + 30) / 16) * 4 + (size_t)(S - 1) * (h_cols[p] + 1) * 8; };
            for (int64_t p = 0; p < n_pairs; p++) { const size_t b = pair_bytes(p); if (b > budget) { fp = false; break; } total_b += b; }
            const size_t n_sub = (total_b + budget - 1) / budget, target = n_sub ? std::min(budget, total_b / n_sub + (size_t)(1 << 20)) : budget;
            for (int64_t p = 0; fp && p < n_pairs; p++) {
                const size_t b = pair_bytes(p);
                if (acc_b > 0 && acc_b + b > target) { cb.push_back(p); acc_b = 0; } // target <= budget, and no single pair exceeds the budget
                acc_b += b;
            }
            cb.push_back(n_pairs);
            rc = -1;
            KParams kpx = kp;
            if (xp) { // transposed score table; column 0 of the transposed problem is the reference's row 0: an ordinary gap
                for (int a = 0; a < 5; a++) for (int b2 = 0; b2 < 5; b2++) kpx.sc4[a * 5 + b2] = kp.sc4[b2 * 5 + a];
                kpx.d00_4 = kp.o4; kpx.ecol4 = kp.e4;
            }
            for (size_t ch = 0; fp && ch + 1 < cb.size(); ch++) {
                const int64_t b = cb[ch], e = cb[ch + 1];
                if (xp) rc = run_device_fp(prm, kpx, tp, e - b, d_b, d_bs + b, d_a, d_as + b, h_blen + b, h_alen + b, rows_per_lane, d_score + b, d_ops, ops_capacity,
                                           d_ops_off + b, out_total, stream, ch == 0, true, S);
                else rc = run_device_fp(prm, kp, tp, e - b, d_a, d_as + b, d_b, d_bs + b, h_alen + b, h_blen + b, rows_per_lane, d_score + b, d_ops, ops_capacity,
                                        d_ops_off + b, out_total, stream, ch == 0, false, S);
                // a CIGAR buffer that is too small does not end the loop: the remaining sub-batches still count their runs (the offset
                // carry runs through them), so that the total handed back with GNX_ECAPACITY is that of the whole batch
                if (rc != GNX_OK && rc != GNX_ECAPACITY) break;
            }
            if (fp && rc != -1) return rc;
        }
    }
    // ---- everything below reads beta as bytes (BetaBytes): windows of the packed resident reference are unpacked first.  What gets
    // here with a packed reference is what the fast path does not take -- windows of fewer than 768 columns, gapOpen > 0, the constant
    // gap functions, the few pairs whose CIGAR overflowed the fast path's staging area ----
    if (kp.b2) {
        std::vector<int64_t> uoff((size_t)n_pairs + 1, 0);
        for (int64_t p = 0; p < n_pairs; p++) uoff[(size_t)p + 1] = uoff[(size_t)p] + h_blen[p];
        if ((rc = c.unpk_b.ensure((size_t)uoff[(size_t)n_pairs] + 64))) return rc;
        if ((rc = c.unpk_off.ensure((size_t)(n_pairs + 1) * 8))) return rc;
        HIPCHK(hipMemcpyAsync(c.unpk_off.p, uoff.data(), (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream)); // (uoff is pageable)
        hipLaunchKernelGGL(unpack_windows_kernel, dim3((unsigned)n_pairs), dim3(256), 0, stream, kp, d_bs, reinterpret_cast<const int64_t *>(c.unpk_off.p), (int)n_pairs,
                           reinterpret_cast<uint8_t *>(c.unpk_b.p));
        HIPCHK(hipGetLastError());
        d_b = reinterpret_cast<const uint8_t *>(c.unpk_b.p); d_bs = reinterpret_cast<const int64_t *>(c.unpk_off.p);
        kp.b2 = nullptr; kp.bflag = nullptr; kp.brank = nullptr; kp.bexc = nullptr;
    }
    // ---- constant gap without a stored direction matrix (const_long.hip.h): pairs of more than one strip; GNX_CLONG=0 / 2 = never / always ----
    const bool oor = first_oor >= 0;
    if (!gsw && !d_smat && !local && (!affine || (prm->gap_open <= 0 && (oor || !getenv("GNX_NO_HFORM"))))) { // (GNX_CLONG also governs the affine form, affine_long.hip.h)
        const char *cl = getenv("GNX_CLONG");
        bool use = oor || !(cl && cl[0] == '0'), any_multi = false;
        // moving bases: what a strip holds at one time -- 160 rows x 16 columns of an anti-diagonal band, the row above, the snapshot -- must fit
        // int32 around the strip's base: neighbouring cells differ by at most one substitution score + two gap opens + two extensions
        const int64_t step4 = 4 * (max_abs_pen(prm, false) + 2 * llabs((long long)prm->gap_open) + (affine ? 2 * llabs((long long)prm->gap_extend) : 0));
        const bool spread_ok = (int64_t)(H + G + std::max(CKC, CKA) + 64) * step4 < ((int64_t)1 << 28); // (a strip moves its base every snapshot: the keys drift that many steps in between)
        const char *rbe = getenv("GNX_REBASE");
        const bool rebase = spread_ok && (oor || (rbe && rbe[0] == '1')); // GNX_REBASE=1 (tests): every pair of this path on moving bases
        if (oor && !spread_ok) use = false;
        // the whole wave on one pair (affine_long64.hip.h, const_long64.hip.h): launches of up to three pairs, which would leave lane groups of the 16-lane
        // kernels' waves idle -- and, since the walk farm (farm64.hip.h), launches of up to 64 pairs of at least two 640-row strips each: the sweeps
        // are level, the walk of a pair is rounds on 17 workgroups instead of one wave (16 x AffineGap 200 kb x 200 kb: 0.69 -> 0.15 s; ConstGap
        // 64 x (20 kb x 100 kb) 55 -> 49 ms, 128 pairs still 76 -> 70; tools/few_long_pairs.py).  GNX_W64 = 0 / 2: never / for every launch of this path
        const char *w64e = getenv("GNX_W64");
        bool few_long = n_pairs <= 64 && w64_farm_tiles() > 0;
        for (int64_t p = 0; few_long && p < n_pairs; p++) if (h_alen[p] < 2 * H64) few_long = false;
        t_w64_r = R; t_w64_rc = R;
        // (steps between two moves of a strip's base: the widest snapshot spacing the keys' spread admits, for the farm's affine sweep)
        t_w64_ck = CKA;
        if (affine && w64_farm_tiles() > 0) for (int v = w64_farm_ck(); v > CKA; v >>= 1) if ((int64_t)(H64 + G64 + v + 64) * step4 < ((int64_t)1 << 28)) { t_w64_ck = v; break; }
        const int64_t ck_w64 = std::max<int64_t>(CKC64, t_w64_ck);
        const bool w64 = !no_pipe() && (int64_t)(H64 + G64 + ck_w64 + 64) * step4 < ((int64_t)1 << 28) && !(w64e && w64e[0] == '0') && (n_pairs <= 3 || few_long || (w64e && w64e[0] == '2'));
        if (w64 && w64_farm_tiles() > 0) { if (affine) t_w64_r = w64_pick_rows(c, true, n_pairs, h_alen, h_blen, step4, t_w64_ck); else t_w64_rc = w64_pick_rows(c, false, n_pairs, h_alen, h_blen, step4, CKC64); }
        long double cells_ld = 0, dir_bytes = 0, rows_ld = 0, cols_ld = 0;
        for (int64_t p = 0; use && p < n_pairs; p++) {
            if (h_alen[p] < 1 || h_blen[p] < 1) use = false;
            if (h_alen[p] > H) any_multi = true;
            cells_ld += (long double)h_alen[p] * h_blen[p];
            rows_ld += (long double)h_alen[p]; cols_ld += (long double)h_blen[p];
            dir_bytes += (long double)((h_alen[p] + H - 1) / H) * ((h_blen[p] + 30) / 16) * (affine ? QA : QC) * G * 16;
        }
        // It pays when the pairs are big: the sweep saves ~1.3e-13 s per cell against the recording fill, the fused re-fill + walk
        // costs ~0.3 us per pair more than the one-lane-per-pair traceback over a stored matrix (tools/bench_shapes.py: 250..3200 x
        // 10 000 gain 15..45 %, 100 000 pairs of 1000 x 1200 lose 2x) -- or when the stored matrix would not fit the workspace at all.
        // Affine (affine_long.hip.h): a global alignment against a long window crosses every column tile, so the re-fill costs about
        // 1 / strips of a full fill at the walk kernel's low occupancy: it pays from ~10 strips on, for batches (1600 x 10 000: +21 %,
        // 3200 x 10 000: +46 %; 480 / 800 x 10 000 and a single 10 kb x 10 kb pair are faster over the stored matrix).
        // (Since the fast path's row blocks take the batches of pairs x blocks >= 8192, what reaches this point with the affine functions
        // is small batches, and there the stored matrix wins -- 256 x (3200 x 10 000): 5.96 ms against 10.2 ms: affine only when the matrix does not fit.)
        const bool big = (cells_ld >= 2.0e6L * (long double)n_pairs && !affine) || dir_bytes > (long double)c.ws_limit;
        // Round 3 (one pair per walk workgroup, tiles of 224 steps): short reads against LONG windows gain as well -- the sweep saves
        // ~8.5e-14 s per cell, the walk costs ~7e-8 s per pair plus ~1e-11 s per column: ConstGap 150 x 10 000 18.6 -> 16.1 ms per 65 536
        // pairs (1000 pairs: 1.81 -> 1.25 ms), but 150 x 2000, 500 x 600 and 1000 x 1200 lose 1.4 .. 2.6 x and stay on the stored matrix.
        const bool long_windows = !affine && cells_ld >= 1.4e6L * (long double)n_pairs && cols_ld >= 48.0L * rows_ld;
        const bool mega_forced = getenv("GNX_MEGA_STRIPS") != nullptr; // (tests: row panels of a few strips)
        if (use && (oor || (any_multi && big) || long_windows || (cl && cl[0] == '2') || mega_forced)) {
            rc = (mega_forced && spread_ok) ? -1 : run_device_clong(prm, kp, tp, affine, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, rebase || w64, w64);
            if (rc != -1) return rc;
            // one pair's bottom rows + snapshots do not fit what the device has: row panels (run_device_mega; always on moving bases)
            if (spread_ok) {
                rc = run_device_mega(prm, kp, tp, affine, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, w64);
                if (rc != -1) return rc;
            }
            if (oor) { set_err("pair %s%lld needs more snapshot workspace than the device has free", "", (long long)first_oor); return GNX_ENOMEM; }
        }
    }
    if (oor) {
        // what is left has absolute int32 keys -- AffineGapLocal, gapOpen > 0, scores too big for moving bases: the int64 kernel (lat_wide.hip.h),
        // limited by the workspace its stored direction matrix needs (1 B per cell), not by a range.  The chunk / graph variants keep int32.
        // (round 6: the chunk / multiple-alignment variants too -- their explicit score matrices must then hold plain 4 * s int32 entries: run_host_scored)
        if (!gsw && !(d_smat && smat16) && (n_pairs == 0 || (long double)(h_alen[first_oor] + h_blen[first_oor] + 2) * (long double)std::max<int64_t>(maxpen, 1) < 1.0e17L)) {
            rc = run_device_lat(prm, kp, tp, affine, local, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, d_smat, h_soff, false, true);
            if (rc >= 0) return rc;
            if (rc == -2) { set_err("a strip of the int64 kernel waited more than 5 s for the strip above it%s", ""); return GNX_EDEVICE; }
            set_err("pair %s%lld is beyond the int32 range of its mode and its direction matrix does not fit the workspace (or a sequence is empty)", "", (long long)first_oor);
            return GNX_ENOMEM;
        }
        set_err("pair %s%lld exceeds the int32 DP range of this variant (graph-extension DPs)", "", (long long)first_oor);
        return GNX_ERANGE;
    }
    // ---- plan ----
    const auto t_plan0 = std::chrono::steady_clock::now();
    bool p16 = true; // 4*score fits a signed 16-bit profile entry
    for (int x = 0; x < 25; x++) { const int64_t v = 4 * (prm->scores[x] - 2 * (affine ? prm->gap_extend : 0)); if (v > 32767 || v < -32768) p16 = false; }
    if (!getenv("GNX_FORCE_P16")) p16 = false; // int32 profile: plain 2-cycle VGPR add instead of a 4-cycle SDWA add
    bool hform = affine && prm->gap_open <= 0;
    if (getenv("GNX_NO_HFORM")) hform = false;
    const int Q = affine ? QA : QC;
    // the plans are built in pinned host memory kept by the context: a fresh std::vector of 96 B per pair costs a page fault per 42 pairs
    // and an upload from pageable memory -- 400 000 pairs of 150 x 216: 16 of the call's 24 ms (round 3)
    if ((rc = c.h_plans.ensure((size_t)std::max<int64_t>(n_pairs, 1) * sizeof(PairPlan)))) return rc;
    struct PlanSpan { PairPlan *p; PairPlan &operator[](size_t i) const { return p[i]; } PairPlan *data() const { return p; } };
    const PlanSpan plans{reinterpret_cast<PairPlan *>(c.h_plans.p)};
    std::vector<int64_t> chunk_begin;
    int64_t cells = 0;
    {
        int64_t toff = 0, hoff = 0, roff = 0, doff = 0;
        chunk_begin.push_back(0);
        const int64_t trace_limit_u4 = std::max<int64_t>(c.ws_limit / 16, 1);
        for (int64_t p = 0; p < n_pairs; p++) {
            const int64_t n = h_alen[p], m = h_blen[p];
            if (n < 0 || m < 0 || n > 0x3fffffff || m > 0x3fffffff) { set_err("bad sequence length at pair %s%lld", "", (long long)p); return GNX_EINVAL; }
            if (lowmem && prm->checkersize_i != prm->checkersize_j && n > prm->checkersize_i) {
                // the reference indexes its saved columns with checkersize_j where checkersize_i is meant
                // (affineGap.go:252-254, constGap.go:207): undefined for non-square tiles once n > checkersize_i
                set_err("non-square checkerboards with n > checkersize_i are undefined in the reference (pair %s%lld)", "", (long long)p); return GNX_EINVAL;
            }
            if (lowmem && (n < 1 || m < 1)) { set_err("empty sequence at pair %s%lld: the reference never terminates on it", "", (long long)p); return GNX_EEMPTY; }
            if ((n + m + 2) * std::max<int64_t>(maxpen, 1) >= ((int64_t)1 << 27)) { set_err("pair %s%lld exceeds the int32 DP range", "", (long long)p); return GNX_ERANGE; }
            if (gsw == 2 && (n > 4095 || m > 4095 || (n + m + 2) * std::max<int64_t>(maxpen, 1) >= ((int64_t)1 << 19))) {
                set_err("pair %s%lld exceeds the range of the packed (score, column) maximum of RightDynamicAln", "", (long long)p); return GNX_ERANGE;
            }
            PairPlan &pl = plans[(size_t)p];
            pl.n = (int32_t)n; pl.m = (int32_t)m;
            pl.src = 0; pl.col_off = 0; pl.ckpt_off = 0; pl.rowi_off = 0; pl.s_off = 0; pl.s_pitch = 0; // src is chunk-relative, set below
            pl.strips = (m > 0) ? (int32_t)((n + H - 1) / H) : 0;
            pl.words = (int32_t)((m + 15 + 15) / 16);
            const int64_t tsz = (int64_t)pl.strips * pl.words * Q * G;
            if (tsz > trace_limit_u4) { set_err("pair %s%lld needs more direction-matrix workspace than the limit", "", (long long)p); return GNX_ENOMEM; }
            if (toff + tsz > trace_limit_u4) { // start a new chunk (keep chunks 4-aligned so waves stay whole)
                int64_t cb = p & ~(int64_t)3;
                if (cb <= chunk_begin.back()) cb = p;
                // re-plan the pairs moved into the new chunk
                toff = 0; hoff = 0; roff = 0; doff = 0;
                for (int64_t q2 = cb; q2 < p; q2++) {
                    PairPlan &pq = plans[(size_t)q2];
                    pq.trace_off = toff; pq.hcol_off = hoff; pq.rowbuf_off = roff; pq.dcol_off = doff;
                    toff += (int64_t)pq.strips * pq.words * Q * G; hoff += pq.n; roff += (int64_t)std::max(pq.strips - 1, 0) * (pq.m + 1); doff += (int64_t)pq.strips * G;
                }
                chunk_begin.push_back(cb);
            }
            pl.trace_off = toff; pl.hcol_off = hoff; pl.rowbuf_off = roff; pl.dcol_off = doff;
            if (h_soff) { pl.s_off = h_soff[p]; pl.s_pitch = (int64_t)std::max<int32_t>(pl.strips, 1) * H; }
            toff += tsz; hoff += n; roff += (int64_t)std::max(pl.strips - 1, 0) * (m + 1); doff += (int64_t)pl.strips * G;
            cells += n * m;
        }
        chunk_begin.push_back(n_pairs);
        for (size_t ch = 0; ch + 1 < chunk_begin.size(); ch++)
            for (int64_t p = chunk_begin[ch]; p < chunk_begin[ch + 1]; p++) plans[(size_t)p].src = (int32_t)(p - chunk_begin[ch]);
    }
    // workspace sizes = max over chunks
    int64_t max_t = 1, max_h = 1, max_r = 1, max_d = 1;
    for (size_t ch = 0; ch + 1 < chunk_begin.size(); ch++) {
        int64_t t = 0, h = 0, r = 0, d = 0;
        for (int64_t p = chunk_begin[ch]; p < chunk_begin[ch + 1]; p++) {
            const PairPlan &pl = plans[(size_t)p];
            t += (int64_t)pl.strips * pl.words * Q * G; h += pl.n; r += (int64_t)std::max(pl.strips - 1, 0) * (pl.m + 1); d += (int64_t)pl.strips * G;
        }
        max_t = std::max(max_t, t); max_h = std::max(max_h, h); max_r = std::max(max_r, r); max_d = std::max(max_d, d);
    }
    if ((rc = c.dcol.ensure((size_t)max_d * 4))) return rc;
    DevBuf &trbuf = c.trace;
    if ((rc = trbuf.ensure((size_t)max_t * 16))) return rc;
    if ((rc = c.hcol.ensure((size_t)max_h * 4))) return rc;
    if ((rc = c.rowbuf.ensure((size_t)max_r * 8))) return rc;
    c.fpc_ptr = nullptr; // the general path's plans replace the fast path's
    if ((rc = c.plans.ensure((size_t)n_pairs * sizeof(PairPlan)))) return rc;
    if ((rc = c.nops.ensure((size_t)n_pairs * 8))) return rc;
    if ((rc = c.misc.ensure(64))) return rc;
    int *d_err = reinterpret_cast<int *>(c.misc.p);
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, stream));
    HIPCHK(hipMemcpyAsync(c.plans.p, plans.data(), (size_t)n_pairs * sizeof(PairPlan), hipMemcpyHostToDevice, stream));
    if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx] general path: %lld pairs, plans built and queued in %.3f ms on the host\n", (long long)n_pairs,
                                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count());

    // ---- launches ----
    double fill_ms = 0, tb_ms = 0;
    int64_t trace_bytes = 0;
    HIPCHK(hipEventRecord(c.ev[0], stream));
    const size_t nchunks = chunk_begin.size() - 1;
    for (size_t ch = 0; ch < nchunks; ch++) {
        const int64_t b = chunk_begin[ch], e = chunk_begin[ch + 1];
        const int np = (int)(e - b);
        if (np <= 0) continue;
        const PairPlan *dpl = reinterpret_cast<const PairPlan *>(c.plans.p) + b;
        uint4 *dtrace = reinterpret_cast<uint4 *>(trbuf.p);
        int *dh = reinterpret_cast<int *>(c.hcol.p);
        int2 *drb = reinterpret_cast<int2 *>(c.rowbuf.p);
        unsigned *ddc = reinterpret_cast<unsigned *>(c.dcol.p);
        int64_t *dn = reinterpret_cast<int64_t *>(c.nops.p) + b;
        bool multi = false;
        for (int64_t q2 = b; q2 < e; q2++) if (plans[(size_t)q2].strips > 1) { multi = true; break; }
        // multi-strip chunks: one workgroup per (group of 4 pairs, strip), pipelined through the row buffer (fill_affine_kernel)
        const int2 *d_smap = nullptr;
        int *d_sprog = nullptr;
        int64_t n_blocks = (np + 3) / 4;
        // pipelined strips pay when one wave per 4 pairs cannot fill the GPU and the strips are long enough to overlap;
        // with plenty of pairs (or short beta) the strips of a group would only wait for each other
        int64_t m_maxc = 0;
        for (int64_t q2 = b; q2 < e; q2++) m_maxc = std::max<int64_t>(m_maxc, plans[(size_t)q2].m);
        const bool piped = multi && n_blocks < 3072 && m_maxc >= 8 * RB_PUB && !no_pipe(); // GNX_NO_PIPE: A/B check of the hand-over protocol
        if (piped) {
            std::vector<int2> smap;
            for (int gq = 0; gq < (np + 3) / 4; gq++) {
                int smax = 0;
                for (int q3 = 0; q3 < 4 && gq * 4 + q3 < np; q3++) smax = std::max(smax, (int)plans[(size_t)(b + gq * 4 + q3)].strips);
                for (int st2 = 0; st2 < smax; st2++) smap.push_back(make_int2(gq, st2));
            }
            n_blocks = (int64_t)smap.size();
            if (n_blocks > 0x7fffffff) { set_err("too many strips in one chunk%s", ""); return GNX_ENOMEM; }
            if ((rc = c.strip_map.ensure((size_t)std::max<int64_t>(n_blocks, 1) * 16 + 8))) return rc; // map, progress words, claim words, test switch
            d_smap = reinterpret_cast<const int2 *>(c.strip_map.p);
            d_sprog = reinterpret_cast<int *>(reinterpret_cast<char *>(c.strip_map.p) + (size_t)std::max<int64_t>(n_blocks, 1) * 8);
            HIPCHK(hipMemcpyAsync(c.strip_map.p, smap.data(), (size_t)n_blocks * 8, hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemsetAsync(d_sprog, 0, (size_t)std::max<int64_t>(n_blocks, 1) * 8 + 8, stream));
            if ((rc = claim_test_switch(d_sprog + 2 * n_blocks, stream))) return rc;
            HIPCHK(hipStreamSynchronize(stream)); // smap is a local
        }
        const dim3 gridF((unsigned)n_blocks), blockF(64);
        const dim3 gridT((unsigned)((np + 63) / 64)), blockT(64);
        HIPCHK(hipEventRecord(c.ev[1], stream));
        if (affine) {
#define GNX_LAUNCH_AFF(L_, M_, P_, H_) hipLaunchKernelGGL((fill_affine_kernel<L_, M_, P_, H_>), gridF, blockF, 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, dtrace, dh, drb, ddc, (const int2 *)nullptr, d_err, (const int *)nullptr, d_smap, d_sprog)
#define GNX_LAUNCH_AFF2(L_, M_, P_) do { if (hform) GNX_LAUNCH_AFF(L_, M_, P_, true); else GNX_LAUNCH_AFF(L_, M_, P_, false); } while (0)
#define GNX_LAUNCH_SC1(M_, P_, H_) hipLaunchKernelGGL((fill_affine_kernel<false, M_, P_, H_, false, true>), gridF, blockF, 0, stream, dpl, np, d_a, d_as, d_b, d_bs, kp, dtrace, dh, drb, ddc, (const int2 *)nullptr, d_err, d_smat, d_smap, d_sprog)
#define GNX_LAUNCH_SC(M_, H_) do { if (smat16) GNX_LAUNCH_SC1(M_, true, H_); else GNX_LAUNCH_SC1(M_, false, H_); } while (0)
            const int sel = d_smat ? 8 : ((local ? 4 : 0) | (multi ? 2 : 0) | (p16 ? 1 : 0));
            switch (sel) {
            case 8:
                if (multi) { if (hform) GNX_LAUNCH_SC(true, true); else GNX_LAUNCH_SC(true, false); }
                else { if (hform) GNX_LAUNCH_SC(false, true); else GNX_LAUNCH_SC(false, false); }
                break;
            case 0: GNX_LAUNCH_AFF2(false, false, false); break;
            case 1: GNX_LAUNCH_AFF2(false, false, true); break;
            case 2: GNX_LAUNCH_AFF2(false, true, false); break;
            case 3: GNX_LAUNCH_AFF2(false, true, true); break;
            case 4: GNX_LAUNCH_AFF2(true, false, false); break;
            case 5: GNX_LAUNCH_AFF2(true, false, true); break;
            case 6: GNX_LAUNCH_AFF2(true, true, false); break;
            default: GNX_LAUNCH_AFF2(true, true, true); break;
            }
#undef GNX_LAUNCH_SC
#undef GNX_LAUNCH_SC1
#undef GNX_LAUNCH_AFF2
#undef GNX_LAUNCH_AFF
        } else {
            // pipelined strips: int16 profile when every entry (4*(s - 2g) + 1 rebased, 4*s + 3 for the gsw variants) fits -- half the
            // LDS, 24 instead of 12 strips resident per CU (C5 miniature fill 125.7 -> 117.9 ms).  One wave per 4 pairs is paced by
            // its instruction count and keeps the int32 profile (C2-shaped ConstGap: 5.44e12 vs 5.18e12 cells/s with the SDWA adds).
            bool cp16 = piped && !getenv("GNX_CONST_P32");
            for (int x = 0; x < 25; x++) { const int64_t v = gsw ? 4 * prm->scores[x] + 3 : 4 * (prm->scores[x] - 2 * prm->gap_open) + 1; if (v > 32767 || v < -32768) cp16 = false; }
#define GNX_LAUNCH_CONST1(M_, G_, P_) hipLaunchKernelGGL((fill_const_kernel<M_, G_, P_>), gridF, blockF, 0, stream, dpl, np, d_a, d_as + b, d_b, d_bs + b, kp, dtrace, dh, drb, ddc, d_err, d_smap, d_sprog)
#define GNX_LAUNCH_CONST(M_, G_) do { if (cp16) GNX_LAUNCH_CONST1(M_, G_, true); else GNX_LAUNCH_CONST1(M_, G_, false); } while (0)
            if (gsw == 1) { if (multi) GNX_LAUNCH_CONST(true, 1); else GNX_LAUNCH_CONST(false, 1); }
            else if (gsw == 2) { if (multi) GNX_LAUNCH_CONST(true, 2); else GNX_LAUNCH_CONST(false, 2); }
            else { if (multi) GNX_LAUNCH_CONST(true, 0); else GNX_LAUNCH_CONST(false, 0); }
#undef GNX_LAUNCH_CONST
#undef GNX_LAUNCH_CONST1
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c.ev[2], stream));
#define GNX_GSW_TB(R_, W_) hipLaunchKernelGGL((gsw_traceback_kernel<R_, W_>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, d_a, d_as + b, d_b, d_bs + b, kp, d_score + b, d_endpos + b, dn, d_ops_off + b, d_ops, ops_capacity, d_err)
        // few, long pairs: one wave per pair, diagonal runs 64 cells at a time (traceback_kernel<.., COOP>)
        int64_t lmax = 0;
        for (int64_t q2 = b; q2 < e; q2++) lmax = std::max<int64_t>(lmax, std::max<int64_t>(plans[(size_t)q2].n, plans[(size_t)q2].m));
        const bool coop = !gsw && np <= 4096 && lmax >= 256; // (round 5: 1 024 pairs of 1 kb x 1 kb -- one lane per pair 1.3 ms, one wave per pair 0.2 ms)
        const dim3 gridC((unsigned)np);
        // ... in a single pass when the staging area (n + m + 2 runs per pair) is small next to the workspace
        bool scr = false;
        if (coop && !getenv("GNX_TB_TWO_PASS")) { // (GNX_TB_TWO_PASS: keep the count + write passes, for the tests)
            std::vector<int64_t> so((size_t)np + 1, 0);
            for (int q2 = 0; q2 < np; q2++) so[(size_t)q2 + 1] = so[(size_t)q2] + plans[(size_t)(b + q2)].n + plans[(size_t)(b + q2)].m + 2;
            const size_t sbytes = (size_t)so[(size_t)np] * sizeof(gnx_cigar);
            if ((int64_t)sbytes <= c.ws_limit / 8) {
                if ((rc = c.tb_scr.ensure(sbytes))) return rc;
                if ((rc = c.tb_scr_off.ensure(((size_t)np + 1) * 8))) return rc;
                HIPCHK(hipMemcpyAsync(c.tb_scr_off.p, so.data(), ((size_t)np + 1) * 8, hipMemcpyHostToDevice, stream));
                HIPCHK(hipStreamSynchronize(stream)); // so is a local
                scr = true;
            }
        }
        gnx_cigar *d_scr = reinterpret_cast<gnx_cigar *>(c.tb_scr.p);
        const int64_t *d_scr_off = reinterpret_cast<const int64_t *>(c.tb_scr_off.p);
        if (gsw == 1) GNX_GSW_TB(false, false);
        else if (gsw == 2) GNX_GSW_TB(true, false);
        else if (scr && affine) hipLaunchKernelGGL((traceback_kernel<true, false, true, true>), gridC, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, d_scr_off, d_scr, (int64_t)0, d_err);
        else if (scr) hipLaunchKernelGGL((traceback_kernel<false, false, true, true>), gridC, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, d_scr_off, d_scr, (int64_t)0, d_err);
        else if (coop && affine) hipLaunchKernelGGL((traceback_kernel<true, false, true>), gridC, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, (const int64_t *)nullptr, (gnx_cigar *)nullptr, (int64_t)0, d_err);
        else if (coop) hipLaunchKernelGGL((traceback_kernel<false, false, true>), gridC, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, (const int64_t *)nullptr, (gnx_cigar *)nullptr, (int64_t)0, d_err);
        else if (affine) hipLaunchKernelGGL((traceback_kernel<true, false>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, (const int64_t *)nullptr, (gnx_cigar *)nullptr, (int64_t)0, d_err);
        else hipLaunchKernelGGL((traceback_kernel<false, false>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, (const int64_t *)nullptr, (gnx_cigar *)nullptr, (int64_t)0, d_err);
        if ((rc = launch_scan(dn, (int)np, d_ops_off + b, d_carry, stream))) return rc;
        if (gsw == 1) GNX_GSW_TB(false, true);
        else if (gsw == 2) GNX_GSW_TB(true, true);
        else if (scr) hipLaunchKernelGGL(reverse_runs_kernel, gridC, dim3(256), 0, stream, dpl, np, d_scr, d_scr_off, dn, d_ops_off + b, d_ops, ops_capacity, d_err);
        else if (coop && affine) hipLaunchKernelGGL((traceback_kernel<true, true, true>), gridC, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, d_ops_off + b, d_ops, ops_capacity, d_err);
        else if (coop) hipLaunchKernelGGL((traceback_kernel<false, true, true>), gridC, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, d_ops_off + b, d_ops, ops_capacity, d_err);
        else if (affine) hipLaunchKernelGGL((traceback_kernel<true, true>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, d_ops_off + b, d_ops, ops_capacity, d_err);
        else hipLaunchKernelGGL((traceback_kernel<false, true>), gridT, blockT, 0, stream, dpl, np, dtrace, dh, ddc, tp, d_score + b, dn, d_ops_off + b, d_ops, ops_capacity, d_err);
#undef GNX_GSW_TB
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c.ev[3], stream));
        { // per-chunk kernel times (the chunks reuse the workspace, so each one is waited for anyway)
            HIPCHK(hipEventSynchronize(c.ev[3]));
            float f1 = 0, f2 = 0;
            HIPCHK(hipEventElapsedTime(&f1, c.ev[1], c.ev[2]));
            HIPCHK(hipEventElapsedTime(&f2, c.ev[2], c.ev[3]));
            fill_ms += f1; tb_ms += f2;
        }
        for (int64_t p = b; p < e; p++) trace_bytes += (int64_t)plans[(size_t)p].strips * plans[(size_t)p].words * Q * G * 16;
    }
    HIPCHK(hipEventRecord(c.ev[2], stream));
    int h_misc[16];
    HIPCHK(hipMemcpyAsync(h_misc, c.misc.p, 64, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float tot = 0;
    HIPCHK(hipEventElapsedTime(&tot, c.ev[0], c.ev[2]));
    c.timing.fill_ms = fill_ms; c.timing.traceback_ms = tb_ms; c.timing.total_ms = tot;
    c.timing.cells = cells; c.timing.n_launches = (int64_t)nchunks; c.timing.trace_bytes = trace_bytes;
    c.timing.dominant_ms = fill_ms; c.timing.dominant_launches = (int64_t)nchunks; c.timing.fast_path = 0;
    int64_t total;
    memcpy(&total, reinterpret_cast<char *>(h_misc) + 16, 8);
    if (out_total) *out_total = total;
    const int ef = h_misc[0];
    if (ef & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (ef & 16) {
        if (t_no_pipe) { set_err("a strip waited more than 5 s for the strip above it%s", ""); return GNX_EDEVICE; }
        if (getenv("GNX_DEBUG")) fprintf(stderr, "[gnx] a pipelined strip timed out: the call runs again with sequential strips\n");
        t_no_pipe = true;
        rc = run_device(prm, n_pairs, d_a, d_as, d_b, d_bs, h_alen, h_blen, d_score, d_ops, ops_capacity, d_ops_off, out_total, stream, d_smat, h_soff, gsw, d_endpos, no_fast_path, smat16);
        t_no_pipe = false;
        return rc;
    }
    if (ef & 2) { set_err("unexpected traceback%s", ""); return GNX_ETRACE; }
    if (ef & 4) { set_err("CIGAR buffer too small: need %s%lld elements", "", (long long)total); return GNX_ECAPACITY; }
    return GNX_OK;
}

// N1 host flow: bases (pairwise sequences or alignment blocks) -> score matrices on the device -> SCORED fill + the
// ordinary highMem traceback -> run lengths times chunk size.  `sp` describes the pairs (offsets into `bases`).
int run_host_scored(const gnx_params *prm, int64_t chunk, bool groups, int64_t n_pairs, std::vector<ScorePair> &sp,
                    const uint8_t *bases, int64_t bases_len, int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off,
                    const uint8_t *bases2 = nullptr, int64_t bases2_len = 0) {
    // bases2: a second host buffer that follows `bases` on the device (the pairwise entry point: alpha_cat, then beta_cat -- no host copy)
    Ctx &c = g_ctx;
    if (!prm || !out_score || !out_ops || !out_ops_off || n_pairs < 0 || chunk < 1 || chunk > (1 << 20)) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    if (prm->mode != GNX_AFFINE_GAP_HIGHMEM) { set_err("the chunk / multiple-alignment variants have AffineGap_highMem semantics (mode %s%lld)", "", (long long)GNX_AFFINE_GAP_HIGHMEM); return GNX_EINVAL; }
    KParams kp0; TbParams tp0; bool aff, loc, low;
    int rc = check_params(prm, kp0, tp0, aff, loc, low);
    if (rc) return rc;
    gnx_params prm2 = *prm; // what the DP sees: gapExtend*chunkSize, |cell score| <= chunkSize*max|score|
    prm2.gap_extend = prm->gap_extend * chunk;
    for (int x = 0; x < 25; x++) prm2.scores[x] = prm->scores[x] * chunk;
    hipStream_t st = c.own_stream;
    const bool dbg = getenv("GNX_DEBUG") != nullptr;
    auto t_now = []() { return std::chrono::steady_clock::now(); };
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const auto t_begin = t_now();
    int64_t smax = 0; // int16 score matrix when every 4 * cell score fits (a cell is a sum of `chunk` scores, or of averages of scores)
    for (int x = 0; x < 25; x++) smax = std::max<int64_t>(smax, llabs((long long)prm->scores[x]));
    // the fill's h-form works on rebased keys: the matrix entries carry the -2e of the diagonal move (e = gapExtend * chunk)
    // a pair beyond the static int32 range of the keys sends the whole call to the int64 kernel (lat_wide_kernel<.., SCORED>): plain int32 entries 4 * s
    bool wide_sc = false;
    {
        const int64_t maxpen2 = max_abs_pen(&prm2, true);
        for (int64_t p = 0; p < n_pairs; p++) if ((sp[(size_t)p].nc + sp[(size_t)p].mc + 2) * std::max<int64_t>(maxpen2, 1) >= ((int64_t)1 << 27)) wide_sc = true;
        if (getenv("GNX_WIDE") && getenv("GNX_WIDE")[0] == '2') wide_sc = true; // (tests: every call)
        if (wide_sc && 4 * chunk * smax > 0x3fffffff) { set_err("chunk score out of range%s", ""); return GNX_ERANGE; }
    }
    t_scored_wide = wide_sc;
    struct ScoredWideReset { ~ScoredWideReset() { t_scored_wide = false; } } scored_wide_reset;
    const bool hform_sc = !wide_sc && prm2.gap_open <= 0 && !getenv("GNX_NO_HFORM");
    const int64_t bias4 = hform_sc ? -8 * prm2.gap_extend : 0;
    const bool s16 = !wide_sc && 4 * chunk * smax + llabs((long long)bias4) <= 32767;
    std::vector<int64_t> hn((size_t)n_pairs), hm((size_t)n_pairs), hso((size_t)n_pairs);
    int64_t stot = 0, worst = 0, maxcols = 1, maxrows = 1, prof_a = 0, prof_b = 0, max_nseq = 1;
    for (int64_t p = 0; p < n_pairs; p++) {
        ScorePair &q = sp[(size_t)p];
        maxrows = std::max<int64_t>(maxrows, q.nc);
        const int64_t strips = std::max<int64_t>((q.nc + H - 1) / H, 1);
        q.s_pitch = strips * H; q.s_off = stot;
        q.pa_off = prof_a; q.pb_off = prof_b; prof_a += q.a_len; prof_b += q.b_len; // (groups: column profiles, score_profiles_kernel)
        max_nseq = std::max<int64_t>(max_nseq, std::max(q.a_nseq, q.b_nseq));
        stot += (int64_t)q.mc * q.s_pitch;
        hn[(size_t)p] = q.nc; hm[(size_t)p] = q.mc; hso[(size_t)p] = q.s_off;
        worst += q.nc + q.mc + 1;
        maxcols = std::max<int64_t>(maxcols, q.mc);
    }
    if ((rc = c.in_a.ensure((size_t)(bases_len + bases2_len) + 16))) return rc;
    if ((rc = c.sc_pairs.ensure((size_t)std::max<int64_t>(n_pairs, 1) * sizeof(ScorePair)))) return rc;
    if ((rc = c.sc_mat.ensure((size_t)std::max<int64_t>(stot, 1) * 4 + 4096))) return rc; // (+ slack: the latency geometry's last strip reads up to 127 padding rows past a column)
    if ((rc = c.sc_err.ensure(16))) return rc;
    const size_t np = (size_t)std::max<int64_t>(n_pairs, 1);
    if ((rc = c.out_score.ensure(np * 8))) return rc;
    if ((rc = c.out_off.ensure((np + 1) * 8))) return rc;
    if (n_pairs) HIPCHK(hipMemcpyAsync(c.sc_pairs.p, sp.data(), (size_t)n_pairs * sizeof(ScorePair), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(c.sc_err.p, 0, 16, st));
    // ---- sub-batches: the bases of sub-batch k + 1 cross PCIe (from the caller's pageable memory: an uploader thread sits in that copy)
    // while the score matrices and the DP of sub-batch k run.  Pairwise entry point only (alpha_cat / beta_cat are in pair order), big inputs only.
    // A sub-batch must still fill the GPU on its own (4 pairs per wave, one wave per group of strips: 8192 pairs = 2048 waves) -- 4096
    // pairs cut in four ran 7.9 instead of 5.1 ms: each quarter takes as long as the whole.  GNX_SCORED_SUB=k forces k sub-batches (tests).
    int K = 1;
    if (!groups && bases2 && n_pairs >= 2) {
        if (bases_len + bases2_len >= ((int64_t)8 << 20)) K = (int)std::min<int64_t>(4, n_pairs / 8192);
        if (const char *e = getenv("GNX_SCORED_SUB")) K = atoi(e);
        K = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(K, 16), n_pairs));
    }
    std::vector<int64_t> kb((size_t)K + 1, n_pairs);
    kb[0] = 0;
    if (K > 1) { // boundaries by bytes of bases
        int next = 1;
        for (int64_t p = 0; p < n_pairs && next < K; p++) {
            const int64_t done_bytes = sp[(size_t)p].a_off + (sp[(size_t)p].b_off - bases_len); // = alpha_off[p] + beta_off[p]
            if (done_bytes * K >= (bases_len + bases2_len) * next) kb[(size_t)next++] = p;
        }
        for (int k = 1; k <= K; k++) kb[(size_t)k] = std::max(kb[(size_t)k], kb[(size_t)k - 1]);
    }
    std::vector<hipEvent_t> evs((size_t)K, nullptr);
    struct EvGuard { std::vector<hipEvent_t> &e; ~EvGuard() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); } } evg{evs};
    for (int k = 0; k < K; k++) HIPCHK(hipEventCreateWithFlags(&evs[(size_t)k], hipEventDisableTiming));
    std::atomic<int> uploaded{0};
    std::atomic<int> up_rc{0};
    auto a_begin = [&](int64_t p) { return p < n_pairs ? sp[(size_t)p].a_off : bases_len; };
    auto b_begin = [&](int64_t p) { return p < n_pairs ? sp[(size_t)p].b_off : bases_len + bases2_len; };
    auto upload = [&]() { // (K == 1: everything; runs on the calling thread then)
        uint8_t *dst = reinterpret_cast<uint8_t *>(c.in_a.p);
        for (int k = 0; k < K; k++) {
            bool ok = true;
            if (K == 1) {
                if (bases_len) ok = ok && hipMemcpyAsync(dst, bases, (size_t)bases_len, hipMemcpyHostToDevice, c.s_in) == hipSuccess;
                if (bases2_len) ok = ok && hipMemcpyAsync(dst + bases_len, bases2, (size_t)bases2_len, hipMemcpyHostToDevice, c.s_in) == hipSuccess;
            } else {
                const int64_t a0 = a_begin(kb[(size_t)k]), a1 = a_begin(kb[(size_t)k + 1]), b0 = b_begin(kb[(size_t)k]), b1 = b_begin(kb[(size_t)k + 1]);
                if (a1 > a0) ok = ok && hipMemcpyAsync(dst + a0, bases + a0, (size_t)(a1 - a0), hipMemcpyHostToDevice, c.s_in) == hipSuccess;
                if (b1 > b0) ok = ok && hipMemcpyAsync(dst + b0, bases2 + (b0 - bases_len), (size_t)(b1 - b0), hipMemcpyHostToDevice, c.s_in) == hipSuccess;
            }
            ok = ok && hipEventRecord(evs[(size_t)k], c.s_in) == hipSuccess;
            if (!ok) up_rc.store(1);
            uploaded.store(k + 1, std::memory_order_release);
        }
    };
    std::thread uploader;
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};
    if (K > 1) {
        const int dev = c.device;
        uploader = std::thread([&, dev]() { if (hipSetDevice(dev) != hipSuccess) { up_rc.store(1); uploaded.store(K, std::memory_order_release); return; } upload(); });
    } else upload();
    int64_t cap = std::max<int64_t>(std::min<int64_t>(worst, std::max<int64_t>((int64_t)1 << 20, 64 * n_pairs)), 1);
    if ((rc = c.out_ops.ensure((size_t)cap * sizeof(gnx_cigar)))) return rc;
    int64_t total = 0;
    gnx_timing tsum = {};
    bool redo = false; // a sub-batch did not fit the CIGAR buffer: once everything is uploaded, the whole batch runs again as one
    const ScorePair *spd0 = reinterpret_cast<const ScorePair *>(c.sc_pairs.p);
    const uint8_t *bd = reinterpret_cast<const uint8_t *>(c.in_a.p);
    int *sm = reinterpret_cast<int *>(c.sc_mat.p), *se = reinterpret_cast<int *>(c.sc_err.p);
    // groups with chunk <= 4: column profiles (score_profiles_kernel + score_matrix_groups_kernel)
    // (int32 sums: members(A) x members(B) x max|score| must fit; int16 counts)
    const bool use_prof = groups && chunk <= 4 && max_nseq < 30000 && max_nseq * max_nseq * std::max<int64_t>(smax, 1) < ((int64_t)1 << 31) && !getenv("GNX_SCORE_GENERIC");
    if (use_prof) {
        if ((rc = c.sc_prof_a.ensure((size_t)std::max<int64_t>(prof_a, 1) * sizeof(ColProfA)))) return rc;
        if ((rc = c.sc_prof_b.ensure((size_t)std::max<int64_t>(prof_b, 1) * sizeof(ColProfB)))) return rc;
    }
    auto score_matrices = [&](int64_t p0, int64_t p1) {
        for (int64_t b = p0; b < p1; b += 32768) {
            const unsigned ny = (unsigned)std::min<int64_t>(32768, p1 - b);
            const unsigned nx = (unsigned)std::min<int64_t>((maxcols + 3) / 4, 1024);
            const ScorePair *spd = spd0 + b;
            if (use_prof) {
                const ColProfA *dpa = reinterpret_cast<const ColProfA *>(c.sc_prof_a.p);
                const ColProfB *dpb = reinterpret_cast<const ColProfB *>(c.sc_prof_b.p);
                hipLaunchKernelGGL(score_profiles_kernel, dim3((unsigned)std::min<int64_t>((maxrows + maxcols) * chunk / 256 + 1, 512), ny), dim3(256), 0, st, spd, bd, kp0,
                                   reinterpret_cast<ColProfA *>(c.sc_prof_a.p), reinterpret_cast<ColProfB *>(c.sc_prof_b.p));
                const dim3 grid((unsigned)std::min<int64_t>((maxcols + 63) / 64, 1024), ny), blk(64, 4);
#define GNX_SMG(S, C) hipLaunchKernelGGL((score_matrix_groups_kernel<S, C>), grid, blk, 0, st, spd, kp0, (int)bias4, dpa, dpb, sm, se)
                if (s16) { if (chunk == 1) GNX_SMG(true, 1); else if (chunk == 2) GNX_SMG(true, 2); else if (chunk == 3) GNX_SMG(true, 3); else GNX_SMG(true, 4); }
                else { if (chunk == 1) GNX_SMG(false, 1); else if (chunk == 2) GNX_SMG(false, 2); else if (chunk == 3) GNX_SMG(false, 3); else GNX_SMG(false, 4); }
#undef GNX_SMG
                continue;
            }
            if (!groups && chunk <= 4 && maxrows * chunk < ((int64_t)1 << 30) && maxcols * chunk < ((int64_t)1 << 30) && !getenv("GNX_SCORE_GENERIC")) {
                // a block walks ~16 column quads, so that the rows' bases it keeps in registers are loaded once per 64 columns
                const dim3 grid((unsigned)std::min<int64_t>((maxcols + 63) / 64, 1024), ny), blk(64, 4);
#define GNX_SMP(S, C) hipLaunchKernelGGL((score_matrix_pairs_kernel<S, C>), grid, blk, 0, st, spd, bd, kp0, (int)bias4, sm, se)
                if (s16) { if (chunk == 1) GNX_SMP(true, 1); else if (chunk == 2) GNX_SMP(true, 2); else if (chunk == 3) GNX_SMP(true, 3); else GNX_SMP(true, 4); }
                else { if (chunk == 1) GNX_SMP(false, 1); else if (chunk == 2) GNX_SMP(false, 2); else if (chunk == 3) GNX_SMP(false, 3); else GNX_SMP(false, 4); }
#undef GNX_SMP
            } else {
                auto ksm = s16 ? score_matrix_kernel<true> : score_matrix_kernel<false>;
                hipLaunchKernelGGL(ksm, dim3(nx, ny), dim3(64, 4), 0, st, spd, bd, kp0, (int)chunk, groups ? 1 : 0, (int)bias4, sm, se);
            }
        }
    };
    double t_scores = 0;
    for (int k = 0; k < K; k++) {
        while (uploaded.load(std::memory_order_acquire) <= k) std::this_thread::yield();
        if (up_rc.load()) { set_err("upload of the bases failed%s", ""); return GNX_EDEVICE; }
        HIPCHK(hipStreamWaitEvent(st, evs[(size_t)k], 0));
        const int64_t p0 = kb[(size_t)k], p1 = kb[(size_t)k + 1], cnt = p1 - p0;
        score_matrices(p0, p1);
        HIPCHK(hipGetLastError());
        if (K == 1) { // one batch: errors of the score kernels are reported before the DP runs (sub-batches: after it; bad bases were read as 'A')
            int f[4] = {0, 0, 0, 0};
            HIPCHK(hipMemcpyAsync(f, c.sc_err.p, 16, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (f[0] & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
            if (f[0] & 16) { set_err("scoreColumnMatch over gap-only columns: the reference panics (integer divide by zero)%s", ""); return GNX_EDIVZERO; }
        }
        if (k == 0) t_scores = ms_since(t_begin);
        if (redo || cnt == 0) continue;
        int64_t tot = 0;
        rc = run_device(&prm2, cnt, nullptr, nullptr, nullptr, nullptr, hn.data() + p0, hm.data() + p0, (int64_t *)c.out_score.p + p0, (gnx_cigar *)c.out_ops.p + total, cap - total,
                        (int64_t *)c.out_off.p + p0, &tot, st, reinterpret_cast<const int *>(c.sc_mat.p), hso.data() + p0, 0, nullptr, false, s16);
        if (rc == GNX_ECAPACITY) { redo = true; continue; }
        if (rc) return rc;
        tsum.fill_ms += c.timing.fill_ms; tsum.traceback_ms += c.timing.traceback_ms; tsum.total_ms += c.timing.total_ms; tsum.cells += c.timing.cells;
        tsum.n_launches += c.timing.n_launches; tsum.trace_bytes += c.timing.trace_bytes; tsum.dominant_ms += c.timing.dominant_ms; tsum.dominant_launches += c.timing.dominant_launches;
        tsum.fast_path = std::max(tsum.fast_path, c.timing.fast_path);
        if (total > 0) { // offsets of a sub-batch start at 0
            hipLaunchKernelGGL(add_offset_kernel, dim3((unsigned)((cnt + 1 + 255) / 256)), dim3(256), 0, st, (int64_t *)c.out_off.p + p0, cnt + 1, total);
            HIPCHK(hipGetLastError());
        }
        total += tot;
    }
    int sflag[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(sflag, c.sc_err.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (sflag[0] & 1) { set_err("a base >= 5 was found: the reference would panic (index out of range)%s", ""); return GNX_EBASE; }
    if (sflag[0] & 16) { set_err("scoreColumnMatch over gap-only columns: the reference panics (integer divide by zero)%s", ""); return GNX_EDIVZERO; }
    if (redo || n_pairs == 0) { // (also the empty batch: run_device writes the single offset)
        total = 0; tsum = gnx_timing{};
        for (int attempt = 0; attempt < 8; attempt++) {
            if ((rc = c.out_ops.ensure((size_t)cap * sizeof(gnx_cigar)))) return rc;
            rc = run_device(&prm2, n_pairs, nullptr, nullptr, nullptr, nullptr, hn.data(), hm.data(), (int64_t *)c.out_score.p, (gnx_cigar *)c.out_ops.p, cap,
                            (int64_t *)c.out_off.p, &total, st, reinterpret_cast<const int *>(c.sc_mat.p), hso.data(), 0, nullptr, false, s16);
            if (rc != GNX_ECAPACITY) break;
            cap = std::max(total, cap + 1);
        }
        tsum = c.timing;
    } else c.timing = tsum;
    if (rc) return rc;
    const double t_dp = ms_since(t_begin);
    if (chunk > 1 && total > 0) hipLaunchKernelGGL(scale_runs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (gnx_cigar *)c.out_ops.p, total, chunk);
    struct HostArr { void *p; explicit HostArr(size_t b) : p(malloc(b)) {} ~HostArr() { free(p); } void *release() { void *q = p; p = nullptr; return q; } };
    HostArr ops_h((size_t)std::max<int64_t>(total, 1) * sizeof(gnx_cigar)), off_h((size_t)(n_pairs + 1) * 8); // freed on every early return below
    gnx_cigar *ops = (gnx_cigar *)ops_h.p;
    int64_t *off = (int64_t *)off_h.p;
    if (!ops || !off) { set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
    if (n_pairs) HIPCHK(hipMemcpyAsync(out_score, c.out_score.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(off, c.out_off.p, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, st));
    if (total) HIPCHK(hipMemcpyAsync(ops, c.out_ops.p, (size_t)total * sizeof(gnx_cigar), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *out_ops = (gnx_cigar *)ops_h.release(); *out_ops_off = (int64_t *)off_h.release();
    if (dbg) fprintf(stderr, "[gnx] scored batch of %lld pairs: upload + score matrices %.3f ms, DP (fill %.3f + traceback %.3f on the device) until %.3f ms, results on the host at %.3f ms\n",
                     (long long)n_pairs, t_scores, c.timing.fill_ms, c.timing.traceback_ms, t_dp, ms_since(t_begin));
    return GNX_OK;
}

int run_host_windows(const gnx_params *prm, int64_t n_pairs,
                     const uint8_t *a_buf, int64_t a_len, const int64_t *a_start, const int64_t *a_lens,
                     const uint8_t *b_buf, int64_t b_len, const int64_t *b_start, const int64_t *b_lens,
                     int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off,
                     int gsw = 0, int64_t *out_end_i = nullptr, int64_t *out_end_j = nullptr) {
    Ctx &c = g_ctx;
    if (!prm || n_pairs < 0 || !out_score || !out_ops || !out_ops_off || a_len < 0 || b_len < 0) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    if (gsw && (!out_end_i || !out_end_j)) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    if (n_pairs > 0 && (!a_start || !a_lens || !b_start || !b_lens)) { set_err("null window table%s", ""); return GNX_EINVAL; }
    for (int64_t p = 0; p < n_pairs; p++) {
        if (a_start[p] < 0 || a_lens[p] < 0 || a_start[p] + a_lens[p] > a_len || b_start[p] < 0 || b_lens[p] < 0 || b_start[p] + b_lens[p] > b_len) {
            set_err("window out of bounds at pair %s%lld", "", (long long)p); return GNX_EINVAL;
        }
    }
    int rc;
    hipStream_t st = c.own_stream;
    const size_t np = (size_t)std::max<int64_t>(n_pairs, 1);
    if ((rc = c.in_a.ensure((size_t)a_len + 16))) return rc;
    if ((rc = c.in_b.ensure((size_t)b_len + 16))) return rc;
    if ((rc = c.in_as.ensure(np * 8))) return rc;
    if ((rc = c.in_bs.ensure(np * 8))) return rc;
    if ((rc = c.out_score.ensure(np * 8))) return rc;
    if ((rc = c.out_off.ensure((np + 1) * 8))) return rc;
    if (gsw && (rc = c.out_end.ensure(np * 8))) return rc;
    if (a_len) HIPCHK(hipMemcpyAsync(c.in_a.p, a_buf, (size_t)a_len, hipMemcpyHostToDevice, st));
    if (b_len) HIPCHK(hipMemcpyAsync(c.in_b.p, b_buf, (size_t)b_len, hipMemcpyHostToDevice, st));
    if (n_pairs) {
        HIPCHK(hipMemcpyAsync(c.in_as.p, a_start, (size_t)n_pairs * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(c.in_bs.p, b_start, (size_t)n_pairs * 8, hipMemcpyHostToDevice, st));
    }
    // CIGAR capacity: exact worst case for small batches, a guess (+ one exact retry) for large ones
    int64_t worst = 0;
    for (int64_t p = 0; p < n_pairs; p++) worst += a_lens[p] + b_lens[p] + 1;
    int64_t cap = std::min<int64_t>(worst, std::max<int64_t>((int64_t)1 << 20, 64 * n_pairs));
    cap = std::max<int64_t>(cap, 1);
    int64_t total = 0;
    for (int attempt = 0; attempt < 8; attempt++) {
        if ((rc = c.out_ops.ensure((size_t)cap * sizeof(gnx_cigar)))) return rc;
        rc = run_device(prm, n_pairs, (const uint8_t *)c.in_a.p, (const int64_t *)c.in_as.p, (const uint8_t *)c.in_b.p, (const int64_t *)c.in_bs.p,
                        a_lens, b_lens, (int64_t *)c.out_score.p, (gnx_cigar *)c.out_ops.p, cap, (int64_t *)c.out_off.p, &total, st,
                        nullptr, nullptr, gsw, gsw ? (int2 *)c.out_end.p : nullptr);
        if (rc != GNX_ECAPACITY) break;
        cap = std::max(total, cap + 1);
    }
    if (rc) return rc;
    struct HostArr { void *p; explicit HostArr(size_t b) : p(malloc(b)) {} ~HostArr() { free(p); } void *release() { void *q = p; p = nullptr; return q; } };
    HostArr ops_h((size_t)std::max<int64_t>(total, 1) * sizeof(gnx_cigar)), off_h((size_t)(n_pairs + 1) * 8); // freed on every early return below
    gnx_cigar *ops = (gnx_cigar *)ops_h.p;
    int64_t *off = (int64_t *)off_h.p;
    if (!ops || !off) { set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
    if (n_pairs) HIPCHK(hipMemcpyAsync(out_score, c.out_score.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(off, c.out_off.p, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, st));
    if (total) HIPCHK(hipMemcpyAsync(ops, c.out_ops.p, (size_t)total * sizeof(gnx_cigar), hipMemcpyDeviceToHost, st));
    std::vector<int2> ends;
    if (gsw && n_pairs) { ends.resize((size_t)n_pairs); HIPCHK(hipMemcpyAsync(ends.data(), c.out_end.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st)); }
    HIPCHK(hipStreamSynchronize(st));
    for (int64_t p = 0; gsw && p < n_pairs; p++) { out_end_i[p] = ends[(size_t)p].x; out_end_j[p] = ends[(size_t)p].y; }
    *out_ops = (gnx_cigar *)ops_h.release(); *out_ops_off = (int64_t *)off_h.release();
    return GNX_OK;
}

} // namespace

#include "gnx_host.hip.h"

// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
extern "C" {

int gnx_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int gnx_init(int device, int64_t workspace_bytes) {
    std::lock_guard<std::mutex> api(g_api_mu);
    Ctx &c = ctx_at(0);
    CtxScope sc(c);
    g_err[0] = 0;
    if (c.inited) {
        if (device == c.device) { if (workspace_bytes > 0) c.ws_limit = workspace_bytes; return GNX_OK; }
        set_err("already bound to another device%s", ""); return GNX_EINVAL;
    }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { set_err("no HIP device available (this library has no CPU fallback)%s", ""); return GNX_EDEVICE; }
    if (device < 0 || device >= cnt) { set_err("device index out of range%s", ""); return GNX_EINVAL; }
    return init_ctx(c, device, workspace_bytes);
}

int gnx_init_devices(int n_devices, const int *devices, int64_t workspace_bytes) {
    std::lock_guard<std::mutex> api(g_api_mu);
    g_err[0] = 0;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { set_err("no HIP device available (this library has no CPU fallback)%s", ""); return GNX_EDEVICE; }
    if (n_devices <= 0) n_devices = cnt;
    if (n_devices > 64) { set_err("too many devices%s", ""); return GNX_EINVAL; }
    std::vector<int> devs((size_t)n_devices);
    bool distinct = true;
    for (int k = 0; k < n_devices; k++) {
        devs[(size_t)k] = devices ? devices[k] : k;
        if (devs[(size_t)k] < 0 || devs[(size_t)k] >= cnt) { set_err("device index out of range%s", ""); return GNX_EINVAL; }
        for (int q = 0; q < k; q++) if (devs[(size_t)q] == devs[(size_t)k]) distinct = false;
    }
    if (!g_rccl.comms.empty()) rccl_drop_comms(); // a previous set of devices: drop its communicators
    // (g_rccl_broken stays: after a RCCL failure the process uses peer copies until gnx_shutdown)
    for (int k = 0; k < n_devices; k++) {
        Ctx &c = ctx_at(k);
        CtxScope sc(c);
        if (c.inited) {
            if (c.device != devs[(size_t)k]) { set_err("context %s%lld is already bound to another device (gnx_shutdown first)", "", (long long)k); return GNX_EINVAL; }
            if (workspace_bytes > 0) c.ws_limit = workspace_bytes;
            continue;
        }
        int rc = init_ctx(c, devs[(size_t)k], workspace_bytes);
        if (rc) return rc;
    }
    g_nctx = n_devices;
    g_shared_dev = !distinct;
    // RCCL communicators, one per context (all in this process).  GNX_RCCL=0: plain peer copies; GNX_RCCL=1: also for one device
    // (a 1-rank communicator: every RCCL call of the flow still runs, which is what a 1-GPU box can check of the plumbing)
    const char *re = getenv("GNX_RCCL");
    const bool want = !g_rccl_broken && distinct && !(re && re[0] == '0') && (n_devices > 1 || (re && re[0] == '1'));
    if (want) {
        int rc = rccl_load();
        if (rc) return rc;
        g_rccl.comms.assign((size_t)n_devices, nullptr);
        ncclResult_t r = g_rccl.CommInitAll(g_rccl.comms.data(), n_devices, devs.data());
        if (r != ncclSuccess) { g_rccl.comms.clear(); set_err("ncclCommInitAll failed: %s", g_rccl.GetErrorString(r)); return GNX_EDEVICE; }
        (void)hipSetDevice(devs[0]);
    }
    return GNX_OK;
}

int gnx_n_devices(void) { std::lock_guard<std::mutex> api(g_api_mu); return g_nctx; }

void gnx_shutdown(void) {
    std::lock_guard<std::mutex> api(g_api_mu);
    rccl_drop_comms();
    int n;
    { std::lock_guard<std::mutex> lk(g_ctxs_mu); n = (int)g_ctxs.size(); }
    for (int k = 0; k < n; k++) {
        Ctx &c = ctx_at(k);
        CtxScope sc(c);
        if (!c.inited) continue;
        (void)hipSetDevice(c.device);
        (void)hipDeviceSynchronize();
        DevBuf *bufs[] = {&c.strip_map, &c.tb_scr, &c.tb_scr_off, &c.scan_tmp, &c.fp_redo, &c.fp_strag, &c.mx_idx, &c.mx_tab, &c.mx_score, &c.mx_off, &c.mx_ops, &c.fp_prog, &c.fp_tail, &c.fp_thcol, &c.fp_ttrace, &c.fp_rowi, &c.fp_ckpt, &c.fp_states, &c.fp_stage,
                          &c.fp_wplans[0], &c.fp_wplans[1], &c.fp_active[0], &c.fp_active[1], &c.trace, &c.hcol, &c.rowbuf, &c.dcol, &c.plans, &c.nops, &c.misc, &c.in_a, &c.in_b,
                          &c.in_as, &c.in_al, &c.in_bs, &c.in_bl, &c.out_score, &c.out_off, &c.out_ops, &c.out_end, &c.sc_pairs, &c.sc_mat, &c.sc_err,
                          &c.pin_a[0], &c.pin_a[1], &c.pin_as[0], &c.pin_as[1], &c.pin_b[0], &c.pin_b[1], &c.pin_bs[0], &c.pin_bs[1], &c.res_score, &c.res_off, &c.res_ops,
                          &c.ref, &c.ref_flag, &c.ref_rank, &c.ref_exc, &c.unpk_b, &c.unpk_off, &c.cl_bases, &c.sc_prof_a, &c.sc_prof_b, &c.mega_rows, &c.mega_state, &c.farm, &c.mega_arena, &c.gat_score, &c.gat_off, &c.gat_ops, &c.sd_keys, &c.sd_locs, &c.sd_nodes, &c.sd_node_off, &c.sd_word_off, &c.sd_words,
                          &c.sd_tmp[0], &c.sd_tmp[1], &c.sd_tmp[2], &c.sd_tmp[3], &c.sd_tmp[4], &c.sd_tmp[5], &c.sd_tmp[6], &c.sd_tmp[7]};
        for (DevBuf *b : bufs) b->release();
        PinBuf *pins[] = {&c.h_plans, &c.st_a[0], &c.st_a[1], &c.st_as[0], &c.st_as[1], &c.st_b[0], &c.st_b[1], &c.st_bs[0], &c.st_bs[1]};
        for (PinBuf *b : pins) b->release();
        c.fpc_ptr = nullptr; c.ref_len = -1; c.sd_n = -1;
        for (int i = 0; i < 8; i++) if (c.ev[i]) { (void)hipEventDestroy(c.ev[i]); c.ev[i] = nullptr; }
        for (int i = 0; i < 2; i++) if (c.ev_in[i]) { (void)hipEventDestroy(c.ev_in[i]); c.ev_in[i] = nullptr; }
        if (c.own_stream) { (void)hipStreamDestroy(c.own_stream); c.own_stream = nullptr; }
        if (c.s_in) { (void)hipStreamDestroy(c.s_in); c.s_in = nullptr; }
        c.inited = false;
        c.ws_limit = 0;
        c.device = -1;
    }
    g_pool.drain();
    g_nctx = 1;
    g_shared_dev = false;
    g_rccl_broken = false;
}

/* the calling thread's last error; if this thread has none, the most recent error of the process (a cgo caller may fetch the
 * text on another OS thread than the one that ran the failing call) */
const char *gnx_last_error(void) {
    if (g_err[0]) return g_err;
    static thread_local char copy[512];
    std::lock_guard<std::mutex> lk(g_lasterr_mu);
    memcpy(copy, g_lasterr, sizeof(copy));
    return copy;
}

void gnx_free(void *p) { if (p && !g_pool.put(p)) free(p); }

int gnx_set_reference(const uint8_t *ref, int64_t len) {
    std::lock_guard<std::mutex> api(g_api_mu);
    g_err[0] = 0;
    if (len < 0 || (len > 0 && !ref)) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    return set_reference_packed(ref, len, 0);
}

int gnx_set_reference_synthetic(int64_t len, uint64_t seed) {
    std::lock_guard<std::mutex> api(g_api_mu);
    g_err[0] = 0;
    if (len < 0) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    return set_reference_packed(nullptr, len, seed);
}

int gnx_reference_info(int64_t *out_bases, int64_t *out_device_bytes, int64_t *out_exception_blocks) {
    std::lock_guard<std::mutex> api(g_api_mu);
    Ctx &c = ctx_at(0);
    CtxScope sc(c);
    if (c.ref_len < 0) { set_err("no resident reference%s", ""); return GNX_EINVAL; }
    if (out_bases) *out_bases = c.ref_len;
    if (out_device_bytes) *out_device_bytes = (int64_t)(ref_words(c.ref_len) * 4 + ref_flagwords(c.ref_len) * 12 + (size_t)c.ref_nexc * 16);
    if (out_exception_blocks) *out_exception_blocks = c.ref_nexc;
    return GNX_OK;
}

int gnx_align_batch_by_offset(const gnx_params *p, int64_t n_pairs, const uint8_t *alpha_cat, const int64_t *alpha_off,
                              const int64_t *ref_start, const int64_t *ref_len,
                              int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> api(g_api_mu);
    g_err[0] = 0;
    if (n_pairs < 0 || !alpha_off || (n_pairs > 0 && (!ref_start || !ref_len))) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    std::vector<int64_t> al((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) al[(size_t)q] = alpha_off[q + 1] - alpha_off[q];
    return run_host_sharded(p, n_pairs, alpha_cat, alpha_off[n_pairs], alpha_off, al.data(), nullptr, 0, ref_start, ref_len, out_score, out_ops, out_ops_off);
}

int gnx_align_batch_windows(const gnx_params *p, int64_t n_pairs,
                            const uint8_t *alpha_buf, int64_t alpha_buf_len, const int64_t *alpha_start, const int64_t *alpha_len,
                            const uint8_t *beta_buf, int64_t beta_buf_len, const int64_t *beta_start, const int64_t *beta_len,
                            int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> api(g_api_mu);
    g_err[0] = 0;
    static const uint8_t none = 0;
    return run_host_sharded(p, n_pairs, alpha_buf ? alpha_buf : &none, alpha_buf_len, alpha_start, alpha_len, beta_buf ? beta_buf : &none, beta_buf_len, beta_start, beta_len,
                            out_score, out_ops, out_ops_off);
}

int gnx_align_batch(const gnx_params *p, int64_t n_pairs, const uint8_t *alpha_cat, const int64_t *alpha_off,
                    const uint8_t *beta_cat, const int64_t *beta_off, int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    if (n_pairs < 0 || !alpha_off || !beta_off) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    std::vector<int64_t> al((size_t)n_pairs), bl((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) { al[(size_t)q] = alpha_off[q + 1] - alpha_off[q]; bl[(size_t)q] = beta_off[q + 1] - beta_off[q]; }
    return gnx_align_batch_windows(p, n_pairs, alpha_cat, alpha_off[n_pairs], alpha_off, al.data(), beta_cat, beta_off[n_pairs], beta_off, bl.data(),
                                   out_score, out_ops, out_ops_off);
}

int gnx_gsw_extend_batch(int side, const int64_t *scores, int64_t gap_pen, int64_t n_pairs,
                         const uint8_t *alpha_cat, const int64_t *alpha_off, const uint8_t *beta_cat, const int64_t *beta_off,
                         int64_t *out_score, int64_t *out_end_i, int64_t *out_end_j, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> api(g_api_mu);
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    if ((side != GNX_GSW_LEFT && side != GNX_GSW_RIGHT) || !scores || n_pairs < 0 || !alpha_off || !beta_off) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    int rc = ensure_init();
    if (rc) return rc;
    gnx_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.mode = GNX_CONST_GAP_HIGHMEM;
    for (int x = 0; x < 25; x++) prm.scores[x] = scores[x];
    prm.gap_open = gap_pen; prm.checkersize_i = 10000; prm.checkersize_j = 10000;
    std::vector<int64_t> al((size_t)n_pairs), bl((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) { al[(size_t)q] = alpha_off[q + 1] - alpha_off[q]; bl[(size_t)q] = beta_off[q + 1] - beta_off[q]; }
    return run_host_windows(&prm, n_pairs, alpha_cat, alpha_off[n_pairs], alpha_off, al.data(), beta_cat, beta_off[n_pairs], beta_off, bl.data(),
                            out_score, out_ops, out_ops_off, side == GNX_GSW_LEFT ? 1 : 2, out_end_i, out_end_j);
}

/* align.AffineGap / ConstGap / ... for ONE pair (the body of every Go-signature function of the shim).
 *
 * Concurrency (the reference's house pattern is a pool of goroutines that each call align.* in a loop,
 * genomeGraph/routines.go:12-65): calls from many threads are COMBINED.  Every caller queues its request and then takes the API lock;
 * whoever holds the lock aligns, as ONE batch, every queued request that has the same parameters as its own (the others were queued
 * while the previous batch was on the device), and hands each its result.  A caller that finds its request done when it gets the lock
 * just returns.  One thread alone pays nothing (a batch of one goes straight through); 16 threads get batches of ~15 pairs and the
 * device's batch throughput instead of 16 serial launches.  A failure inside a combined batch (a bad base in one pair) is re-run pair
 * by pair, so that every caller sees the return code and message of its own pair only. */
namespace {
struct PairReq {
    const gnx_params *p; const uint8_t *a; int64_t n; const uint8_t *b; int64_t m;
    int64_t score = 0; gnx_cigar *ops = nullptr; int64_t n_ops = 0;
    int rc = GNX_OK; char err[512] = ""; bool done = false;
};
std::mutex g_pq_mu;
std::vector<PairReq *> g_pq;
std::atomic<int64_t> g_pq_batches{0}, g_pq_pairs{0}; // combined batches run / pairs in them (gnx_debug_counter(1 / 2))

int run_pair_alone(PairReq &r) {
    const int64_t zero = 0;
    int64_t *off = nullptr;
    static const uint8_t none = 0;
    g_err[0] = 0;
    int rc = run_host_sharded(r.p, 1, r.a ? r.a : &none, r.n, &zero, &r.n, r.b ? r.b : &none, r.m, &zero, &r.m, &r.score, &r.ops, &off);
    if (rc == GNX_OK) { r.n_ops = off[1]; gnx_free(off); }
    else memcpy(r.err, g_err, sizeof(r.err));
    return rc;
}
// aligns the requests of `batch` (equal parameters) -- as ONE device batch when there are several; the caller holds the API lock
void run_pair_batch(const gnx_params *p, std::vector<PairReq *> &batch) {
    if (batch.size() == 1) batch[0]->rc = run_pair_alone(*batch[0]);
    else if (batch.size() > 1) {
        const int64_t nb = (int64_t)batch.size();
        std::vector<int64_t> as((size_t)nb), al((size_t)nb), bs((size_t)nb), bl((size_t)nb), sc((size_t)nb);
        int64_t ta = 0, tb = 0;
        for (int64_t k = 0; k < nb; k++) { as[(size_t)k] = ta; al[(size_t)k] = batch[(size_t)k]->n; ta += batch[(size_t)k]->n; bs[(size_t)k] = tb; bl[(size_t)k] = batch[(size_t)k]->m; tb += batch[(size_t)k]->m; }
        std::vector<uint8_t> ca((size_t)ta + 1), cb((size_t)tb + 1);
        for (int64_t k = 0; k < nb; k++) {
            if (batch[(size_t)k]->n) memcpy(ca.data() + as[(size_t)k], batch[(size_t)k]->a, (size_t)batch[(size_t)k]->n);
            if (batch[(size_t)k]->m) memcpy(cb.data() + bs[(size_t)k], batch[(size_t)k]->b, (size_t)batch[(size_t)k]->m);
        }
        gnx_cigar *ops = nullptr;
        int64_t *off = nullptr;
        g_err[0] = 0;
        int rc = run_host_sharded(p, nb, ca.data(), ta, as.data(), al.data(), cb.data(), tb, bs.data(), bl.data(), sc.data(), &ops, &off);
        if (rc == GNX_OK) {
            for (int64_t k = 0; k < nb && rc == GNX_OK; k++) {
                PairReq &r = *batch[(size_t)k];
                r.score = sc[(size_t)k]; r.n_ops = off[k + 1] - off[k];
                r.ops = (gnx_cigar *)malloc((size_t)std::max<int64_t>(r.n_ops, 1) * sizeof(gnx_cigar)); // (gnx_free: not from the pinned pool -> free())
                if (!r.ops) { rc = GNX_ENOMEM; break; }
                memcpy(r.ops, ops + off[k], (size_t)r.n_ops * sizeof(gnx_cigar));
                r.rc = GNX_OK;
            }
            gnx_free(ops); gnx_free(off);
            g_pq_batches++; g_pq_pairs += nb;
        }
        if (rc != GNX_OK) { // one pair's error must not become its neighbours': every request on its own
            for (PairReq *r : batch) { if (r->ops) { free(r->ops); r->ops = nullptr; } r->rc = run_pair_alone(*r); }
        }
    }
}
std::condition_variable g_pq_cv;        // "a batch is done": followers wait here
std::condition_variable g_pq_cv_arrive; // "a request was queued": only the collecting combiner waits here (no herd of followers woken per arrival)
bool g_pq_leader = false; // a combined batch is being collected / aligned (guarded by g_pq_mu)
size_t g_pq_prev_batch = 0; // pairs in the batch before (guarded by g_pq_mu)
} // namespace

int gnx_align_pair(const gnx_params *p, const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m,
                   int64_t *out_score, gnx_cigar **out_ops, int64_t *out_n_ops) {
    if (!p || !out_score || !out_ops || !out_n_ops || n < 0 || m < 0) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    PairReq self;
    self.p = p; self.a = alpha; self.n = n; self.b = beta; self.m = m;
    std::unique_lock<std::mutex> lk(g_pq_mu);
    try { g_pq.push_back(&self); } catch (...) { set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
    if (g_pq_leader) g_pq_cv_arrive.notify_one(); // (a combiner may be collecting: it counts arrivals)
    while (!self.done) {
        if (g_pq_leader) { g_pq_cv.wait(lk); continue; } // a batch is on the device: this request rides in the next one
        // become the combiner: everything queued with this request's parameters is one batch (requests with other parameters stay
        // queued; one of their owners combines them next)
        g_pq_leader = true;
        if (g_pq.size() > 1 || g_pq_prev_batch > 1) {
            // Other threads are calling too: the callers of the batch that has just been handed out are on their way back with their next
            // pair.  Collect until nobody has arrived for ~25 us (at most 200 us): without this a pool of 16 threads settles into two groups of 8
            // that take turns (measured: 8 pairs per batch, 6.5 x the serial rate); one thread alone never waits.
            // ... and no longer than it takes the callers of the batch before to be back: their number is the best guess of how many threads are calling
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
            size_t seen = g_pq.size();
            while (seen < g_pq_prev_batch && std::chrono::steady_clock::now() < t_end) {
                g_pq_cv_arrive.wait_for(lk, std::chrono::microseconds(25));
                if (g_pq.size() == seen) break;
                seen = g_pq.size();
            }
        }
        //  the combiner's section must not leave by an exception: the requests it took point at other callers' stack frames,
        // and they wait for `done`.  Whatever is thrown (std::bad_alloc of the batch vectors), every request taken so far -- at the least
        // this caller's own -- is answered with GNX_ENOMEM, the leader flag is given back and everybody is woken.
        std::vector<PairReq *> batch;
        auto same_params = [](const gnx_params *x, const gnx_params *y) { // (field by field: _reserved and padding do not keep requests apart)
            return x->mode == y->mode && x->gap_open == y->gap_open && x->gap_extend == y->gap_extend && x->checkersize_i == y->checkersize_i &&
                   x->checkersize_j == y->checkersize_j && memcmp(x->scores, y->scores, sizeof(x->scores)) == 0;
        };
        bool failed = false;
        try {
            batch.reserve(g_pq.size());
            for (size_t k = 0; k < g_pq.size();) {
                if (g_pq[k] == &self || same_params(g_pq[k]->p, p)) { batch.push_back(g_pq[k]); g_pq.erase(g_pq.begin() + (long)k); }
                else k++;
            }
        } catch (...) { failed = true; }
        lk.unlock();
        if (!failed) {
            try {
                std::lock_guard<std::mutex> api(g_api_mu);
                run_pair_batch(p, batch);
            } catch (...) { failed = true; }
        }
        lk.lock();
        if (failed) {
            bool mine = false;
            for (PairReq *r : batch) { if (r == &self) mine = true; if (r->ops) { free(r->ops); r->ops = nullptr; } r->rc = GNX_ENOMEM; snprintf(r->err, sizeof(r->err), "host allocation failed while combining concurrent gnx_align_pair calls"); }
            if (!mine) { // (the reserve itself failed: this request is still queued)
                for (size_t k = 0; k < g_pq.size(); k++) if (g_pq[k] == &self) { g_pq.erase(g_pq.begin() + (long)k); break; }
                self.rc = GNX_ENOMEM; snprintf(self.err, sizeof(self.err), "host allocation failed while combining concurrent gnx_align_pair calls"); self.done = true;
            }
        }
        for (PairReq *r : batch) r->done = true;
        g_pq_prev_batch = batch.size();
        g_pq_leader = false;
        g_pq_cv.notify_all();
    }
    lk.unlock();
    // (self.done is set: either by this thread's batch or by the batch of the thread that held the lock before)
    if (self.rc != GNX_OK) { memcpy(g_err, self.err, sizeof(g_err)); publish_err(); return self.rc; }
    *out_score = self.score; *out_ops = self.ops; *out_n_ops = self.n_ops;
    return GNX_OK;
}

int gnx_align_batch_device(const gnx_params *p, int64_t n_pairs,
                           const uint8_t *d_alpha_buf, const int64_t *d_alpha_start, const int64_t *d_alpha_len,
                           const uint8_t *d_beta_buf, const int64_t *d_beta_start, const int64_t *d_beta_len,
                           const int64_t *h_alpha_len, const int64_t *h_beta_len,
                           int64_t *d_score, gnx_cigar *d_ops, int64_t ops_capacity, int64_t *d_ops_off,
                           int64_t *out_total_ops, void *stream) {
    (void)d_alpha_len; (void)d_beta_len; // lengths are taken from the host copies (planning needs them anyway)
    std::lock_guard<std::mutex> api(g_api_mu);
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    int rc = ensure_init();
    if (rc) return rc;
    if (!p || n_pairs < 0 || !d_score || !d_ops_off || ops_capacity < 0 || (n_pairs > 0 && (!h_alpha_len || !h_beta_len || !d_alpha_start || !d_beta_start))) {
        set_err("bad argument%s", ""); return GNX_EINVAL;
    }
    return run_device(p, n_pairs, d_alpha_buf, d_alpha_start, d_beta_buf, d_beta_start, h_alpha_len, h_beta_len,
                      d_score, d_ops, ops_capacity, d_ops_off, out_total_ops, (hipStream_t)stream);
}

int gnx_affine_gap_chunk_batch(const gnx_params *p, int64_t chunk_size, int64_t n_pairs,
                               const uint8_t *alpha_cat, const int64_t *alpha_off, const uint8_t *beta_cat, const int64_t *beta_off,
                               int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> api(g_api_mu);
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    int rc = ensure_init();
    if (rc) return rc;
    if (n_pairs < 0 || chunk_size < 1 || (n_pairs > 0 && (!alpha_off || !beta_off))) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    const int64_t la = n_pairs ? alpha_off[n_pairs] : 0, lb = n_pairs ? beta_off[n_pairs] : 0;
    std::vector<ScorePair> sp((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) {
        const int64_t n = alpha_off[q + 1] - alpha_off[q], m = beta_off[q + 1] - beta_off[q];
        if (n < 0 || m < 0 || n > 0x3fffffff || m > 0x3fffffff) { set_err("bad sequence length at pair %s%lld", "", (long long)q); return GNX_EINVAL; }
        if (n % chunk_size != 0 || m % chunk_size != 0) { // log.Fatalf in the reference (affineGap_highMem.go:229-234)
            set_err("pair %s%lld: sequence length is not a multiple of the chunk size", "", (long long)q); return GNX_EINVAL;
        }
        ScorePair &s = sp[(size_t)q];
        s.a_off = alpha_off[q]; s.b_off = la + beta_off[q]; s.a_nseq = 1; s.b_nseq = 1; s.a_len = (int32_t)n; s.b_len = (int32_t)m;
        s.nc = (int32_t)(n / chunk_size); s.mc = (int32_t)(m / chunk_size); s.s_off = 0; s.s_pitch = 0;
    }
    return run_host_scored(p, chunk_size, false, n_pairs, sp, alpha_cat, la, out_score, out_ops, out_ops_off, beta_cat, lb);
}

int gnx_multiple_affine_gap_batch(const gnx_params *p, int64_t chunk_size, int64_t n_groups, const uint8_t *group_bases,
                                  const int64_t *group_off, const int32_t *group_nseq, const int64_t *group_len,
                                  int64_t n_pairs, const int32_t *pair_a, const int32_t *pair_b,
                                  int64_t *out_score, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::lock_guard<std::mutex> api(g_api_mu);
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    int rc = ensure_init();
    if (rc) return rc;
    if (n_pairs < 0 || n_groups < 0 || chunk_size < 1 || (n_groups > 0 && (!group_off || !group_nseq || !group_len)) || (n_pairs > 0 && (!pair_a || !pair_b))) {
        set_err("bad argument%s", ""); return GNX_EINVAL;
    }
    for (int64_t g = 0; g < n_groups; g++) {
        if (group_nseq[g] < 1 || group_len[g] < 0 || group_len[g] > 0x3fffffff || group_off[g + 1] - group_off[g] != (int64_t)group_nseq[g] * group_len[g]) {
            set_err("group %s%lld: bases do not match nseq x len", "", (long long)g); return GNX_EINVAL;
        }
    }
    std::vector<ScorePair> sp((size_t)n_pairs);
    for (int64_t q = 0; q < n_pairs; q++) {
        const int32_t a = pair_a[q], b = pair_b[q];
        if (a < 0 || b < 0 || a >= n_groups || b >= n_groups) { set_err("pair %s%lld: group index out of range", "", (long long)q); return GNX_EINVAL; }
        if (group_len[a] % chunk_size != 0 || group_len[b] % chunk_size != 0) { // log.Fatalf (affineGap_highMem.go:310-315)
            set_err("pair %s%lld: alignment length is not a multiple of the chunk size", "", (long long)q); return GNX_EINVAL;
        }
        ScorePair &s = sp[(size_t)q];
        s.a_off = group_off[a]; s.b_off = group_off[b]; s.a_nseq = group_nseq[a]; s.b_nseq = group_nseq[b];
        s.a_len = (int32_t)group_len[a]; s.b_len = (int32_t)group_len[b];
        s.nc = (int32_t)(group_len[a] / chunk_size); s.mc = (int32_t)(group_len[b] / chunk_size); s.s_off = 0; s.s_pitch = 0;
    }
    return run_host_scored(p, chunk_size, true, n_pairs, sp, group_bases, n_groups ? group_off[n_groups] : 0, out_score, out_ops, out_ops_off);
}

/* ---- "next" row N4: seed index and seed search of the graph aligner (see seed_kernels.hip.h) ---- */
int gnx_seed_index_build(const uint8_t *node_cat, const int64_t *node_off, int64_t n_nodes, int seed_len, int seed_step,
                         uint64_t **out_keys, uint64_t **out_locs, int64_t *out_n) {
    std::lock_guard<std::mutex> api(g_api_mu);
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    if (!node_off || n_nodes < 0 || n_nodes > 0x7ffffff0 || seed_len < 2 || seed_len > 32 || seed_step < 1 || !out_keys || !out_locs || !out_n) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    int rc = ensure_init();
    if (rc) return rc;
    Ctx &c = g_ctx;
    hipStream_t st = c.own_stream;
    std::vector<int64_t> slot_off((size_t)n_nodes + 1, 0);
    for (int64_t k = 0; k < n_nodes; k++) {
        const int64_t L = node_off[k + 1] - node_off[k];
        if (L < 0 || L > 0x7fffffff) { set_err("bad node length at node %s%lld", "", (long long)k); return GNX_EINVAL; }
        slot_off[(size_t)k + 1] = slot_off[(size_t)k] + (L >= seed_len ? (L - seed_len) / seed_step + 1 : 0);
    }
    const int64_t n_slots = slot_off[(size_t)n_nodes], total = n_nodes ? node_off[n_nodes] : 0;
    *out_keys = nullptr; *out_locs = nullptr; *out_n = 0;
    if (n_slots == 0) return GNX_OK;
    DevBuf *t = c.sd_tmp;
    c.sd_n = -1; // the build uploads its nodes into the buffers of the resident index: gnx_seed_find_batch needs a new gnx_seed_index_set
    if ((rc = c.sd_nodes.ensure((size_t)total + 64))) return rc;
    if ((rc = c.sd_node_off.ensure((size_t)(n_nodes + 1) * 8))) return rc;
    if ((rc = t[0].ensure((size_t)(n_nodes + 1) * 8))) return rc;   // slot_off
    if ((rc = t[1].ensure((size_t)n_slots * 8))) return rc;          // keys (all slots)
    if ((rc = t[2].ensure((size_t)n_slots * 8))) return rc;          // locs
    if ((rc = t[3].ensure((size_t)n_slots * 4))) return rc;          // flags
    if ((rc = t[4].ensure((size_t)(n_slots + 1) * 8))) return rc;    // flags as int64, then offsets
    if ((rc = t[5].ensure((size_t)(n_slots + 1) * 8))) return rc;
    if ((rc = c.misc.ensure(64))) return rc;
    HIPCHK(hipMemcpyAsync(c.sd_nodes.p, node_cat, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c.sd_node_off.p, node_off, (size_t)(n_nodes + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(t[0].p, slot_off.data(), (size_t)(n_nodes + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, st));
    const unsigned gb = (unsigned)((n_slots + 255) / 256);
    hipLaunchKernelGGL(seed_index_count_kernel, dim3(gb), dim3(256), 0, st, (const uint8_t *)c.sd_nodes.p, (const int64_t *)c.sd_node_off.p, (int)n_nodes, (const int64_t *)t[0].p,
                       seed_len, seed_step, (uint64_t *)t[1].p, (uint64_t *)t[2].p, (int *)t[3].p, n_slots);
    hipLaunchKernelGGL(flag_to_i64_kernel, dim3(gb), dim3(256), 0, st, (const int *)t[3].p, n_slots, (int64_t *)t[4].p);
    HIPCHK(hipGetLastError());
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    if (n_slots > 0x7ffffff0) { set_err("too many k-mer positions%s", ""); return GNX_EINVAL; }
    if ((rc = launch_scan((const int64_t *)t[4].p, (int)n_slots, (int64_t *)t[5].p, d_carry, st))) return rc;
    int64_t n_kmers = 0;
    HIPCHK(hipMemcpyAsync(&n_kmers, d_carry, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_kmers == 0) return GNX_OK;
    if ((rc = t[6].ensure((size_t)n_kmers * 16))) return rc; // compacted keys | locs
    if ((rc = t[7].ensure((size_t)n_kmers * 16))) return rc; // sorted keys | locs
    uint64_t *ck = (uint64_t *)t[6].p, *cl = ck + n_kmers, *sk = (uint64_t *)t[7].p, *sl = sk + n_kmers;
    hipLaunchKernelGGL(seed_index_compact_kernel, dim3(gb), dim3(256), 0, st, (const uint64_t *)t[1].p, (const uint64_t *)t[2].p, (const int *)t[3].p, (const int64_t *)t[5].p, n_slots, ck, cl);
    HIPCHK(hipGetLastError());
    size_t tmp_bytes = 0;
    if (n_kmers > 0x7ffffff0) { set_err("too many k-mers for one sort%s", ""); return GNX_EINVAL; }
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ck, sk, cl, sl, (size_t)n_kmers, 0u, (unsigned)(2 * seed_len), st));
    if ((rc = t[1].ensure(tmp_bytes + 16))) return rc; // (the per-slot keys are compacted by now)
    HIPCHK(rocprim::radix_sort_pairs(t[1].p, tmp_bytes, ck, sk, cl, sl, (size_t)n_kmers, 0u, (unsigned)(2 * seed_len), st)); // stable LSD radix sort: insertion order within a key
    uint64_t *hk = (uint64_t *)malloc((size_t)n_kmers * 8), *hl = (uint64_t *)malloc((size_t)n_kmers * 8);
    if (!hk || !hl) { free(hk); free(hl); set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
    if (hipMemcpyAsync(hk, sk, (size_t)n_kmers * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(hl, sl, (size_t)n_kmers * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { free(hk); free(hl); set_err("D2H of the index failed%s", ""); return GNX_EDEVICE; }
    *out_keys = hk; *out_locs = hl; *out_n = n_kmers;
    return GNX_OK;
}

// The device holds ONE resident seed index.  Every gnx_seed_index_set gives it a new GENERATION (under the API lock); a caller that must
// be sure it searches ITS index -- the read path of a graph handle, while other threads use other handles or the raw entry points --
// sets with gnx_seed_index_set_gen and searches with gnx_seed_find_batch_gen(generation): a search against a replaced index is refused
// with GNX_ESTALE, inside the same lock that runs the search (the check used to sit outside it), and the caller uploads again.
static uint64_t g_seed_gen = 0;      // generation of the resident index (0: none); guarded by g_api_mu
static int seed_index_set_locked(const uint64_t *keys, const uint64_t *locs, int64_t n_index, const uint8_t *node_cat, const int64_t *node_off, int64_t n_nodes, int seed_len);
int gnx_seed_index_set(const uint64_t *keys, const uint64_t *locs, int64_t n_index, const uint8_t *node_cat, const int64_t *node_off, int64_t n_nodes, int seed_len) {
    std::lock_guard<std::mutex> api(g_api_mu);
    return seed_index_set_locked(keys, locs, n_index, node_cat, node_off, n_nodes, seed_len);
}
int gnx_seed_index_set_gen(const uint64_t *keys, const uint64_t *locs, int64_t n_index, const uint8_t *node_cat, const int64_t *node_off, int64_t n_nodes, int seed_len, uint64_t *out_generation) {
    std::lock_guard<std::mutex> api(g_api_mu);
    if (!out_generation) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    const int rc = seed_index_set_locked(keys, locs, n_index, node_cat, node_off, n_nodes, seed_len);
    *out_generation = rc ? 0 : g_seed_gen;
    return rc;
}
static int seed_index_set_locked(const uint64_t *keys, const uint64_t *locs, int64_t n_index, const uint8_t *node_cat, const int64_t *node_off, int64_t n_nodes, int seed_len) {
    g_seed_gen++; // (also when the call fails half way: whatever was resident is not any more)
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    if (n_index < 0 || (n_index > 0 && (!keys || !locs)) || !node_off || n_nodes < 0 || n_nodes > 0x7ffffff0 || seed_len < 2 || seed_len > 32) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    int rc = ensure_init();
    if (rc) return rc;
    Ctx &c = g_ctx;
    hipStream_t st = c.own_stream;
    std::vector<int64_t> word_off((size_t)n_nodes + 1, 0);
    for (int64_t k = 0; k < n_nodes; k++) word_off[(size_t)k + 1] = word_off[(size_t)k] + (node_off[k + 1] - node_off[k] + 31) / 32;
    const int64_t n_words = word_off[(size_t)n_nodes], total = n_nodes ? node_off[n_nodes] : 0;
    if ((rc = c.sd_keys.ensure((size_t)std::max<int64_t>(n_index, 1) * 8))) return rc;
    if ((rc = c.sd_locs.ensure((size_t)std::max<int64_t>(n_index, 1) * 8))) return rc;
    if ((rc = c.sd_nodes.ensure((size_t)total + 64))) return rc;
    if ((rc = c.sd_node_off.ensure((size_t)(n_nodes + 1) * 8))) return rc;
    if ((rc = c.sd_word_off.ensure((size_t)(n_nodes + 1) * 8))) return rc;
    if ((rc = c.sd_words.ensure((size_t)std::max<int64_t>(n_words, 1) * 8))) return rc;
    if (n_index) { HIPCHK(hipMemcpyAsync(c.sd_keys.p, keys, (size_t)n_index * 8, hipMemcpyHostToDevice, st)); HIPCHK(hipMemcpyAsync(c.sd_locs.p, locs, (size_t)n_index * 8, hipMemcpyHostToDevice, st)); }
    if (total) HIPCHK(hipMemcpyAsync(c.sd_nodes.p, node_cat, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c.sd_node_off.p, node_off, (size_t)(n_nodes + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c.sd_word_off.p, word_off.data(), (size_t)(n_nodes + 1) * 8, hipMemcpyHostToDevice, st));
    if (n_words) hipLaunchKernelGGL(pack_nodes_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, (const uint8_t *)c.sd_nodes.p, (const int64_t *)c.sd_node_off.p,
                                    (const int64_t *)c.sd_word_off.p, (int)n_nodes, n_words, (uint64_t *)c.sd_words.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st)); // word_off is a local
    c.sd_n = n_index; c.sd_nodes_n = n_nodes; c.sd_seed_len = seed_len;
    return GNX_OK;
}

static int seed_find_locked(const uint8_t *read_cat, const int64_t *read_off, int64_t n_reads, gnx_seed_hit **out_hits, int64_t **out_hit_off);
int gnx_seed_find_batch(const uint8_t *read_cat, const int64_t *read_off, int64_t n_reads, gnx_seed_hit **out_hits, int64_t **out_hit_off) {
    std::lock_guard<std::mutex> api(g_api_mu);
    return seed_find_locked(read_cat, read_off, n_reads, out_hits, out_hit_off);
}
int gnx_seed_find_batch_gen(uint64_t generation, const uint8_t *read_cat, const int64_t *read_off, int64_t n_reads, gnx_seed_hit **out_hits, int64_t **out_hit_off) {
    std::lock_guard<std::mutex> api(g_api_mu);
    if (generation == 0 || generation != g_seed_gen) { set_err("the resident seed index has been replaced since generation %s%lld was set", "", (long long)generation); return GNX_ESTALE; }
    return seed_find_locked(read_cat, read_off, n_reads, out_hits, out_hit_off);
}
static int seed_find_locked(const uint8_t *read_cat, const int64_t *read_off, int64_t n_reads, gnx_seed_hit **out_hits, int64_t **out_hit_off) {
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    if (!read_off || n_reads < 0 || n_reads > 0x3ffffff0 || !out_hits || !out_hit_off) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    int rc = ensure_init();
    if (rc) return rc;
    Ctx &c = g_ctx;
    if (c.sd_n < 0) { set_err("no resident seed index: call gnx_seed_index_set first%s", ""); return GNX_EINVAL; }
    hipStream_t st = c.own_stream;
    const int seed_len = c.sd_seed_len;
    std::vector<int64_t> slot_off((size_t)n_reads + 1, 0);
    int64_t maxlen = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        const int64_t L = read_off[r + 1] - read_off[r];
        if (L < 0 || L > 100000) { set_err("bad read length at read %s%lld", "", (long long)r); return GNX_EINVAL; }
        slot_off[(size_t)r + 1] = slot_off[(size_t)r] + 2 * std::max<int64_t>(L - seed_len + 1, 0);
        maxlen = std::max(maxlen, L);
    }
    const int64_t n_slots = slot_off[(size_t)n_reads], total = n_reads ? read_off[n_reads] : 0;
    int64_t *hoff = (int64_t *)malloc((size_t)(n_reads + 1) * 8);
    if (!hoff) { set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
    struct Guard { void *a, *b; ~Guard() { free(a); free(b); } } guard{hoff, nullptr};
    *out_hits = nullptr; *out_hit_off = nullptr;
    if (n_slots == 0) { for (int64_t r = 0; r <= n_reads; r++) hoff[r] = 0; *out_hit_off = hoff; guard.a = nullptr; return GNX_OK; }
    if (n_slots > 0x7ffffff0) { set_err("too many read positions in one batch%s", ""); return GNX_EINVAL; }
    const int RW = (int)((maxlen + 31 + 31) / 32);
    DevBuf *t = c.sd_tmp;
    if ((rc = t[0].ensure((size_t)total + 64))) return rc;                       // reads
    if ((rc = t[1].ensure((size_t)total + 64))) return rc;                       // reverse complements
    if ((rc = t[2].ensure((size_t)(n_reads + 1) * 8))) return rc;                // read_off
    if ((rc = t[3].ensure((size_t)(n_reads + 1) * 8))) return rc;                // slot_off
    if ((rc = t[4].ensure((size_t)n_reads * 64 * RW * 8))) return rc;            // rainbows
    if ((rc = t[5].ensure((size_t)(std::max(n_slots, n_reads) + 1) * 8))) return rc; // counts, then the hit offsets per read
    if ((rc = t[6].ensure((size_t)(n_slots + 1) * 8))) return rc;                // offsets
    if ((rc = c.misc.ensure(64))) return rc;
    HIPCHK(hipMemcpyAsync(t[0].p, read_cat, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(t[2].p, read_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(t[3].p, slot_off.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(c.misc.p, 0, 64, st));
    if (total) hipLaunchKernelGGL(revcomp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint8_t *)t[0].p, (const int64_t *)t[2].p, (int)n_reads, total, (uint8_t *)t[1].p);
    const int64_t n_rw = n_reads * 64 * RW;
    hipLaunchKernelGGL(pack_reads_kernel, dim3((unsigned)((n_rw + 255) / 256)), dim3(256), 0, st, (const uint8_t *)t[0].p, (const int64_t *)t[2].p, (int)n_reads, RW, (uint8_t *)t[1].p, (uint64_t *)t[4].p);
    SeedCtx sx;
    sx.keys = (const uint64_t *)c.sd_keys.p; sx.locs = (const uint64_t *)c.sd_locs.p; sx.n_index = c.sd_n;
    sx.node_words = (const uint64_t *)c.sd_words.p; sx.node_off = (const int64_t *)c.sd_node_off.p; sx.word_off = (const int64_t *)c.sd_word_off.p;
    sx.read_words = (const uint64_t *)t[4].p; sx.read_off = (const int64_t *)t[2].p; sx.RW = RW; sx.seed_len = seed_len;
    const unsigned gb = (unsigned)((n_slots + 127) / 128);
    hipLaunchKernelGGL(seed_find_kernel<false>, dim3(gb), dim3(128), 0, st, sx, (const int64_t *)t[3].p, (int)n_reads, n_slots, (int64_t *)t[5].p, (const int64_t *)nullptr, (SeedHit *)nullptr);
    HIPCHK(hipGetLastError());
    int64_t *d_carry = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(c.misc.p) + 16);
    if ((rc = launch_scan((const int64_t *)t[5].p, (int)n_slots, (int64_t *)t[6].p, d_carry, st))) return rc;
    int64_t n_hits = 0;
    HIPCHK(hipMemcpyAsync(&n_hits, d_carry, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st)); // slot_off is a local, too
    if ((rc = t[7].ensure((size_t)std::max<int64_t>(n_hits, 1) * sizeof(SeedHit)))) return rc;
    if (n_hits) hipLaunchKernelGGL(seed_find_kernel<true>, dim3(gb), dim3(128), 0, st, sx, (const int64_t *)t[3].p, (int)n_reads, n_slots, (int64_t *)nullptr, (const int64_t *)t[6].p, (SeedHit *)t[7].p);
    HIPCHK(hipGetLastError());
    gnx_seed_hit *hh = (gnx_seed_hit *)malloc((size_t)std::max<int64_t>(n_hits, 1) * sizeof(gnx_seed_hit));
    guard.b = hh;
    if (!hh) { set_err("host allocation failed%s", ""); return GNX_ENOMEM; }
    static_assert(sizeof(gnx_seed_hit) == sizeof(SeedHit), "hit layout");
    if (n_hits) HIPCHK(hipMemcpyAsync(hh, t[7].p, (size_t)n_hits * sizeof(SeedHit), hipMemcpyDeviceToHost, st));
    // the offsets of the reads' first slots are all the caller wants of the slot offsets (8 bytes per read position would be 38 MB for 20 000 reads)
    hipLaunchKernelGGL(seed_read_off_kernel, dim3((unsigned)((n_reads + 1 + 255) / 256)), dim3(256), 0, st, (const int64_t *)t[6].p, (const int64_t *)t[3].p, n_reads + 1, (int64_t *)t[5].p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hoff, t[5].p, (size_t)(n_reads + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *out_hits = hh; *out_hit_off = hoff;
    guard.a = nullptr; guard.b = nullptr;
    return GNX_OK;
}

/* diagnostics (refused unless the process runs with GNX_DEBUG_ENTRY=1: a production caller cannot park the GPU by accident):
 * n_workgroups workgroups that each hold a whole CU's LDS and spin for `milliseconds`, on a stream of their own; returns
 * at once.  The stress leg of the claim protocol (tests/test_ticket.py): piped launches must finish with half the CUs taken away. */
int gnx_debug_occupy(int n_workgroups, int milliseconds) {
    std::lock_guard<std::mutex> api(g_api_mu);
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    if (!getenv("GNX_DEBUG_ENTRY")) { set_err("gnx_debug_occupy is a test hook: set GNX_DEBUG_ENTRY=1%s", ""); return GNX_EINVAL; }
    int rc = ensure_init();
    if (rc) return rc;
    if (n_workgroups <= 0 || milliseconds < 0 || milliseconds > 10000) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    // (function attributes are per device: set on every call -- a process may have gone through gnx_shutdown and another device)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)n_workgroups), dim3(64), 160 * 1024, g_ctx.s_in, (long long)milliseconds * 100000LL);
    HIPCHK(hipGetLastError());
    return GNX_OK;
}

/* diagnostics: counters kept on context 0's device.  which = 0: items of piped launches that were run by a workgroup other than
 * their own since the last reset (claim_items: the abnormal path the tests force); reset != 0 zeroes it after reading. */
int gnx_debug_counter(int which, int reset, int64_t *out) {
    std::lock_guard<std::mutex> api(g_api_mu);
    CtxScope sc(ctx_at(0));
    g_err[0] = 0;
    if (which < 0 || which > 6 || !out) { set_err("bad argument%s", ""); return GNX_EINVAL; }
    if (which == 1 || which == 2) { // combined gnx_align_pair batches run / pairs served by them
        *out = which == 1 ? g_pq_batches.load() : g_pq_pairs.load();
        if (reset) { if (which == 1) g_pq_batches = 0; else g_pq_pairs = 0; }
        return GNX_OK;
    }
    if (which >= 5) { *out = which == 5 ? g_last_w64_r.load() : g_last_w64_ck.load(); return GNX_OK; } // geometry of the last 64-lane affine sweep
    int rc = ensure_init();
    if (rc) return rc;
    HIPCHK(hipSetDevice(g_ctx.device));
    HIPCHK(hipDeviceSynchronize());
    unsigned long long v = 0;
    if (which >= 3) { // quirk-Q1 restarts of the snapshot path's affine walks: 3 = those that changed the state, 4 = all
        unsigned long long q[2] = {0, 0};
        HIPCHK(hipMemcpyFromSymbol(q, HIP_SYMBOL(g_dev_q1), sizeof(q)));
        if (reset) { const unsigned long long z[2] = {0, 0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dev_q1), z, sizeof(z))); }
        *out = (int64_t)(which == 3 ? q[1] : q[0]);
        return GNX_OK;
    }
    HIPCHK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_dev_claims_stolen), sizeof(v)));
    if (reset) { const unsigned long long z = 0; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dev_claims_stolen), &z, sizeof(z))); }
    *out = (int64_t)v;
    return GNX_OK;
}

int gnx_get_timing(gnx_timing *out) {
    if (!out) return GNX_EINVAL;
    std::lock_guard<std::mutex> api(g_api_mu);
    *out = ctx_at(0).timing;
    return GNX_OK;
}

} // extern "C"

#include "gsw_reads.hip.h"
