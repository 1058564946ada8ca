// Many host threads calling the single-pair entry point at once -- the reference's way of going parallel is a pool of goroutines that
// each call align.* in a loop (genomeGraph/routines.go:12-65).  T threads x N calls of gnx_align_pair must give exactly the results
// of the same calls made one after the other, and at >= 8 x the serial rate with 16 threads: the library combines
// concurrent calls into device batches.  Output: one JSON line; exit 0 ok, 1 mismatch / too slow, 2 no device.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "gnx_align.h"

static uint64_t sm64(uint64_t &x) { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

struct Res { int64_t score; std::vector<gnx_cigar> ops; int rc; };
static gnx_params prm_a;

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 1000;
    const double want = argc > 3 ? atof(argv[3]) : 8.0;
    const bool mixed = argc > 4; // a fourth argument: every fourth thread calls with other parameters (ConstGap)
    if (gnx_init(0, 0) != GNX_OK) { fprintf(stderr, "%s\n", gnx_last_error()); return 2; }
    gnx_params &prm = prm_a;
    memset(&prm, 0, sizeof(prm));
    prm.mode = GNX_AFFINE_GAP;
    const int64_t hc2[25] = {90, -330, -236, -356, -208, -330, 100, -318, -236, -196, -236, -318, 100, -330, -196, -356, -236, -330, 90, -208, -208, -196, -196, -208, -202};
    for (int k = 0; k < 25; k++) prm.scores[k] = hc2[k];
    prm.gap_open = -600; prm.gap_extend = -150; prm.checkersize_i = 10000; prm.checkersize_j = 10000;
    // T * N pairs: reads of 100 .. 150 bases against windows of 1000 .. 1500 bases (what cmd/globalAlignmentAnchor's loop sees)
    const int P = T * N;
    std::vector<std::vector<uint8_t>> A((size_t)P), B((size_t)P);
    uint64_t seed = 12345;
    for (int q = 0; q < P; q++) {
        const int m = 1000 + (int)(sm64(seed) % 501), n = 100 + (int)(sm64(seed) % 51);
        B[(size_t)q].resize((size_t)m);
        for (auto &x : B[(size_t)q]) x = (uint8_t)(sm64(seed) & 3);
        const int o = (int)(sm64(seed) % (uint64_t)(m - n));
        A[(size_t)q].assign(B[(size_t)q].begin() + o, B[(size_t)q].begin() + o + n);
        for (auto &x : A[(size_t)q]) if (sm64(seed) % 50 == 0) x = (uint8_t)(sm64(seed) & 3);
        if (q % 997 == 5) A[(size_t)q][3] = 9; // a base the Go code would panic on: GNX_EBASE for THIS pair only
    }
    // every fourth thread's pairs go through align.ConstGap instead: requests with different parameters wait in the same queue and
    // must never be combined into one batch
    gnx_params prm_c = prm;
    prm_c.mode = GNX_CONST_GAP; prm_c.gap_open = -430; prm_c.gap_extend = 0;
    auto call = [&](int q, Res &r) {
        int64_t sc = 0, nops = 0;
        gnx_cigar *ops = nullptr;
        const gnx_params &prm = (mixed && (q / N) % 4 == 3) ? prm_c : ::prm_a;
        r.rc = gnx_align_pair(&prm, A[(size_t)q].data(), (int64_t)A[(size_t)q].size(), B[(size_t)q].data(), (int64_t)B[(size_t)q].size(), &sc, &ops, &nops);
        if (r.rc == GNX_OK) { r.score = sc; r.ops.assign(ops, ops + nops); gnx_free(ops); }
    };
    std::vector<Res> serial((size_t)P), par((size_t)P);
    { Res w; call(0, w); } // warm-up: workspace allocation
    const int NS = P < 2000 ? P : 2000; // the serial rate from a sample (a serial call is ~0.2 ms)
    auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < NS; q++) call(q, serial[(size_t)q]);
    const double serial_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int q = NS; q < P; q++) call(q, serial[(size_t)q]);
    int64_t b0 = 0, p0 = 0;
    gnx_debug_counter(1, 1, &b0); gnx_debug_counter(2, 1, &p0);
    t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() { for (int k = 0; k < N; k++) call(t * N + k, par[(size_t)(t * N + k)]); });
    for (auto &x : th) x.join();
    const double par_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int64_t batches = 0, served = 0;
    gnx_debug_counter(1, 0, &batches); gnx_debug_counter(2, 0, &served);
    int bad = 0, errs = 0;
    for (int q = 0; q < P; q++) {
        const Res &a = serial[(size_t)q], &b = par[(size_t)q];
        if (a.rc != b.rc) { bad++; continue; }
        if (a.rc != GNX_OK) { errs++; continue; }
        if (a.score != b.score || a.ops.size() != b.ops.size()) { bad++; continue; }
        for (size_t k = 0; k < a.ops.size(); k++) if (a.ops[k].run_length != b.ops[k].run_length || a.ops[k].op != b.ops[k].op) { bad++; break; }
    }
    const double serial_rate = NS / serial_s, par_rate = P / par_s;
    printf("{\"threads\": %d, \"calls_per_thread\": %d, \"serial_calls_per_s\": %.0f, \"concurrent_calls_per_s\": %.0f, \"speedup\": %.2f, \"combined_batches\": %lld, \"pairs_in_combined_batches\": %lld, "
           "\"mismatches\": %d, \"calls_with_GNX_EBASE\": %d}\n", T, N, serial_rate, par_rate, par_rate / serial_rate, (long long)batches, (long long)served, bad, errs);
    gnx_shutdown();
    return (bad == 0 && errs > 0 && par_rate >= want * serial_rate) ? 0 : 1;
}
