// Wall clock of the N1 / N2 batch entry points through the raw C ABI from a compiled host (what a cgo shim pays), next to the kernels'
// own time: the Python tools (bench_n1.py / bench_n2.py) add the binding's list handling on top.
// Build + run on the GPU box:  g++ -std=c++17 -O2 -Iinclude -o tools/bench_cabi.bin tools/bench_cabi.cpp gonomics_amd/libgonomics_align_hip.so \
//                                   -Wl,-rpath,$PWD/gonomics_amd -L/opt/rocm/lib -lamdhip64 && tools/bench_cabi.bin
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "gnx_align.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    if (gnx_device_count() <= 0) { std::printf("{\"error\": \"no HIP device\"}\n"); return 2; }
    gnx_init(0, (int64_t)32 << 30);
    std::mt19937_64 rng(7);
    const int64_t sc[25] = {90, -330, -236, -356, -208, -330, 100, -318, -236, -196, -236, -318, 100, -330, -196, -356, -236, -330, 90, -208, -208, -196, -196, -208, -202};
    {   // N2: 200 000 extensions of a 75-base read part against 87 target bases
        const int64_t n = 200000, la = 87, lb = 75;
        std::vector<uint8_t> a((size_t)(n * la)), b((size_t)(n * lb));
        for (auto &x : a) x = (uint8_t)(rng() & 3);
        for (int64_t p = 0; p < n; p++) for (int64_t k = 0; k < lb; k++) b[(size_t)(p * lb + k)] = (rng() % 50 == 0) ? (uint8_t)(rng() & 3) : a[(size_t)(p * la + k + 6)];
        std::vector<int64_t> ao((size_t)n + 1), bo((size_t)n + 1), score((size_t)n), ei((size_t)n), ej((size_t)n);
        for (int64_t p = 0; p <= n; p++) { ao[(size_t)p] = p * la; bo[(size_t)p] = p * lb; }
        for (int side = 0; side < 2; side++) {
            double best = 1e30; gnx_timing tm = {};
            for (int it = 0; it < 4; it++) {
                gnx_cigar *ops = nullptr; int64_t *off = nullptr;
                const double t0 = now_ms();
                const int rc = gnx_gsw_extend_batch(side, sc, -600, n, a.data(), ao.data(), b.data(), bo.data(), score.data(), ei.data(), ej.data(), &ops, &off);
                const double dt = now_ms() - t0;
                if (rc) { std::printf("{\"error\": \"%s\"}\n", gnx_last_error()); return 1; }
                gnx_free(ops); gnx_free(off);
                if (it && dt < best) { best = dt; gnx_get_timing(&tm); }
            }
            std::printf("{\"series\": \"%s through the C ABI (compiled host)\", \"pairs\": %lld, \"host_call_ms\": %.3f, \"kernels_ms\": %.3f, \"pairs_per_s\": %.0f}\n",
                        side ? "RightDynamicAln" : "LeftDynamicAln", (long long)n, best, tm.total_ms, n / (best * 1e-3));
        }
    }
    for (const int64_t n : {(int64_t)4096, (int64_t)32768}) {   // N1: pairs of 160 x 3000 chunks of 3 (32 768 pairs: the bases cross PCIe in sub-batches under the DP)
        const int64_t chunk = 3, na = 160 * chunk, nb = 3000 * chunk;
        std::vector<uint8_t> a((size_t)(n * na)), b((size_t)(n * nb));
        for (auto &x : b) x = (uint8_t)(rng() & 3);
        for (int64_t p = 0; p < n; p++) for (int64_t k = 0; k < na; k++) a[(size_t)(p * na + k)] = b[(size_t)(p * nb + 300 * chunk + k)];
        std::vector<int64_t> ao((size_t)n + 1), bo((size_t)n + 1), score((size_t)n);
        for (int64_t p = 0; p <= n; p++) { ao[(size_t)p] = p * na; bo[(size_t)p] = p * nb; }
        gnx_params prm = {};
        prm.mode = GNX_AFFINE_GAP_HIGHMEM; for (int x = 0; x < 25; x++) prm.scores[x] = sc[x];
        prm.gap_open = -400; prm.gap_extend = -30; prm.checkersize_i = prm.checkersize_j = 10000;
        double best = 1e30; gnx_timing tm = {};
        for (int it = 0; it < 4; it++) {
            gnx_cigar *ops = nullptr; int64_t *off = nullptr;
            const double t0 = now_ms();
            const int rc = gnx_affine_gap_chunk_batch(&prm, chunk, n, a.data(), ao.data(), b.data(), bo.data(), score.data(), &ops, &off);
            const double dt = now_ms() - t0;
            if (rc) { std::printf("{\"error\": \"%s\"}\n", gnx_last_error()); return 1; }
            gnx_free(ops); gnx_free(off);
            if (it && dt < best) { best = dt; gnx_get_timing(&tm); }
        }
        std::printf("{\"series\": \"AffineGapChunk through the C ABI (compiled host)\", \"pairs\": %lld, \"host_call_ms\": %.3f, \"fill_traceback_ms\": %.3f, \"chunk_cells_per_s_call\": %.3e}\n",
                    (long long)n, best, tm.total_ms, (double)n * 160 * 3000 / (best * 1e-3));
    }
    return 0;
}
