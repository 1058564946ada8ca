// TEST INFRASTRUCTURE (CPU baseline of the graph aligner's read path): gnx_gsw_extend_batch's signature served by
// the CPU oracle (oracle/gnx_oracle.c: or_gsw_extend, the literal restatement of genomeGraph/search.go:234-321) on all host threads --
// the reference's own way of going parallel (-t worker goroutines, genomeGraph/routines.go:12-65).  Linked only into the benchmark
// binary tests/cpp/gsw_mirror_test.bin and selected there by the "cpu" argument; the product never sees it.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "gnx_align.h"

extern "C" {
typedef struct { int64_t run; uint8_t op; uint8_t pad[7]; } or_cigar;
int or_gsw_extend(int side, const uint8_t *alpha, int64_t n, const uint8_t *beta, int64_t m, const int64_t *sc, int64_t gapPen,
                  const or_cigar *route_in, int64_t n_in, int64_t curr_max_in,
                  int64_t *out_score, int64_t *out_i, int64_t *out_j, or_cigar **out_route, int64_t *out_len);
void or_free(void *p);
}

int cpu_gsw_threads = 0; // 0 = every hardware thread

extern "C" int cpu_gsw_extend_batch(int side, const int64_t *scores, int64_t gap_pen, int64_t n_pairs,
                                    const uint8_t *alpha_cat, const int64_t *alpha_off, const uint8_t *beta_cat, const int64_t *beta_off,
                                    int64_t *out_score, int64_t *out_end_i, int64_t *out_end_j, gnx_cigar **out_ops, int64_t **out_ops_off) {
    std::vector<or_cigar *> routes((size_t)n_pairs, nullptr);
    std::vector<int64_t> lens((size_t)n_pairs, 0);
    std::atomic<int64_t> next{0};
    std::atomic<int> bad{0};
    int nt = cpu_gsw_threads > 0 ? cpu_gsw_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    auto work = [&]() {
        for (;;) {
            const int64_t p = next.fetch_add(1);
            if (p >= n_pairs) break;
            const int rc = or_gsw_extend(side, alpha_cat + alpha_off[p], alpha_off[p + 1] - alpha_off[p], beta_cat + beta_off[p], beta_off[p + 1] - beta_off[p],
                                         scores, gap_pen, nullptr, 0, 0, &out_score[p], &out_end_i[p], &out_end_j[p], &routes[(size_t)p], &lens[(size_t)p]);
            if (rc) bad = rc;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (bad) return GNX_EINVAL;
    int64_t *off = (int64_t *)malloc((size_t)(n_pairs + 1) * 8);
    off[0] = 0;
    for (int64_t p = 0; p < n_pairs; p++) off[p + 1] = off[p] + lens[(size_t)p];
    gnx_cigar *ops = (gnx_cigar *)calloc((size_t)(off[n_pairs] + 1), sizeof(gnx_cigar));
    for (int64_t p = 0; p < n_pairs; p++) {
        for (int64_t k = 0; k < lens[(size_t)p]; k++) { ops[off[p] + k].run_length = routes[(size_t)p][k].run; ops[off[p] + k].op = routes[(size_t)p][k].op; }
        or_free(routes[(size_t)p]);
    }
    *out_ops = ops; *out_ops_off = off; // (gnx_free releases what is not from the library's pinned pool with free())
    return GNX_OK;
}
