// The read path's worker pool (include/gonomics_genomegraph.hpp: GswPool, parallelFor) on the CPU: every index exactly once in both
// partitions, more threads than items, regions of changing width one after the other, the exception of the LOWEST index rethrown after
// all other items ran, concurrent callers taking turns.  Host code only -- nothing of the library is called.
#include <atomic>
#include <cstdio>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>

#include "gonomics_genomegraph.hpp"

using namespace gonomics::genomeGraph;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); fails++; } } while (0)

int main() {
    for (int threads : {1, 2, 3, 16, 64}) {
        for (size_t n : {(size_t)0, (size_t)1, (size_t)31, (size_t)32, (size_t)33, (size_t)1000, (size_t)20001}) {
            for (bool blocks : {false, true}) {
                std::vector<std::atomic<int>> hit(n);
                for (auto &h : hit) h = 0;
                parallelFor(n, threads, [&](size_t k) { hit[k]++; }, blocks);
                bool once = true;
                for (auto &h : hit) once = once && h.load() == 1;
                CHECK(once);
            }
        }
    }
    // the lowest failing index wins, everything else still runs
    for (bool blocks : {false, true}) {
        std::vector<std::atomic<int>> hit(5000);
        for (auto &h : hit) h = 0;
        size_t got = 0;
        try {
            parallelFor(hit.size(), 8, [&](size_t k) { hit[k]++; if (k == 4321 || k == 77 || k == 2500) throw GoPanic(std::to_string(k)); }, blocks);
        } catch (const GoPanic &e) { got = (size_t)std::stoul(e.what()); }
        CHECK(got == 77);
        int total = 0;
        for (auto &h : hit) total += h.load();
        CHECK(total == 5000);
    }
    // two callers at once: regions take turns, nothing is lost
    {
        std::atomic<long> sum{0};
        auto caller = [&](int threads) { for (int rep = 0; rep < 50; rep++) parallelFor(2000, threads, [&](size_t k) { sum += (long)k; }); };
        std::thread a(caller, 4), b(caller, 7);
        a.join(); b.join();
        CHECK(sum.load() == 2L * 50 * (1999L * 2000 / 2));
    }
    CHECK(gswThreads(5) == 5 && gswThreads(100000) == 256 && gswThreads(0) >= 1);
    // Go's append capacities (16-byte elements), as pinned in tests/test_n2_gsw.py::test_go_slice_model
    CHECK(goNextCap(1, 0) == 1 && goNextCap(2, 1) == 2 && goNextCap(3, 2) == 4 && goNextCap(5, 4) == 8 && goNextCap(513, 512) == 848);
    if (fails) { fprintf(stderr, "%d checks failed\n", fails); return 1; }
    printf("pool ok\n");
    return 0;
}
