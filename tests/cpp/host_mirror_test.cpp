// Compiles against include/gonomics_align.hpp + the C ABI and re-states three of the reference's table tests
// (align/affineGap_test.go:45-55,120-155; align/view_test.go:28-38) through the C++ host mirror.
// Exit code 0 = all good, 2 = no GPU (the library has no CPU fallback), 1 = mismatch.
#include <cstdio>
#include "gonomics_align.hpp"

int main() {
    if (gnx_device_count() <= 0) { std::printf("no HIP device: %s\n", "skipping compute"); return 2; }
    struct T { const char *a, *b, *aln; };
    const T aff[] = {{"ACGT", "ACGT", "ACGT\nACGT\n"}, {"ACGT", "CGT", "ACGT\n-CGT\n"}, {"CGCGCGCGCG", "CGAAAACGCGTTTTCGCG", "CG----CGCG----CGCG\nCGAAAACGCGTTTTCGCG\n"}};
    int bad = 0;
    for (const T &t : aff) {
        auto a = dna::StringToBases(t.a), b = dna::StringToBases(t.b);
        auto r = align::AffineGap_highMem(a, b, align::DefaultScoreMatrix(), -400, -30);
        if (align::View(a, b, r.second) != t.aln) { std::printf("affine mismatch %s %s\n", t.a, t.b); bad++; }
        auto c = align::ConstGap(a, b, align::DefaultScoreMatrix(), -430);
        if (align::View(a, b, c.second) != t.aln) { std::printf("const mismatch %s %s\n", t.a, t.b); bad++; }
    }
    auto l = align::AffineGapLocal(dna::StringToBases("TCACTTTCGCACGTT"), dna::StringToBases("CACACG"), align::DefaultScoreMatrix(), -600, -150);
    if (l.first != 460 || align::PrintCigar(l.second) != "7D6M2D") { std::printf("local mismatch\n"); bad++; }
    try { align::ConstGap(dna::StringToBases("ACgT"), dna::StringToBases("ACGT"), align::DefaultScoreMatrix(), -430); bad++; }
    catch (const std::out_of_range &) {}
    // N1: affineAlignChunkTests row 1 (align/affineGap_test.go:27-36, chunk size 3)
    {
        auto a = dna::StringToBases("ACG"), b = dna::StringToBases("ACGACG");
        auto r = align::AffineGapChunk(a, b, align::DefaultScoreMatrix(), -400, -30, 3);
        int64_t cols = 0;
        for (const auto &c : r.second) cols += c.RunLength;
        if (cols != 6) { std::printf("chunk mismatch\n"); bad++; }
    }
    // N2: a perfect match extends to the origin (left) / ends at (n, m) (right)
    {
        auto a = dna::StringToBases("ACGTACGTGG");
        int64_t perfect = 0;
        for (auto x : a) perfect += align::HumanChimpTwoScoreMatrix()[x][x];
        auto l2 = genomeGraph::LeftDynamicAln(a, a, align::HumanChimpTwoScoreMatrix(), -600);
        auto r2 = genomeGraph::RightDynamicAln(a, a, align::HumanChimpTwoScoreMatrix(), -600);
        const std::vector<cigar::Cigar> want = {cigar::Cigar{10, 'M'}};
        if (l2.score != perfect || !(l2.route == want) || l2.i != 0 || l2.j != 0) { std::printf("left extension mismatch\n"); bad++; }
        if (r2.score != perfect || !(r2.route == want) || r2.i != 10 || r2.j != 10) { std::printf("right extension mismatch\n"); bad++; }
    }
    std::printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
