// gonomics_align.hpp -- C++ host-side mirror of gonomics' `align` package API for the DP hot path, written above the
// C ABI (gnx_align.h).  The reference is Go (no Go toolchain in the build image), so the compiled-language host side
// is C++: same function names, argument order and meaning, and error behaviour as the Go functions:
//
//   align::AffineGap / AffineGap_customizeCheckersize      /root/reference/align/affineGap.go:59,73
//   align::ConstGap / ConstGap_customizeCheckersize        /root/reference/align/constGap.go:13,73
//   align::AffineGap_highMem / AffineGapLocal              /root/reference/align/affineGap_highMem.go:99,105
//   align::ConstGap_highMem                                /root/reference/align/constGap_highMem.go:11
//   align::AlignBatch (the batched loop of cmd/globalAlignmentAnchor.go:352-384 and the FIFO engine of
//                      affineGap_highMem.go:120-179: results in input order)
//   align::View / PrintCigar                               /root/reference/align/view.go:26-60
//   align::Cigar, ColM/ColI/ColD, score matrices           /root/reference/align/align.go:12-64
//
// Go panics become C++ exceptions: base >= 5 -> std::out_of_range ("index out of range"), empty input to a
// low-memory function (the Go code never terminates) -> std::invalid_argument, everything else std::runtime_error.
#ifndef GONOMICS_ALIGN_HPP
#define GONOMICS_ALIGN_HPP

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gnx_align.h"

namespace dna {
using Base = uint8_t; // /root/reference/dna/dna.go:5-21
enum : Base { A = 0, C = 1, G = 2, T = 3, N = 4, LowerA = 5, LowerC = 6, LowerG = 7, LowerT = 8, LowerN = 9, Gap = 10, Dot = 11, Nil = 12 };

inline std::vector<Base> StringToBases(const std::string &s) { // dna/convert.go:143-153
    std::vector<Base> out(s.size());
    for (size_t i = 0; i < s.size(); i++) {
        switch (s[i]) {
        case 'A': out[i] = A; break; case 'C': out[i] = C; break; case 'G': out[i] = G; break; case 'T': out[i] = T; break;
        case 'N': out[i] = N; break; case 'a': out[i] = LowerA; break; case 'c': out[i] = LowerC; break;
        case 'g': out[i] = LowerG; break; case 't': out[i] = LowerT; break; case 'n': out[i] = LowerN; break;
        case '-': out[i] = Gap; break; case '*': out[i] = Nil; break; case '.': out[i] = Dot; break;
        default: throw std::invalid_argument(std::string("Error: '") + s[i] + "' is an invalid base.");
        }
    }
    return out;
}
inline std::string BasesToString(const std::vector<Base> &b) { // dna/convert.go:178-187
    static const char tab[] = "ACGTNacgtn-.*";
    std::string s(b.size(), '?');
    for (size_t i = 0; i < b.size(); i++) { if (b[i] > Nil) throw std::out_of_range("index out of range"); s[i] = tab[b[i]]; }
    return s;
}
inline void AllToUpper(std::vector<Base> &b) { // dna/modify.go:60
    for (auto &x : b) if (x >= LowerA && x <= LowerN) x = (Base)(x - 5);
}
} // namespace dna

namespace align {

using ColType = uint8_t;
constexpr ColType ColM = 0, ColI = 1, ColD = 2;
struct Cigar { int64_t RunLength; ColType Op; };
using ScoreMatrix = std::array<std::array<int64_t, 5>, 5>;

inline const ScoreMatrix &DefaultScoreMatrix() { static const ScoreMatrix m = {{{91, -114, -31, -123, -44}, {-114, 100, -125, -31, -43}, {-31, -125, 100, -114, -43}, {-123, -31, -114, 91, -44}, {-44, -43, -43, -44, -43}}}; return m; }
inline const ScoreMatrix &HoxD55ScoreMatrix() { static const ScoreMatrix m = {{{91, -114, -31, -123, 0}, {-114, 100, -125, -31, 0}, {-31, -125, 100, -114, 0}, {-123, -31, -114, 91, 0}, {0, 0, 0, 0, 0}}}; return m; }
inline const ScoreMatrix &MouseRatScoreMatrix() { return HoxD55ScoreMatrix(); }
inline const ScoreMatrix &HumanChimpTwoScoreMatrix() { static const ScoreMatrix m = {{{90, -330, -236, -356, -208}, {-330, 100, -318, -236, -196}, {-236, -318, 100, -330, -196}, {-356, -236, -330, 90, -208}, {-208, -196, -196, -208, -202}}}; return m; }

namespace detail {
inline gnx_params params(int mode, const ScoreMatrix &s, int64_t gapOpen, int64_t gapExtend, int64_t ci, int64_t cj) {
    gnx_params p{};
    p.mode = mode;
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) p.scores[a * 5 + b] = s[a][b];
    p.gap_open = gapOpen; p.gap_extend = gapExtend; p.checkersize_i = ci; p.checkersize_j = cj;
    return p;
}
[[noreturn]] inline void raise(int rc) {
    const std::string msg = gnx_last_error();
    if (rc == GNX_EBASE) throw std::out_of_range("runtime error: index out of range: " + msg);
    if (rc == GNX_EEMPTY) throw std::invalid_argument(msg);
    throw std::runtime_error("gnx error " + std::to_string(rc) + ": " + msg);
}
inline std::pair<int64_t, std::vector<Cigar>> one(const gnx_params &p, const std::vector<dna::Base> &a, const std::vector<dna::Base> &b) {
    int64_t score = 0, n = 0;
    gnx_cigar *ops = nullptr;
    const int rc = gnx_align_pair(&p, a.data(), (int64_t)a.size(), b.data(), (int64_t)b.size(), &score, &ops, &n);
    if (rc) raise(rc);
    std::vector<Cigar> route((size_t)n);
    for (int64_t k = 0; k < n; k++) route[(size_t)k] = Cigar{ops[k].run_length, ops[k].op};
    gnx_free(ops);
    return {score, std::move(route)};
}
} // namespace detail

inline std::pair<int64_t, std::vector<Cigar>> AffineGap_customizeCheckersize(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const ScoreMatrix &scores, int64_t gapOpen, int64_t gapExtend, int checkersize_i, int checkersize_j) {
    return detail::one(detail::params(GNX_AFFINE_GAP, scores, gapOpen, gapExtend, checkersize_i, checkersize_j), alpha, beta);
}
inline std::pair<int64_t, std::vector<Cigar>> AffineGap(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const ScoreMatrix &scores, int64_t gapOpen, int64_t gapExtend) {
    return AffineGap_customizeCheckersize(alpha, beta, scores, gapOpen, gapExtend, 10000, 10000);
}
inline std::pair<int64_t, std::vector<Cigar>> ConstGap_customizeCheckersize(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const ScoreMatrix &scores, int64_t gapPen, int checkersize_i, int checkersize_j) {
    return detail::one(detail::params(GNX_CONST_GAP, scores, gapPen, 0, checkersize_i, checkersize_j), alpha, beta);
}
inline std::pair<int64_t, std::vector<Cigar>> ConstGap(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const ScoreMatrix &scores, int64_t gapPen) {
    return ConstGap_customizeCheckersize(alpha, beta, scores, gapPen, 10000, 10000);
}
inline std::pair<int64_t, std::vector<Cigar>> AffineGap_highMem(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const ScoreMatrix &scores, int64_t gapOpen, int64_t gapExtend) {
    return detail::one(detail::params(GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend, 10000, 10000), alpha, beta);
}
inline std::pair<int64_t, std::vector<Cigar>> AffineGapLocal(const std::vector<dna::Base> &target, const std::vector<dna::Base> &query, const ScoreMatrix &scores, int64_t gapOpen, int64_t gapExtend) {
    return detail::one(detail::params(GNX_AFFINE_GAP_LOCAL, scores, gapOpen, gapExtend, 10000, 10000), target, query);
}
inline std::pair<int64_t, std::vector<Cigar>> ConstGap_highMem(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const ScoreMatrix &scores, int64_t gapPen) {
    return detail::one(detail::params(GNX_CONST_GAP_HIGHMEM, scores, gapPen, 0, 10000, 10000), alpha, beta);
}

// TargetQueryPair + batched engine: GoAffineGapLocalEngine's channels become one call over a batch, FIFO order kept.
struct TargetQueryPair { std::vector<dna::Base> Target, Query; int64_t Score = 0; std::vector<Cigar> Cigar_; };

inline void AlignBatch(int mode, const ScoreMatrix &scores, int64_t gapOpen, int64_t gapExtend, int checkersize_i, int checkersize_j,
                       const std::vector<std::vector<dna::Base>> &alphas, const std::vector<std::vector<dna::Base>> &betas,
                       std::vector<int64_t> &out_scores, std::vector<std::vector<Cigar>> &out_routes) {
    const int64_t n = (int64_t)alphas.size();
    std::vector<int64_t> aoff((size_t)n + 1, 0), boff((size_t)n + 1, 0);
    for (int64_t k = 0; k < n; k++) { aoff[(size_t)k + 1] = aoff[(size_t)k] + (int64_t)alphas[(size_t)k].size(); boff[(size_t)k + 1] = boff[(size_t)k] + (int64_t)betas[(size_t)k].size(); }
    std::vector<dna::Base> acat((size_t)aoff[(size_t)n] + 1), bcat((size_t)boff[(size_t)n] + 1);
    for (int64_t k = 0; k < n; k++) {
        std::copy(alphas[(size_t)k].begin(), alphas[(size_t)k].end(), acat.begin() + aoff[(size_t)k]);
        std::copy(betas[(size_t)k].begin(), betas[(size_t)k].end(), bcat.begin() + boff[(size_t)k]);
    }
    const gnx_params p = detail::params(mode, scores, gapOpen, gapExtend, checkersize_i, checkersize_j);
    out_scores.assign((size_t)n, 0);
    gnx_cigar *ops = nullptr; int64_t *off = nullptr;
    const int rc = gnx_align_batch(&p, n, acat.data(), aoff.data(), bcat.data(), boff.data(), out_scores.data(), &ops, &off);
    if (rc) detail::raise(rc);
    out_routes.assign((size_t)n, {});
    for (int64_t k = 0; k < n; k++) for (int64_t x = off[k]; x < off[k + 1]; x++) out_routes[(size_t)k].push_back(Cigar{ops[x].run_length, ops[x].op});
    gnx_free(ops); gnx_free(off);
}

inline void AffineGapLocalEngine(const ScoreMatrix &scores, int64_t gapOpen, int64_t gapExtend, std::vector<TargetQueryPair> &pairs) {
    std::vector<std::vector<dna::Base>> t, q;
    for (auto &p : pairs) { t.push_back(p.Target); q.push_back(p.Query); }
    std::vector<int64_t> sc; std::vector<std::vector<Cigar>> rt;
    AlignBatch(GNX_AFFINE_GAP_LOCAL, scores, gapOpen, gapExtend, 10000, 10000, t, q, sc, rt);
    for (size_t k = 0; k < pairs.size(); k++) { pairs[k].Score = sc[k]; pairs[k].Cigar_ = std::move(rt[k]); }
}

// align.AffineGapChunk (align/affineGap_highMem.go:227-268), "next" row N1
inline std::pair<int64_t, std::vector<Cigar>> AffineGapChunk(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const ScoreMatrix &scores, int64_t gapOpen, int64_t gapExtend, int64_t chunkSize) {
    const gnx_params p = detail::params(GNX_AFFINE_GAP_HIGHMEM, scores, gapOpen, gapExtend, 10000, 10000);
    const int64_t aoff[2] = {0, (int64_t)alpha.size()}, boff[2] = {0, (int64_t)beta.size()};
    int64_t score = 0, *off = nullptr;
    gnx_cigar *ops = nullptr;
    const int rc = gnx_affine_gap_chunk_batch(&p, chunkSize, 1, alpha.data(), aoff, beta.data(), boff, &score, &ops, &off);
    if (rc) detail::raise(rc);
    std::vector<Cigar> route((size_t)off[1]);
    for (int64_t k = 0; k < off[1]; k++) route[(size_t)k] = Cigar{ops[k].run_length, ops[k].op};
    gnx_free(ops); gnx_free(off);
    return {score, std::move(route)};
}

inline std::string PrintCigar(const std::vector<Cigar> &ops) { // align/view.go:26-33
    std::string s;
    for (const auto &c : ops) { s += std::to_string(c.RunLength); s += (c.Op == ColM ? 'M' : c.Op == ColI ? 'I' : c.Op == ColD ? 'D' : '?'); }
    return s;
}
inline std::string View(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const std::vector<Cigar> &ops) { // align/view.go:37-60
    const std::string a = dna::BasesToString(alpha), b = dna::BasesToString(beta);
    std::string one, two;
    size_t i = 0, j = 0;
    for (const auto &c : ops) {
        const size_t n = (size_t)c.RunLength;
        if (c.Op == ColM) { one += a.substr(i, n); two += b.substr(j, n); i += n; j += n; }
        else if (c.Op == ColI) { one += std::string(n, '-'); two += b.substr(j, n); j += n; }
        else if (c.Op == ColD) { one += a.substr(i, n); two += std::string(n, '-'); i += n; }
    }
    return one + "\n" + two + "\n";
}
} // namespace align
#endif

// ---- "next" row N2: the seed-extension DPs of the graph aligner (genomeGraph/search.go:234-321) -------------------------
namespace cigar {
struct Cigar { int64_t RunLength; uint8_t Op; }; // cigar.Cigar{RunLength int; Op byte}: 'M', 'I', 'D'
inline bool operator==(const Cigar &a, const Cigar &b) { return a.RunLength == b.RunLength && a.Op == b.Op; }
} // namespace cigar

namespace genomeGraph {
struct DynamicAln { int64_t score; std::vector<cigar::Cigar> route; int i, j; };

namespace detail {
inline DynamicAln extend(int side, const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const align::ScoreMatrix &scores, int64_t gapPen) {
    int64_t flat[25];
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) flat[a * 5 + b] = scores[(size_t)a][(size_t)b];
    const int64_t aoff[2] = {0, (int64_t)alpha.size()}, boff[2] = {0, (int64_t)beta.size()};
    int64_t score = 0, ei = 0, ej = 0, *off = nullptr;
    gnx_cigar *ops = nullptr;
    const int rc = gnx_gsw_extend_batch(side, flat, gapPen, 1, alpha.data(), aoff, beta.data(), boff, &score, &ei, &ej, &ops, &off);
    if (rc) align::detail::raise(rc);
    DynamicAln out{score, {}, (int)ei, (int)ej};
    static const uint8_t letter[3] = {'M', 'I', 'D'};
    for (int64_t k = 0; k < off[1]; k++) out.route.push_back(cigar::Cigar{ops[k].run_length, letter[ops[k].op]});
    gnx_free(ops); gnx_free(off);
    return out;
}
} // namespace detail

// Route in traceback order, as the reference returns it for an empty incoming dynamicScore.route.
inline DynamicAln LeftDynamicAln(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const align::ScoreMatrix &scores, int64_t gapPen) {
    return detail::extend(GNX_GSW_LEFT, alpha, beta, scores, gapPen);
}
inline DynamicAln RightDynamicAln(const std::vector<dna::Base> &alpha, const std::vector<dna::Base> &beta, const align::ScoreMatrix &scores, int64_t gapPen) {
    return detail::extend(GNX_GSW_RIGHT, alpha, beta, scores, gapPen);
}
} // namespace genomeGraph
