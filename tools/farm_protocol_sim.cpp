// CPU simulation of the walk farm's round protocol (csrc/farm64.hip.h: farm_predict + the set handling of the overlapped rounds) on a noisy diagonal
// path of a 1 Mb x 1 Mb pair: rounds needed with plain and overlapped rounds.  Not product code and not the oracle: a model used to find and fix the
// first overlapped protocol's collapse after a wrong guess (profiles/r5_experiments.md section 10).
//   g++ -O2 -o /tmp/farm_sim tools/farm_protocol_sim.cpp && /tmp/farm_sim <tiles per round> <overlapped 0/1> [1 = the first protocol: skip the other set's tiles]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <vector>
#include <functional>
using namespace std;
struct int2 { int x, y; };
static int2 make_int2(int a, int b) { return {a, b}; }
constexpr int R = 10, H64 = 640, CK64 = 128, CKC64 = 224;
template <bool AFF> struct Geo {
    static constexpr int CK = AFF ? CK64 : CKC64;
    static int block_of(int te) { return AFF ? (te >= 3 ? (te - 3) / CK : 0) : (te - 1) / CK; }
    static int tmin_of(int c) { return AFF ? (c > 0 ? 2 : 0) : 0; }
};
template <bool AFF, typename Skip>
int farm_predict(int2 *tile, int i, int j, const int virt, const int nt, const bool store, const int da, const int db, Skip skip) {
    using G = Geo<AFF>;
    float fa = 1.0f, fb = 1.0f;
    if (da > 0 || db > 0) { const float mx = (float)max(da, db); fa = (float)da / mx; fb = (float)db / mx; }
    const float den = fb + fa * (1.0f / R);
    int n = 0, ps = -1, pc = -1; bool taking = false; extern bool g_old_rule;
    for (int guard = 0; guard < 6 * nt && n < nt && i > 0 && j > 0 && !(virt > 0 && i <= virt); guard++) {
        const int s = (i - 1) / H64, i0 = i - 1 - s * H64, lw = i0 / R, te = j + lw;
        const int c = G::block_of(te), tbeg = c * G::CK, tmin = G::tmin_of(c);
        if (s == ps && c == pc) { if (fb >= fa) j -= 2; else i -= 2; continue; }
        ps = s; pc = c;
        if (taking || !skip(s, c)) { taking = !g_old_rule; if (store) tile[n] = make_int2(s, c); n++; }
        const float xl = (float)(te - 1 - tbeg - tmin + 1) / den;
        const float xt = fa > 0.0f ? (float)(i0 + 1) / fa : 3.0e9f;
        const float xc = fb > 0.0f ? (float)j / fb : 3.0e9f;
        const float x = fminf(xl, fminf(xt, xc));
        int di = (int)(fa * x + 0.999f), dj = (int)(fb * x + 0.999f);
        if (di + dj == 0) { di = fa >= fb; dj = fb > fa; }
        i -= min(di, i0 + 1); j -= dj;
    }
    return n;
}
bool g_old_rule = false;
int main(int argc, char **argv) {
    g_old_rule = argc > 3 && atoi(argv[3]) != 0;
    const int nt = argc > 1 ? atoi(argv[1]) : 16;
    const bool pipe = argc > 2 ? atoi(argv[2]) : 1;
    int i = 1000000, j = 999886;
    int2 L[2][32]; int n[2];
    n[0] = farm_predict<true>(L[0], i, j, 0, nt, true, 0, 0, [](int, int) { return false; });
    n[1] = farm_predict<true>(L[1], i, j, 0, nt, true, 0, 0, [&](int s, int c) { for (int x = 0; x < n[0]; x++) if (L[0][x].x == s && L[0][x].y == c) return true; return false; });
    int rounds = 0, acc_i = 0, acc_j = 0; long tiles = 0;
    unsigned rng = 12345;
    for (int r = 0; i > 0 && j > 0 && r < 100000; r++) {
        const int par = pipe ? (r & 1) : 0;
        const int i_in = i, j_in = j;
        // walk: cell by cell along the diagonal with rare indels
        int ls = -1, lc = -1;
        while (i > 0 && j > 0) {
            const int s = (i - 1) / H64, lw = (i - 1 - s * H64) / R, c = Geo<true>::block_of(j + lw);
            if (s != ls || c != lc) {
                bool found = false;
                for (int x = 0; x < n[par]; x++) if (L[par][x].x == s && L[par][x].y == c) found = true;
                if (!found) break;
                ls = s; lc = c; tiles++;
            }
            rng = rng * 1664525u + 1013904223u;
            const unsigned u = rng >> 8;
            if (u % 500 == 0) i--; else if (u % 500 == 1) j--; else { i--; j--; }
        }
        rounds++;
        const int da = acc_i / 2 + (i_in - i), db = acc_j / 2 + (j_in - j);
        n[par] = farm_predict<true>(L[par], i, j, 0, nt, true, da, db, [&](int s, int c) { if (!pipe) return false; for (int x = 0; x < n[par ^ 1]; x++) if (L[par ^ 1][x].x == s && L[par ^ 1][x].y == c) return true; return false; });
        acc_i = da; acc_j = db;
        if (r < 0) { printf("round %d: at (%d,%d) da %d db %d next set:", r, i, j, da, db); for (int x = 0; x < n[par]; x++) printf(" (%d,%d)", L[par][x].x, L[par][x].y); printf("\n"); }
    }
    printf("nt %d pipe %d: rounds %d tiles %ld  end (%d,%d)\n", nt, (int)pipe, rounds, tiles, i, j);
}
