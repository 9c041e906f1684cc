// tools/part_asan.cpp — dev tool: partition.hpp (PartPlan::build, build_combined, build_all_ext, the 4x4 pattern test) under the host address and
// undefined-behaviour sanitizers on random banded / multi-band / node-blocked patterns and random cuts:
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Inavierstokes_amd/csrc -o /tmp/part_asan tools/part_asan.cpp && /tmp/part_asan
#include "partition.hpp"
#include <cstdio>
#include <random>
using namespace mi355;
int main()
{
    std::mt19937 rng(11);
    int bad = 0, plans = 0;
    for (int it = 0; it < 80; it++) {
        const bool blocked = it % 3 == 0;
        const int nn = 300 + (int)(rng() % 6000), n = blocked ? 4 * nn : nn;
        const int R = 1 + (int)(rng() % 6);
        const int kb = 1 + (int)(rng() % 3), gap = 20 + (int)(rng() % (nn / 3 + 1)), wid = 1 + (int)(rng() % 4);
        // node-level pattern, then expanded 4x4 for the blocked case
        std::vector<std::vector<int>> rows((size_t)nn);
        for (int i = 0; i < nn; i++)
            for (int b = 0; b < kb; b++)
                for (int w = 0; w < wid; w++) {
                    const int c = i + (b - kb / 2) * gap + w;
                    if (c >= 0 && c < nn && (rng() % 100) >= 3) rows[i].push_back(c);
                }
        std::vector<int> P(1, 0), C;
        std::vector<double> V;
        for (int i = 0; i < nn; i++)
            for (int q = 0; q < (blocked ? 4 : 1); q++) {
                for (int c : rows[i])
                    for (int k = 0; k < (blocked ? 4 : 1); k++) { C.push_back(blocked ? 4 * c + k : c); V.push_back(1.0 + C.size()); }
                P.push_back((int)C.size());
            }
        std::vector<long long> rs(R + 1, 0);
        for (int p = 1; p < R; p++) { long long cut = (long long)n * p / R + (long long)(rng() % 7) - 3; if (blocked) cut -= cut % 4; rs[p] = std::max(rs[p - 1], std::min<long long>(cut, n)); }
        rs[R] = n;
        for (int rank = 0; rank < R; rank++) {
            const int lo = (int)rs[rank], hi = (int)rs[rank + 1];
            std::vector<int> p(P.begin() + lo, P.begin() + hi + 1), c(C.begin() + P[lo], C.begin() + P[hi]);
            std::vector<double> v(V.begin() + P[lo], V.begin() + P[hi]);
            for (int& e : p) e -= P[lo];
            PartPlan pl;
            const std::string err = pl.build(R, rank, rs.data(), p.data(), c.data(), v.data());
            if (!err.empty()) { printf("it %d rank %d: %s\n", it, rank, err.c_str()); bad++; continue; }
            plans++;
            pl.build_combined();
            pl.build_all_ext();
            // every nonzero of the rank must reappear in both combined pieces, in the caller's order
            if (pl.all.indcol.size() != c.size() || pl.all_ext.indcol.size() != c.size()) { printf("it %d rank %d: combined piece lost nonzeros\n", it, rank); bad++; continue; }
            for (size_t k = 0; k < c.size(); k++) {
                const int e = pl.all_ext.indcol[k];
                const long long g = e < pl.n_local ? lo + e : pl.halo_ids[e - pl.n_local];
                if (g != c[k] || pl.all_ext.coef[k] != v[k]) { printf("it %d rank %d: all_ext nonzero %zu names column %lld, caller's %d\n", it, rank, k, g, c[k]); bad++; break; }
            }
            if (blocked && pl.n_local % 4 == 0 && pl.n_halo % 4 == 0 && !csr_has_block4_pattern(pl.n_local, pl.all_ext.ptrow.data(), pl.all_ext.indcol.data()) && pl.n_local > 0) {
                printf("it %d rank %d: the [owned | halo] piece of a node-blocked matrix lost its 4x4 structure\n", it, rank);
                bad++;
            }
        }
    }
    printf("partition plans %d bad %d\n", plans, bad);
    return bad != 0;
}
