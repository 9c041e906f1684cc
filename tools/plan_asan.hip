// tools/plan_asan.hip — dev tool: the sliced stream's two planners (spmv_sstream.hpp, spmv_sstream_mw.hpp) and their host replays on random
// multi-band patterns (1-5 bands, ragged ends, dropped entries, empty rows, both row shifts) under the HOST address and undefined-behaviour
// sanitizers (GPU sanitizers are not available on this pool; nothing here launches a kernel):
//   hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -Inavierstokes_amd/csrc -Iinclude -o /tmp/plan_asan tools/plan_asan.hip
//   ASAN_OPTIONS=detect_leaks=0 /tmp/plan_asan
#include "spmv_sstream_mw.hpp"
#include <cstdio>
#include <random>
using namespace mi355;
int main()
{
    std::mt19937 rng(7);
    int bad = 0, elig = 0, total = 0;
    for (int it = 0; it < 60; it++) {
        const int n = 2000 + (int)(rng() % 150000);
        const int kb = 1 + (int)(rng() % 5);
        const int gap = 600 + (int)(rng() % 30000);
        const int wid = 1 + (int)(rng() % 6), st = 1 + (int)(rng() % 60);
        std::vector<int> p(1, 0), c;
        for (int i = 0; i < n; i++) {
            for (int b = 0; b < kb; b++)
                for (int w = 0; w < wid; w++) {
                    const long long col = (long long)i + (long long)(b - kb / 2) * gap + (long long)w * st;
                    if (col >= 0 && col < n && (rng() % 100) >= 4) c.push_back((int)col);
                }
            if (rng() % 997 == 0) c.resize(p.back()); // an empty row now and then
            p.push_back((int)c.size());
        }
        for (int shift = 0; shift < 2; shift++) {
            SsMwPlanHost P;
            build_sstream_mw_plan(n, n, p.data(), c.data(), 0.5, P, shift);
            total++;
            if (!P.eligible) continue;
            elig++;
            if (const char* why = check_sstream_mw_plan(P, n, p.data(), c.data())) { printf("it %d n %d kb %d gap %d shift %d: %s\n", it, n, kb, gap, shift, why); bad++; }
        }
        SsPlanHost Q; // the one-window planner on the same pattern
        build_sstream_plan(n, n, p.data(), c.data(), 0.5, Q);
        if (Q.eligible)
            if (const char* why = check_sstream_plan(Q, n, p.data(), c.data())) { printf("classic it %d: %s\n", it, why); bad++; }
    }
    printf("plans %d eligible %d bad %d\n", total, elig, bad);
    return bad != 0;
}
