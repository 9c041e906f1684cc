// tools/sstream_trace.hip — dev tool (round 5): WHERE the library's sliced-stream kernel (navierstokes_amd/csrc/spmv_sstream.hpp) spends a
// small launch: s_memrealtime stamps per workgroup (start, in front of its loop, behind it, end) on an S15-like band.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Inavierstokes_amd/csrc -o tools/sstream_trace tools/sstream_trace.hip && ./tools/sstream_trace [rows] [w] [per]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "spmv_sstream.hpp"
using namespace mi355;
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static unsigned long long rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }


template <int D, bool NT, int ABL>
static double run(const SsView& S, const double* x, double* y, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((spmv_sstream<D, NT, ABL>), dim3(S.nwg), dim3(256), 0, nullptr, S, x, y);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((spmv_sstream<D, NT, ABL>), dim3(S.nwg), dim3(256), 0, nullptr, S, x, y);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1e3 / reps;
}

static double pct(std::vector<double> v, double q)
{
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, (size_t)(q * (v.size() - 1) + 0.5))];
}

template <int D, bool NT>
static void trace(const char* name, SsView S, const SsPlanHost& P, const double* x, double* y)
{
    unsigned long long* d_tr;
    const int G = S.nwg, L = 20;
    CK(hipMalloc(&d_tr, sizeof(unsigned long long) * 4 * G));
    S.trace = d_tr;
    std::vector<unsigned long long> h(4 * (size_t)G);
    std::vector<double> start, fill, loop, tail, endt, total, end_by[64];
    for (int it = 0; it < L + 3; it++) {
        hipLaunchKernelGGL((spmv_sstream<D, NT, 8>), dim3(G), dim3(256), 0, nullptr, S, x, y);
        CK(hipDeviceSynchronize());
        if (it < 3) continue;
        CK(hipMemcpy(h.data(), d_tr, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int g = 0; g < G; g++) { t0 = std::min(t0, h[4 * g]); t1 = std::max(t1, h[4 * g + 3]); }
        total.push_back((t1 - t0) * 0.01);
        for (int g = 0; g < G; g++) {
            start.push_back((h[4 * g] - t0) * 0.01);
            fill.push_back((h[4 * g + 1] - h[4 * g]) * 0.01);
            loop.push_back((h[4 * g + 2] - h[4 * g + 1]) * 0.01);
            tail.push_back((h[4 * g + 3] - h[4 * g + 2]) * 0.01);
            endt.push_back((h[4 * g + 3] - t0) * 0.01);
            const int nr = P.rptr[g + 1] - P.rptr[g];
            if (nr < 64) end_by[nr].push_back((h[4 * g + 3] - t0) * 0.01);
        }
    }
    printf("%s: %d workgroups, %d traced launches (one per synchronise); microseconds, s_memrealtime (10 ns)\n", name, G, L);
    printf("  first start -> last end             median %6.2f  min %6.2f  max %6.2f\n", pct(total, 0.5), pct(total, 0), pct(total, 1));
    printf("  start after the first workgroup's   median %6.2f  p90 %6.2f  max %6.2f\n", pct(start, 0.5), pct(start, 0.9), pct(start, 1));
    printf("  start -> loop (first window fill)   median %6.2f  p90 %6.2f  max %6.2f\n", pct(fill, 0.5), pct(fill, 0.9), pct(fill, 1));
    printf("  loop (wave 0)                       median %6.2f  p10 %6.2f  p90 %6.2f  max %6.2f\n", pct(loop, 0.5), pct(loop, 0.1), pct(loop, 0.9), pct(loop, 1));
    printf("  loop end -> all stores drained      median %6.2f  p90 %6.2f  max %6.2f\n", pct(tail, 0.5), pct(tail, 0.9), pct(tail, 1));
    printf("  end after the first start           p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f\n", pct(endt, 0.1), pct(endt, 0.5), pct(endt, 0.9), pct(endt, 1));
    for (int nr = 0; nr < 64; nr++)
        if (!end_by[nr].empty())
            printf("    workgroups of %2d rounds (%4zu per launch): end p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f\n", nr, end_by[nr].size() / L, pct(end_by[nr], 0.1),
                   pct(end_by[nr], 0.5), pct(end_by[nr], 0.9), pct(end_by[nr], 1));
    CK(hipFree(d_tr));
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, w = argc > 2 ? atoi(argv[2]) : 2000, per = argc > 3 ? atoi(argv[3]) : 15;
    std::vector<int> ptrow(n + 1, 0), indcol;
    std::vector<double> coef;
    indcol.reserve((size_t)n * per);
    coef.reserve((size_t)n * per);
    std::vector<int> cols;
    for (int i = 0; i < n; i++) {
        cols.assign(1, i);
        while ((int)cols.size() < per) {
            const int c = i - w + (int)(rnd() % (2 * w + 1));
            if (c < 0 || c >= n) continue;
            if (std::find(cols.begin(), cols.end(), c) == cols.end()) cols.push_back(c);
        }
        std::sort(cols.begin(), cols.end());
        for (int c : cols) {
            indcol.push_back(c);
            coef.push_back(c == i ? 1.0 : ((double)(rnd() >> 11) / 9007199254740992.0 * 2 - 1) / per);
        }
        ptrow[i + 1] = (int)indcol.size();
    }
    const long long nnz = indcol.size();
    SsPlanHost P;
    build_sstream_plan(n, n, ptrow.data(), indcol.data(), 0.12, P);
    printf("n %d nnz %lld  eligible %d (%s)  workgroups %d rounds %d steps %lld\n", n, nnz, (int)P.eligible, P.why, P.nwg, P.rounds, P.steps);
    if (!P.eligible) return 1;
    int* d_ptrow;
    double *d_coef, *d_x, *d_y;
    CK(hipMalloc(&d_ptrow, sizeof(int) * (n + 1)));
    CK(hipMemcpy(d_ptrow, ptrow.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_coef, sizeof(double) * nnz));
    CK(hipMemcpy(d_coef, coef.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
    SsDevice Dv;
    CK(ss_upload(P, Dv, false));
    sstream_fill_values(P.rounds, n, 0, d_ptrow, d_coef, nullptr, Dv.slice_step, Dv.slice_len, Dv.val, P.max_slice_nnz, nullptr);
    CK(hipGetLastError());
    std::vector<double> hx(n), href(n);
    for (int i = 0; i < n; i++) hx[i] = sin(0.001 * i);
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) s = fma(coef[k], hx[indcol[k]], s);
        href[i] = s;
    }
    CK(hipMalloc(&d_x, sizeof(double) * n));
    CK(hipMemcpy(d_x, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_y, sizeof(double) * (n + 2)));
    SsView S{Dv.val, Dv.slot, Dv.wg, Dv.win, P.nwg, n, n, nullptr, 0};
    const double B = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    auto line = [&](const char* name, double us) { printf("%-56s %8.2f us   %6.0f GB/s algorithmic  (%.3f of 8 TB/s)\n", name, us, B / us / 1e3, B / us / 1e3 / 8000); fflush(stdout); };
    const int R = 200;
    CK(hipMemset(d_y, 0xff, sizeof(double) * n));
    line("library kernel  D=8 temporal, back to back", run<8, false, 0>(S, d_x, d_y, R));
    {
        std::vector<double> hy(n);
        CK(hipMemcpy(hy.data(), d_y, sizeof(double) * n, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (int i = 0; i < n; i++) bad += memcmp(&hy[i], &href[i], 8) != 0;
        printf("    %lld of %d rows differ bitwise from the host's fma chain\n", bad, n);
    }
    line("library kernel  D=8 nt, back to back", run<8, true, 0>(S, d_x, d_y, R));
    line("library kernel  D=12 temporal, back to back", run<12, false, 0>(S, d_x, d_y, R));
    line("  with the stamps compiled in  D=8 temporal", run<8, false, 8>(([&] { SsView T = S; unsigned long long* t; CK(hipMalloc(&t, 32 * (size_t)S.nwg)); T.trace = t; return T; })(), d_x, d_y, R));
    trace<8, false>("D=8 temporal", S, P, d_x, d_y);
    { // the same loop with every byte served by the XCDs' L2s: every workgroup runs workgroup 0's rounds (the same ~0.6 MB of the stream;
      // all of them store the same rows of y with the same values).  What a k-step that re-read its matrix slices from L2 would stream at.
        SsPlanHost Q = P;
        for (int g = 0; g < Q.nwg; g++) Q.wg[g] = P.wg[0];
        for (int g = 0; g <= Q.nwg; g++) Q.rptr[g] = g == 0 ? P.rptr[0] : P.rptr[1]; // (for the trace's grouping by round count only)
        SsWg* d_wg2;
        CK(hipMalloc(&d_wg2, sizeof(SsWg) * Q.wg.size()));
        CK(hipMemcpy(d_wg2, Q.wg.data(), sizeof(SsWg) * Q.wg.size(), hipMemcpyHostToDevice));
        SsView S2 = S;
        S2.wg = d_wg2;
        const int rounds0 = P.rptr[1] - P.rptr[0];
        const double us = run<8, false, 0>(S2, d_x, d_y, R);
        printf("ALIASED to workgroup 0's %d rounds (matrix stream from L2): %.2f us per launch back to back\n", rounds0, us);
        trace<8, false>("D=8 temporal, every workgroup on workgroup 0's rounds (L2-served)", S2, Q, d_x, d_y);
    }
    trace<12, false>("D=12 temporal", S, P, d_x, d_y);
    trace<8, true>("D=8 nt", S, P, d_x, d_y);
    return 0;
}
