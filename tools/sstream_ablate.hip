// tools/sstream_ablate.hip — dev tool (round 4): the LIBRARY's sliced-stream kernel (navierstokes_amd/csrc/spmv_sstream.hpp, its planner and
// its kernel) on an S15-like band, with the kernel's ablation bits (invalid results): what each part of the kernel costs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Inavierstokes_amd/csrc -o tools/sstream_ablate tools/sstream_ablate.hip && ./tools/sstream_ablate [rows] [w] [per]
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

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 5000000, w = argc > 2 ? atoi(argv[2]) : 2000, per = argc > 3 ? atoi(argv[3]) : 15;
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
    const int R = 50;
    CK(hipMemset(d_y, 0xff, sizeof(double) * n));
    line("library kernel  D=8 nt", run<8, true, 0>(S, d_x, d_y, R));
    {
        std::vector<double> hy(n);
        CK(hipMemcpy(hy.data(), d_y, sizeof(double) * n, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (int i = 0; i < n; i++) bad += memcmp(&hy[i], &href[i], 8) != 0;
        printf("    %lld of %d rows differ bitwise from the host's fma chain\n", bad, n);
    }
    line("library kernel  D=12 nt", run<12, true, 0>(S, d_x, d_y, R));
    line("  no boundary loads / refills (invalid)  D=8 nt", run<8, true, 4>(S, d_x, d_y, R));
    line("  no boundary loads / refills (invalid)  D=12 nt", run<12, true, 4>(S, d_x, d_y, R));
    line("  no y stores (invalid)  D=8 nt", run<8, true, 2>(S, d_x, d_y, R));
    line("  no boundary loads, no y stores (invalid)  D=8 nt", run<8, true, 6>(S, d_x, d_y, R));
    line("  none of the three (invalid)  D=8 nt", run<8, true, 7>(S, d_x, d_y, R));
    line("library kernel  D=8 nt (again)", run<8, true, 0>(S, d_x, d_y, R));
    return 0;
}
