// tools/sell_bench.hip — where spmv_bcsr4_sell's time goes (development tool; ablated variants return wrong results).
// An FE-shaped block pattern (15 blocks per block row at the Kuhn-mesh offsets of a 69^3-node box), sliced as mi_bcsr4_create does;
// times the product kernel and its ablations: no x gather, no y stores, no column stream, and combinations; waves per SIMD 2 / 4.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Inavierstokes_amd/csrc -o tools/sell_bench tools/sell_bench.hip && ./tools/sell_bench
#include "spmv_bcsr_sell.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mi355;

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

template <int D, bool NT, int ABL, int YM = 0, int NW = 4>
static double run(const SellView& S, int nwaves, const double* x, double* y, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int g = nwaves / NW;
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((spmv_bcsr4_sell<D, NT, ABL, YM, NW>), dim3(g), dim3(64 * NW), 0, nullptr, S, x, y, g);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((spmv_bcsr4_sell<D, NT, ABL, YM, NW>), dim3(g), dim3(64 * NW), 0, nullptr, S, x, y, g);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 69;
    const int nbr = m * m * m;
    const int off[15] = {-m * m - m - 1, -m * m - m, -m * m - 1, -m * m, -m - 1, -m, -1, 0, 1, m, m + 1, m * m, m * m + 1, m * m + m, m * m + m + 1};
    std::vector<int> ptrow(nbr + 1, 0), indcol;
    for (int i = 0; i < nbr; i++) {
        for (int k = 0; k < 15; k++) {
            const int c = i + off[k];
            if (c >= 0 && c < nbr) indcol.push_back(c);
        }
        ptrow[i + 1] = (int)indcol.size();
    }
    const long long nb = indcol.size();
    SellPlanHost P, P2;
    build_sell_plan(nbr, ptrow.data(), indcol.data(), 2048, P);
    build_sell_wave_ranges(P, 4096, P2.wrng, P2.nwaves);
    SellPlanHost P3;
    build_sell_wave_ranges(P, 1024, P3.wrng, P3.nwaves);
    printf("block rows %d blocks %lld steps %lld (padding %.3f %%) waves %d / %d\n", nbr, nb, P.nsteps, 100.0 * (P.nsteps * 16.0 / nb - 1), P.nwaves, P2.nwaves);
    double *val, *x, *y;
    unsigned* col;
    int *sptr, *w1, *w2;
    const size_t vb = sizeof(double) * (size_t)(P.nsteps + kSellPadSteps) * kSellStepDoubles;
    CK(hipMalloc(&val, vb));
    CK(hipMemset(val, 0, vb));
    CK(hipMalloc(&col, sizeof(unsigned) * P.col.size()));
    CK(hipMemcpy(col, P.col.data(), sizeof(unsigned) * P.col.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&sptr, sizeof(int) * P.sptr.size()));
    CK(hipMemcpy(sptr, P.sptr.data(), sizeof(int) * P.sptr.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&w1, sizeof(int) * P.wrng.size()));
    CK(hipMemcpy(w1, P.wrng.data(), sizeof(int) * P.wrng.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&w2, sizeof(int) * P2.wrng.size()));
    CK(hipMemcpy(w2, P2.wrng.data(), sizeof(int) * P2.wrng.size(), hipMemcpyHostToDevice));
    int* w3;
    CK(hipMalloc(&w3, sizeof(int) * P3.wrng.size()));
    CK(hipMemcpy(w3, P3.wrng.data(), sizeof(int) * P3.wrng.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&x, sizeof(double) * 4 * (size_t)nbr));
    CK(hipMemset(x, 0, sizeof(double) * 4 * (size_t)nbr));
    CK(hipMalloc(&y, sizeof(double) * (4 * (size_t)nbr + 2 * 4096 + 16)));
    SellView S1{val, col, sptr, w1, P.nslices, nbr}, S2{val, col, sptr, w2, P.nslices, nbr};
    SellView S3{val, col, sptr, w3, P.nslices, nbr};
    const double bytes = 132.0 * nb + 4.0 * (nbr + 1) + 64.0 * nbr;
    const int R = 40;
    auto line = [&](const char* name, double us) { printf("%-52s %8.2f us  %6.0f GB/s on the 132 B/block model  (%.3f of 8 TB/s)\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000); };
    for (int rep = 0; rep < 1; rep++) {
        line("product           D=4 nt   2 waves/SIMD", run<4, true, 0>(S1, P.nwaves, x, y, R));
        line("product           D=4 nt   4 waves/SIMD", run<4, true, 0>(S2, P2.nwaves, x, y, R));
        line("product           D=4 temporal", run<4, false, 0>(S1, P.nwaves, x, y, R));
        line("product, y stores non-temporal        D=4 nt", run<4, true, 0, 1>(S1, P.nwaves, x, y, R));
        line("product, y stores write-through (sc1) D=4 nt", run<4, true, 0, 3>(S1, P.nwaves, x, y, R));
        line("product, y parked in LDS, stored at the wave's end  D=4 nt", run<4, true, 0, 2>(S1, P.nwaves, x, y, R));
        line("product, y parked in LDS  D=6 nt", run<6, true, 0, 2>(S1, P.nwaves, x, y, R));
        line("product, y parked in LDS  D=4 nt 4 waves/SIMD", run<4, true, 0, 2>(S2, P2.nwaves, x, y, R));
        line("product, y parked in LDS  D=4 temporal", run<4, false, 0, 2>(S1, P.nwaves, x, y, R));
        line("ONE workgroup of 8 waves per CU: y parked, D=4 nt", run<4, true, 0, 2, 8>(S1, P.nwaves, x, y, R));
        line("ONE workgroup of 8 waves per CU: y parked, D=6 nt", run<6, true, 0, 2, 8>(S1, P.nwaves, x, y, R));
        line("ONE workgroup of 8 waves per CU: y parked, D=8 nt", run<8, true, 0, 2, 8>(S1, P.nwaves, x, y, R));
        line("ONE workgroup of 8 waves per CU: y direct, D=4 nt", run<4, true, 0, 0, 8>(S1, P.nwaves, x, y, R));
        line("ONE workgroup of 8 waves per CU: no y (invalid), D=4 nt", run<4, true, 2, 0, 8>(S1, P.nwaves, x, y, R));
        line("ONE wave per SIMD (1024 waves): y parked, D=8 nt", run<8, true, 0, 2, 4>(S3, P3.nwaves, x, y, R));
        line("ONE wave per SIMD (1024 waves): y parked, D=12 nt", run<12, true, 0, 2, 4>(S3, P3.nwaves, x, y, R));
        line("ONE wave per SIMD (1024 waves): y parked, D=16 nt", run<16, true, 0, 2, 4>(S3, P3.nwaves, x, y, R));
        line("ONE wave per SIMD (1024 waves): no y (invalid), D=12 nt", run<12, true, 2, 0, 4>(S3, P3.nwaves, x, y, R));
        line("no x gather       D=4 nt", run<4, true, 1>(S1, P.nwaves, x, y, R));
        line("no y stores       D=4 nt", run<4, true, 2>(S1, P.nwaves, x, y, R));
        line("no x, no y        D=4 nt", run<4, true, 3>(S1, P.nwaves, x, y, R));
        line("no x, no y, no column stream  D=4 nt", run<4, true, 7>(S1, P.nwaves, x, y, R));
        line("no x, no y, no column stream  D=4 nt 4 waves/SIMD", run<4, true, 7>(S2, P2.nwaves, x, y, R));
        line("no x, no y, no column stream  D=6 nt", run<6, true, 7>(S1, P.nwaves, x, y, R));
        line("no x, no y, no column stream  D=4 temporal", run<4, false, 7>(S1, P.nwaves, x, y, R));
        line("no x gather, y and columns kept  D=4 nt 4 waves", run<4, true, 1>(S2, P2.nwaves, x, y, R));
    }
    // when do the waves end?  one traced launch of the parked form (behind a few warm ones), and one of the direct form
    auto trace = [&](const char* name, auto launch, int nw) {
        for (int i = 0; i < 3; i++) launch();
        CK(hipDeviceSynchronize());
        launch();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> tr(2 * (size_t)nw);
        CK(hipMemcpy(tr.data(), y + 4 * (size_t)nbr, sizeof(unsigned long long) * tr.size(), hipMemcpyDeviceToHost));
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < nw; w++) if (tr[2 * w]) { lo = std::min(lo, tr[2 * w]); hi = std::max(hi, tr[2 * w + 1]); }
        int hist[16] = {0}; // loop ends in 5 us bins from the first
        double sum_store = 0;
        int cnt = 0;
        double xs[8] = {0}, xmin[8], xmax[8];
        int xc[8] = {0};
        for (int k = 0; k < 8; k++) { xmin[k] = 1e30; xmax[k] = -1e30; }
        const int per = nw / 4 / 8; // workgroups per XCD label
        for (int w = 0; w < nw; w++) if (tr[2 * w]) {
            const double e = (tr[2 * w] - lo) / 100.0;
            hist[std::min(15, (int)(e / 5))]++;
            sum_store += (double)(tr[2 * w + 1] - tr[2 * w]) / 100.0;
            cnt++;
            const int xcd = (w / 4) / per; // logical workgroup (w / 4) = xcd * per + pos
            xs[xcd] += e; xc[xcd]++; xmin[xcd] = std::min(xmin[xcd], e); xmax[xcd] = std::max(xmax[xcd], e);
        }
        printf("%s: waves %d, first loop end -> last store %.2f us; mean (loop end -> stores out) %.2f us; loop ends per 5 us from the first:", name, cnt, (hi - lo) / 100.0, sum_store / cnt);
        for (int b = 0; b < 16; b++) printf(" %d", hist[b]);
        // who is late?  by wave number inside its workgroup, and by the workgroup's place in its XCD's dispatch order (first / second half)
        double byw[4] = {0}, byh[2] = {0};
        int cw[4] = {0}, ch[2] = {0};
        for (int w = 0; w < nw; w++) if (tr[2 * w]) {
            const double e = (tr[2 * w] - lo) / 100.0;
            byw[w & 3] += e; cw[w & 3]++;
            const int pos = (w / 4) % per; // place among the XCD label's workgroups: physical workgroup = pos * 8 + xcd
            byh[pos >= per / 2] += e; ch[pos >= per / 2]++;
        }
        printf("\n    mean loop end by wave-in-workgroup: %.1f %.1f %.1f %.1f; by dispatch half (blockIdx < grid/2 | >=): %.1f | %.1f",
               byw[0] / std::max(1, cw[0]), byw[1] / std::max(1, cw[1]), byw[2] / std::max(1, cw[2]), byw[3] / std::max(1, cw[3]), byh[0] / std::max(1, ch[0]), byh[1] / std::max(1, ch[1]));
        printf("\n    per XCD label (mean / min / max loop end, us):");
        for (int k = 0; k < 8; k++) printf("  %.1f/%.1f/%.1f", xs[k] / std::max(1, xc[k]), xmin[k], xmax[k]);
        printf("\n");
    };
    const int g1 = P.nwaves / 4;
    trace("TRACE parked", [&] { hipLaunchKernelGGL((spmv_bcsr4_sell<4, true, 8, 2>), dim3(g1), dim3(256), 0, nullptr, S1, x, y, g1); }, P.nwaves);
    trace("TRACE parked, final stores skipped", [&] { hipLaunchKernelGGL((spmv_bcsr4_sell<4, true, 8 + 16, 2>), dim3(g1), dim3(256), 0, nullptr, S1, x, y, g1); }, P.nwaves);
    const int g8 = P.nwaves / 8;
    trace("TRACE parked, one workgroup of 8 waves per CU", [&] { hipLaunchKernelGGL((spmv_bcsr4_sell<4, true, 8, 2, 8>), dim3(g8), dim3(512), 0, nullptr, S1, x, y, g8); }, P.nwaves);
    const int g3 = P3.nwaves / 4;
    trace("TRACE parked, one wave per SIMD, D=12", [&] { hipLaunchKernelGGL((spmv_bcsr4_sell<12, true, 8, 2, 4>), dim3(g3), dim3(256), 0, nullptr, S3, x, y, g3); }, P3.nwaves);
    trace("TRACE direct", [&] { hipLaunchKernelGGL((spmv_bcsr4_sell<4, true, 8, 0>), dim3(g1), dim3(256), 0, nullptr, S1, x, y, g1); }, P.nwaves);
    line("parked, final stores skipped (invalid)", run<4, true, 16, 2>(S1, P.nwaves, x, y, R));
    { // is it the same waves that end late, launch after launch?  correlation of the per-wave loop ends of traced launches (one wave per SIMD, D = 8)
        std::vector<std::vector<double>> ends;
        for (int rep = 0; rep < 4; rep++) {
            for (int i = 0; i < 3; i++) hipLaunchKernelGGL((spmv_bcsr4_sell<8, true, 0, 2, 4>), dim3(g3), dim3(256), 0, nullptr, S3, x, y, g3);
            hipLaunchKernelGGL((spmv_bcsr4_sell<8, true, 8, 2, 4>), dim3(g3), dim3(256), 0, nullptr, S3, x, y, g3);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> tr(2 * (size_t)P3.nwaves);
            CK(hipMemcpy(tr.data(), y + 4 * (size_t)nbr, sizeof(unsigned long long) * tr.size(), hipMemcpyDeviceToHost));
            unsigned long long lo = ~0ull;
            for (int w = 0; w < P3.nwaves; w++) lo = std::min(lo, tr[2 * w]);
            std::vector<double> e(P3.nwaves);
            for (int w = 0; w < P3.nwaves; w++) e[w] = (tr[2 * w] - lo) / 100.0;
            ends.push_back(e);
        }
        auto corr = [&](const std::vector<double>& a, const std::vector<double>& b) {
            double ma = 0, mb = 0; const int n = (int)a.size();
            for (int i = 0; i < n; i++) { ma += a[i]; mb += b[i]; }
            ma /= n; mb /= n;
            double sab = 0, saa = 0, sbb = 0;
            for (int i = 0; i < n; i++) { sab += (a[i] - ma) * (b[i] - mb); saa += (a[i] - ma) * (a[i] - ma); sbb += (b[i] - mb) * (b[i] - mb); }
            return sab / sqrt(saa * sbb + 1e-30);
        };
        printf("per-wave loop-end correlation between traced launches (launched behind 3 untraced ones): 0-1 %.3f, 0-2 %.3f, 1-3 %.3f, 2-3 %.3f\n", corr(ends[0], ends[1]), corr(ends[0], ends[2]), corr(ends[1], ends[3]), corr(ends[2], ends[3]));
        // by CU (4 waves of a workgroup) and by XCD label: mean end per group in launch 0 vs launch 1
        double x0[8] = {0}, x1[8] = {0};
        for (int w = 0; w < P3.nwaves; w++) { const int xcd = (w / 4) / (P3.nwaves / 4 / 8); x0[xcd] += ends[0][w] / (P3.nwaves / 8); x1[xcd] += ends[1][w] / (P3.nwaves / 8); }
        printf("mean loop end per XCD label, launch 0:"); for (int k = 0; k < 8; k++) printf(" %.1f", x0[k]);
        printf("   launch 1:"); for (int k = 0; k < 8; k++) printf(" %.1f", x1[k]);
        printf("\n");
    }
    return 0;
}
