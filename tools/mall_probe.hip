// Dev tool: what survives in the 256 MB Infinity Cache when a large array streams through it?  The product's warm launches differ by
// 15 % with the placement of their arrays (profiles/NOTES.md §4.12); this isolates the mechanism with plain read sweeps:
//   R = a "re-used" buffer (r MB), S = a "streamed" buffer (s MB).  Loop: read R, read S with load flavour m, then TIME the next read
//   of R.  R served from HBM: r / ~5.5 TB/s; R still cached: faster.  Flavours of the S loads: plain, nt, sc1, sc0 sc1, sc0 sc1 nt,
//   sc0 nt, sc1 nt.  Several fresh allocations of R and S per configuration (placement).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ double2 ld(const double2* p)
{
    double2 v;
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// every workgroup reads a contiguous chunk, 8 loads in flight per thread (independent asm blocks would serialise on the waitcnt
// above, so the loads of one trip are issued through the compiler instead when MODE < 0 ... keep it simple: unroll by hand)
template <int MODE>
__global__ __launch_bounds__(256) void sweep(const double2* __restrict__ p, size_t n16, double* __restrict__ sink)
{
    double s = 0.0;
    const size_t per_wg = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per_wg, hi = lo + per_wg < n16 ? lo + per_wg : n16;
    for (size_t i = lo + threadIdx.x; i < hi; i += 8 * 256) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t k = i + (size_t)u * 256 < hi ? i + (size_t)u * 256 : hi - 1;
            const double2* q = p + k;
            if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(q) : "memory");
            if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[u]) : "v"(q) : "memory");
            if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(q) : "memory");
            if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(q) : "memory");
            if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v[u]) : "v"(q) : "memory");
            if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v[u]) : "v"(q) : "memory");
            if (MODE == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v[u]) : "v"(q) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u].x + v[u].y;
    }
    if (s == 123.456) sink[0] = s;
}

static void launch(int mode, const void* buf, size_t bytes, double* sink)
{
    const dim3 g(2048), b(256);
    const double2* p = (const double2*)buf;
    const size_t n16 = bytes / 16;
    switch (mode) {
    case 0: hipLaunchKernelGGL(sweep<0>, g, b, 0, nullptr, p, n16, sink); break;
    case 1: hipLaunchKernelGGL(sweep<1>, g, b, 0, nullptr, p, n16, sink); break;
    case 2: hipLaunchKernelGGL(sweep<2>, g, b, 0, nullptr, p, n16, sink); break;
    case 3: hipLaunchKernelGGL(sweep<3>, g, b, 0, nullptr, p, n16, sink); break;
    case 4: hipLaunchKernelGGL(sweep<4>, g, b, 0, nullptr, p, n16, sink); break;
    case 5: hipLaunchKernelGGL(sweep<5>, g, b, 0, nullptr, p, n16, sink); break;
    default: hipLaunchKernelGGL(sweep<6>, g, b, 0, nullptr, p, n16, sink); break;
    }
}

int main(int argc, char** argv)
{
    const char* names[7] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0 nt", "sc1 nt"};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double* sink;
    CK(hipMalloc(&sink, 64));
    const size_t MB = 1u << 20;
    auto timed_R = [&](const void* R, size_t r, const void* S, size_t s, int mode_s, int mode_r, float* us_r, float* us_s) -> int {
        // a few cycles to reach the steady state, then time R and S of the last ones
        std::vector<float> tr, tsv;
        for (int it = 0; it < 8; it++) {
            CK(hipEventRecord(e0, nullptr));
            launch(mode_r, R, r, sink);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 3) tr.push_back(ms * 1e3f);
            if (s > 0) {
                CK(hipEventRecord(e0, nullptr));
                launch(mode_s, S, s, sink);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 3) tsv.push_back(ms * 1e3f);
            }
        }
        std::sort(tr.begin(), tr.end());
        std::sort(tsv.begin(), tsv.end());
        *us_r = tr[tr.size() / 2];
        *us_s = tsv.empty() ? 0.f : tsv[tsv.size() / 2];
        return 0;
    };
    const int draws = argc > 1 ? atoi(argv[1]) : 3;
    for (size_t r_mb : {64u, 128u, 200u, 240u}) {
        for (int d = 0; d < draws; d++) {
            void *R = nullptr, *S = nullptr;
            const size_t r = r_mb * MB, s = 700 * MB;
            CK(hipMalloc(&R, r));
            CK(hipMalloc(&S, s));
            CK(hipMemset(R, 1, r));
            CK(hipMemset(S, 1, s));
            float alone, dummy;
            if (timed_R(R, r, nullptr, 0, 0, 0, &alone, &dummy)) return 1;
            printf("R = %3zu MB (draw %d): re-read alone %6.1f us (%5.2f TB/s) | after 700 MB of S:", r_mb, d, alone, r / alone * 1e-6);
            for (int m = 0; m < 7; m++) {
                float ur, us;
                if (timed_R(R, r, S, s, m, 0, &ur, &us)) return 1;
                printf("  [%s] R %5.1f S %5.1f", names[m], ur, us);
            }
            printf("\n");
            fflush(stdout);
            CK(hipFree(R));
            CK(hipFree(S));
        }
    }
    // and R itself read with a flavour (does the flavour of the RE-USED data matter?)
    {
        void *R = nullptr, *S = nullptr;
        const size_t r = 200 * MB, s = 700 * MB;
        CK(hipMalloc(&R, r));
        CK(hipMalloc(&S, s));
        CK(hipMemset(R, 1, r));
        CK(hipMemset(S, 1, s));
        printf("R = 200 MB read with a flavour, S = 700 MB nt:");
        for (int m = 0; m < 7; m++) {
            float ur, us;
            if (timed_R(R, r, S, s, 1, m, &ur, &us)) return 1;
            printf("  [%s] R %5.1f", names[m], ur);
        }
        printf("\n");
    }
    return 0;
}
