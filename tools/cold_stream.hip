// Dev tool: what a COLD single launch can reach at all — a plain read sweep (16-byte loads, grid-stride, non-temporal or
// not) over buffers of the sizes of the benchmark matrices, timed with HIP events behind mi_flush_cache() exactly like
// bench.py's cold_single_shot, and back to back for comparison.  Gives the ceiling the cold SpMV numbers should be read against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mi355_spmv.h"

template <bool NT>
__global__ __launch_bounds__(256) void sweep(const double2* __restrict__ p, size_t n16, double* __restrict__ sink)
{
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        double2 v;
        if (NT) { v.x = __builtin_nontemporal_load(&p[i].x); v.y = __builtin_nontemporal_load(&p[i].y); }
        else v = p[i];
        s += v.x + v.y;
    }
    if (s == 123.456) sink[0] = s;
}

int main()
{
    const size_t sizes[] = {170u << 20, 660u << 20, 851u << 20};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double* sink;
    hipMalloc(&sink, 64);
    for (size_t bytes : sizes) {
        void* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
        hipMemset(buf, 1, bytes);
        for (int nt = 0; nt < 2; nt++)
            for (int grid : {2048, 8192}) {
                auto launch = [&]() {
                    if (nt) hipLaunchKernelGGL(sweep<true>, dim3(grid), dim3(256), 0, nullptr, (const double2*)buf, bytes / 16, sink);
                    else hipLaunchKernelGGL(sweep<false>, dim3(grid), dim3(256), 0, nullptr, (const double2*)buf, bytes / 16, sink);
                };
                for (int i = 0; i < 3; i++) launch();
                hipEventRecord(e0);
                for (int i = 0; i < 20; i++) launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double warm = ms * 1e3 / 20;
                double cold = 0;
                for (int i = 0; i < 10; i++) {
                    mi_flush_cache();
                    hipEventRecord(e0);
                    launch();
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                    cold += ms * 1e3 / 10;
                }
                printf("%4zu MiB  %s grid %5d: back-to-back %7.1f us (%5.2f TB/s)   cold single shot %7.1f us (%5.2f TB/s)\n", bytes >> 20,
                       nt ? "non-temporal" : "temporal    ", grid, warm, bytes / warm / 1e6, cold, bytes / cold / 1e6);
            }
        hipFree(buf);
    }
    return 0;
}
