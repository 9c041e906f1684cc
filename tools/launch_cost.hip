// launch_cost.hip — where do the ~5.4 us of fixed cost per ring-kernel launch go?  Back-to-back launches of
// (a) an empty kernel of the same shape, (b) + plan load and the two barriers, (c) + first-window fill and the
// prologue's stream loads (one block), on 512 workgroups x 256 threads x 79 KB LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const int4* plan, const double* x, const double* coef, double* sink)
{
    __shared__ double ring[5120];
    __shared__ double stage[2 * 2113];
    __shared__ int4 s_plan[332];
    const int tid = threadIdx.x;
    if (MODE == 0) { if (sink == nullptr) stage[tid] = ring[tid]; return; }
    for (int i = tid; i < 30; i += 256) s_plan[i] = plan[blockIdx.x * 30 + i];
    __syncthreads();
    if (tid < 6) s_plan[30 + tid] = s_plan[28];
    __syncthreads();
    double acc = 0;
    if (MODE >= 2) {
        const int c0 = s_plan[1].x + tid;
        double v[20];
#pragma unroll
        for (int u = 0; u < 20; u++) v[u] = x[c0 + u * 256];
        double c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = coef[(size_t)s_plan[0].y + tid + i * 256];
#pragma unroll
        for (int u = 0; u < 20; u++) ring[tid + u * 256] = v[u];
#pragma unroll
        for (int i = 0; i < 8; i++) acc += c[i];
        __syncthreads();
        acc += ring[(tid * 7) % 5120];
    }
    if (acc == 123.456) sink[0] = acc + s_plan[tid % 30].x;
}

int main()
{
    const int wgs = 512;
    int4* plan; double *x, *coef, *sink;
    CK(hipMalloc(&plan, sizeof(int4) * wgs * 30)); CK(hipMemset(plan, 0, sizeof(int4) * wgs * 30));
    CK(hipMalloc(&x, 8 * 8000000)); CK(hipMemset(x, 0, 8 * 8000000));
    CK(hipMalloc(&coef, 8 * 20000000)); CK(hipMemset(coef, 0, 8 * 20000000));
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* name) {
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, plan, x, coef, sink);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 500; i++) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, plan, x, coef, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("LAUNCH %-60s %6.2f us per launch\n", name, ms * 1e3 / 500);
    };
    run(k<0>, "empty kernel, 512 x 256 threads, 79 KB LDS");
    run(k<1>, "+ plan load (30 int4) and two barriers");
    run(k<2>, "+ first window (20 loads/thread) + first block's 8 value loads");
    return 0;
}
