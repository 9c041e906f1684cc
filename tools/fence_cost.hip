// fence_cost.hip — what does an agent-scope release / acquire pair cost a workgroup on MI355X
// (eight XCD-private L2s: release = write back dirty L2 lines, acquire = invalidate)?  Sizes the
// inter-workgroup hand-off a single-launch matrix-powers kernel would need (profiles/NOTES.md §4.6).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE> // 0: stores only, 1: + release fence, 2: + release + flag store + acquire load of the neighbour's flag + acquire fence
__global__ __launch_bounds__(256) void k(double* y, int* flags, int rounds, int rows_per_round)
{
    __shared__ double hog[9000];
    hog[threadIdx.x] = 0;
    const int g = blockIdx.x;
    for (int r = 0; r < rounds; r++) {
        for (int i = threadIdx.x; i < rows_per_round; i += 256) y[((size_t)g * rounds + r) * rows_per_round + i] = r + hog[threadIdx.x];
        if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (MODE >= 2) {
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_store(&flags[g], r + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const int nb = (g + 8) % gridDim.x; // a workgroup on the same XCD, so that progress never depends on dispatch order
                int spins = 0;
                while (__hip_atomic_load(&flags[nb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < r + 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
}

int main()
{
    const int wgs = 512, rounds = 64, rows = 2048; // 16 KB of y per workgroup and round, like a C2-sized run
    double* y; int* flags;
    CK(hipMalloc(&y, sizeof(double) * (size_t)wgs * rounds * rows));
    CK(hipMalloc(&flags, sizeof(int) * wgs));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* name) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(flags, 0, sizeof(int) * wgs));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, y, flags, rounds, rows);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("FENCE %-46s %8.1f us per launch = %6.2f us per round\n", name, best * 1e3, best * 1e3 / rounds);
    };
    run(k<0>, "stores only");
    run(k<1>, "stores + release fence (agent)");
    run(k<2>, "stores + release, flag, neighbour wait, acquire");
    return 0;
}
