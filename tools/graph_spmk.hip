// tools/graph_spmk.hip — development tool: does replaying the k launches of mi_spmk_dev as ONE HIP graph
// remove anything from a matrix-powers step?  (no RCCL involved.)
//   ./tools/graph_spmk [n=1000000] [k=4] [steps=200]
// Prints the step time with direct launches and with graph replay, and checks that both produce the same bits.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mi355_spmv.h"

extern "C" {
long long synth_count(int kind, unsigned long long seed, int n, int w, long long rb, long long re);
int synth_rows(int kind, unsigned long long seed, int n, int w, long long rb, long long re, int* ptrow, int* indcol, double* coef);
void synth_x_sin(long long jb, long long je, double* x);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define MI(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s -> %s (%s)\n", #x, mi_strerror(r_), mi_last_error()); exit(3); } } while (0)

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1000000, k = argc > 2 ? atoi(argv[2]) : 4, steps = argc > 3 ? atoi(argv[3]) : 200;
    const long long nnz = synth_count(0, 0x5EED, n, 2000, 0, n);
    std::vector<int> p((size_t)n + 1), c((size_t)nnz);
    std::vector<double> v((size_t)nnz), x((size_t)n);
    synth_rows(0, 0x5EED, n, 2000, 0, n, p.data(), c.data(), v.data());
    synth_x_sin(0, n, x.data());
    mi_csr_t A;
    MI(mi_csr_create(n, n, p.data(), c.data(), v.data(), &A));
    double* dx;
    std::vector<double*> dy(k), dz(k);
    CK(hipMalloc(&dx, 8 * (size_t)n));
    CK(hipMemcpy(dx, x.data(), 8 * (size_t)n, hipMemcpyHostToDevice));
    for (int i = 0; i < k; i++) { CK(hipMalloc(&dy[i], 8 * (size_t)n)); CK(hipMalloc(&dz[i], 8 * (size_t)n)); }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 10; w++) MI(mi_spmk_dev(A, k, dx, dy.data(), s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < steps; i++) MI(mi_spmk_dev(A, k, dx, dy.data(), s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms_direct; CK(hipEventElapsedTime(&ms_direct, e0, e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    MI(mi_spmk_dev(A, k, dx, dz.data(), s));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 10; w++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < steps; i++) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms_graph; CK(hipEventElapsedTime(&ms_graph, e0, e1));
    std::vector<double> a((size_t)n), b((size_t)n);
    bool same = true;
    for (int i = 0; i < k; i++) {
        CK(hipMemcpy(a.data(), dy[i], 8 * (size_t)n, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), dz[i], 8 * (size_t)n, hipMemcpyDeviceToHost));
        same = same && !memcmp(a.data(), b.data(), 8 * (size_t)n);
    }
    printf("GRAPH n=%d k=%d kernel=%s: direct %.2f us per step (%.2f per product), graph replay %.2f us per step (%.2f per product), same bits %d\n",
           n, k, mi_csr_kernel_name(A), ms_direct * 1e3 / steps, ms_direct * 1e3 / steps / k, ms_graph * 1e3 / steps, ms_graph * 1e3 / steps / k, (int)same);
    return same ? 0 : 1;
}
