// tools/comm_timing.hip — development tool (not product): what one step of mi_part_spmv_dev costs AROUND its kernels.
//
//   ./tools/comm_timing [steps=1000] [count=2000]
//
// A size-1 RCCL communicator sends `count` doubles to itself through the same grouped send/recv call
// sequence the library issues, with the cross-stream hand-offs done (a) by HIP events, (b) by the
// flag kernels of handoff_kernels.hpp, (c) as one ncclAllToAllv.  Prints host time to enqueue and
// device time per step.  (This code lived inside mi_comm_selftest in round 1; profiles/NOTES.md §6 quotes its
// numbers.)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../navierstokes_amd/csrc/handoff_kernels.hpp"
#include "../navierstokes_amd/csrc/rccl_loader.hpp"

using namespace mi355;

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));                 \
            return 1;                                                                  \
        }                                                                              \
    } while (0)
#define NCCL_TRY(expr)                                                                 \
    do {                                                                               \
        int r_ = (expr);                                                               \
        if (r_ != 0) {                                                                 \
            fprintf(stderr, "%s: %s\n", #expr, g_rccl.GetErrorString(r_));             \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static Rccl& g_rccl = rccl_state();

struct Plan1 { // a 1-rank "partition" that sends `count` entries to itself
    int count;
};

static int enqueue_exchange(const Plan1& pl, void* comm, const double* d_src, double* d_dst, hipStream_t cs)
{
    NCCL_TRY(g_rccl.GroupStart());
    NCCL_TRY(g_rccl.Send(d_src, (size_t)pl.count, kNcclDouble, 0, comm, cs));
    NCCL_TRY(g_rccl.Recv(d_dst, (size_t)pl.count, kNcclDouble, 0, comm, cs));
    NCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

int main(int argc, char** argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 1000;
    const int count = argc > 2 ? atoi(argv[2]) : 2000;
    if (!rccl_load()) {
        fprintf(stderr, "RCCL unavailable: %s\n", g_rccl.why.c_str());
        return 2;
    }
    IdByValue id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    void* comm = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&comm, 1, id, 0));
    Plan1 pl{count};
    int rc = 0;
    double *d_src = nullptr, *d_dst = nullptr;
    hipStream_t s0 = nullptr, cs = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipMalloc(&d_src, sizeof(double) * count));
    HIP_TRY(hipMalloc(&d_dst, sizeof(double) * count));
    HIP_TRY(hipMemset(d_src, 0, sizeof(double) * count));
    HIP_TRY(hipStreamCreate(&s0));
    HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    {
        hipEvent_t t0 = nullptr, t1 = nullptr;
        HIP_TRY(hipEventCreate(&t0));
        HIP_TRY(hipEventCreate(&t1));
        for (int rep = 0; rep < 2; rep++) {
            HIP_TRY(hipStreamSynchronize(s0));
            const auto w0 = std::chrono::steady_clock::now();
            HIP_TRY(hipEventRecord(t0, s0));
            for (int i = 0; i < steps; i++) { // the call sequence of mi_part_spmv_dev without its three kernels
                HIP_TRY(hipEventRecord(e0, s0));
                HIP_TRY(hipStreamWaitEvent(cs, e0, 0));
                if ((rc = enqueue_exchange(pl, comm, d_src, d_dst, cs))) return rc;
                HIP_TRY(hipEventRecord(e1, cs));
                HIP_TRY(hipStreamWaitEvent(s0, e1, 0));
            }
            HIP_TRY(hipEventRecord(t1, s0));
            const auto w1 = std::chrono::steady_clock::now();
            HIP_TRY(hipStreamSynchronize(s0));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
            fprintf(stderr, "comm_timing: %d steps, host %.1f us/step to enqueue, device %.1f us/step (self send/recv of %d doubles)\n",
                    steps, std::chrono::duration<double, std::micro>(w1 - w0).count() / steps, ms * 1e3 / steps, count);
        }
        for (int variant = 0; variant < 2; variant++) { // the hand-offs alone / the exchange alone on one stream
            HIP_TRY(hipStreamSynchronize(s0));
            HIP_TRY(hipStreamSynchronize(cs));
            const auto w0 = std::chrono::steady_clock::now();
            HIP_TRY(hipEventRecord(t0, variant == 0 ? s0 : cs));
            for (int i = 0; i < steps; i++) {
                if (variant == 0) {
                    HIP_TRY(hipEventRecord(e0, s0));
                    HIP_TRY(hipStreamWaitEvent(cs, e0, 0));
                    HIP_TRY(hipEventRecord(e1, cs));
                    HIP_TRY(hipStreamWaitEvent(s0, e1, 0));
                } else if ((rc = enqueue_exchange(pl, comm, d_src, d_dst, cs))) return rc;
            }
            HIP_TRY(hipEventRecord(t1, variant == 0 ? s0 : cs));
            const auto w1 = std::chrono::steady_clock::now();
            HIP_TRY(hipStreamSynchronize(s0));
            HIP_TRY(hipStreamSynchronize(cs));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
            fprintf(stderr, "comm_timing (%s): host %.1f us/step, device %.1f us/step\n",
                    variant == 0 ? "two event hand-offs only" : "grouped send/recv only, one stream",
                    std::chrono::duration<double, std::micro>(w1 - w0).count() / steps, ms * 1e3 / steps);
        }
        { // hand-offs by flag kernels: s0 sets flag A, cs waits for it, exchange, cs sets flag B, s0 waits for it
            unsigned* fl = nullptr;
            HIP_TRY(hipMalloc(&fl, 4 * sizeof(unsigned)));
            HIP_TRY(hipMemset(fl, 0, 4 * sizeof(unsigned)));
            for (int variant = 0; variant < 2; variant++) { // 0: hand-offs only, 1: with the exchange
                HIP_TRY(hipStreamSynchronize(s0));
                HIP_TRY(hipStreamSynchronize(cs));
                HIP_TRY(hipMemset(fl, 0, 4 * sizeof(unsigned)));
                const auto w0 = std::chrono::steady_clock::now();
                HIP_TRY(hipEventRecord(t0, s0));
                for (int i = 0; i < steps; i++) {
                    hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(64), 0, s0, fl, (unsigned)(i + 1));
                    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, cs, fl, (unsigned)(i + 1), fl + 2);
                    if (variant == 1 && (rc = enqueue_exchange(pl, comm, d_src, d_dst, cs))) return rc;
                    hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(64), 0, cs, fl + 1, (unsigned)(i + 1));
                    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, s0, fl + 1, (unsigned)(i + 1), fl + 2);
                }
                HIP_TRY(hipEventRecord(t1, s0));
                const auto w1 = std::chrono::steady_clock::now();
                HIP_TRY(hipStreamSynchronize(s0));
                HIP_TRY(hipStreamSynchronize(cs));
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
                unsigned to = 0;
                HIP_TRY(hipMemcpy(&to, fl + 2, sizeof to, hipMemcpyDeviceToHost));
                fprintf(stderr, "comm_timing (flag-kernel hand-offs%s): host %.1f us/step, device %.1f us/step, %u spin timeouts\n",
                        variant ? " + grouped send/recv" : " only", std::chrono::duration<double, std::micro>(w1 - w0).count() / steps,
                        ms * 1e3 / steps, to);
            }
            (void)hipFree(fl);
        }
        if (g_rccl.AllToAllv) { // the same exchange as ONE ncclAllToAllv call
            const size_t sc[1] = {(size_t)count}, sd[1] = {0};
            for (int rep = 0; rep < 2; rep++) {
                HIP_TRY(hipStreamSynchronize(s0));
                const auto w0 = std::chrono::steady_clock::now();
                HIP_TRY(hipEventRecord(t0, s0));
                for (int i = 0; i < steps; i++) {
                    HIP_TRY(hipEventRecord(e0, s0));
                    HIP_TRY(hipStreamWaitEvent(cs, e0, 0));
                    NCCL_TRY(g_rccl.AllToAllv(d_src, sc, sd, d_dst, sc, sd, kNcclDouble, comm, cs));
                    HIP_TRY(hipEventRecord(e1, cs));
                    HIP_TRY(hipStreamWaitEvent(s0, e1, 0));
                }
                HIP_TRY(hipEventRecord(t1, s0));
                const auto w1 = std::chrono::steady_clock::now();
                HIP_TRY(hipStreamSynchronize(s0));
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
                fprintf(stderr, "comm_timing (ncclAllToAllv): host %.1f us/step, device %.1f us/step\n",
                        std::chrono::duration<double, std::micro>(w1 - w0).count() / steps, ms * 1e3 / steps);
            }
        }
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
    }
    g_rccl.CommDestroy(comm);
    return rc;
}
