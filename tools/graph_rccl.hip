// tools/graph_rccl.hip — development probe: can a grouped ncclSend/ncclRecv be captured into a HIP graph on this image?
// (Round 1 saw a host segfault inside capture and recorded it without a cause; run this under rocgdb for the frame:
//    rocgdb -batch -ex run -ex bt --args ./tools/graph_rccl [relaxed|global|threadlocal])
// A size-1 communicator sends 2000 doubles to itself inside hipStreamBeginCapture / EndCapture.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../navierstokes_amd/csrc/rccl_loader.hpp"
using namespace mi355;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NC(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s -> %s\n", #x, R.GetErrorString(r_)); return 3; } } while (0)

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    Rccl& R = rccl_state();
    if (!rccl_load()) { fprintf(stderr, "no RCCL: %s\n", R.why.c_str()); return 1; }
    const char* mode_s = argc > 1 ? argv[1] : "relaxed";
    const hipStreamCaptureMode mode = !strcmp(mode_s, "global") ? hipStreamCaptureModeGlobal
                                      : !strcmp(mode_s, "threadlocal") ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
    IdByValue id;
    NC(R.GetUniqueId(&id));
    void* comm = nullptr;
    NC(R.CommInitRank(&comm, 1, id, 0));
    const int n = 2000;
    double *a, *b;
    CK(hipMalloc(&a, 8 * n)); CK(hipMalloc(&b, 8 * n));
    CK(hipMemset(a, 0, 8 * n));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // eager once, so that RCCL's lazy channel set-up (allocations, proxy threads) is not inside the capture
    NC(R.GroupStart()); NC(R.Send(a, n, kNcclDouble, 0, comm, s)); NC(R.Recv(b, n, kNcclDouble, 0, comm, s)); NC(R.GroupEnd());
    CK(hipStreamSynchronize(s));
    printf("eager exchange ok; capturing (mode %s)...\n", mode_s);
    hipGraph_t g = nullptr;
    CK(hipStreamBeginCapture(s, mode));
    NC(R.GroupStart()); NC(R.Send(a, n, kNcclDouble, 0, comm, s)); NC(R.Recv(b, n, kNcclDouble, 0, comm, s)); NC(R.GroupEnd());
    printf("enqueued inside capture; ending capture...\n");
    CK(hipStreamEndCapture(s, &g));
    hipGraphExec_t ge = nullptr;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    printf("GRAPH_RCCL capture + 10 replays OK\n");
    R.CommDestroy(comm);
    return 0;
}
