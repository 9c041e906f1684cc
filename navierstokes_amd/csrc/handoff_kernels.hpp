// handoff_kernels.hpp — cross-stream hand-off by a flag in device memory instead of a HIP event
// (optional path of mi_part_spmv_dev, MI355_PART_HANDOFF=flags; timed by tools/comm_timing.hip).
//
// An event record + wait between two HIP streams costs ~10 us of latency on this runtime; a one-wave
// kernel that sets a counter on one stream and a one-wave kernel that spins on it on the other cost a
// launch each.  Data visibility does not rest on the flag: the producer's kernels completed before
// flag_set_kernel started (stream order, end-of-kernel release) and the consumer's kernels start after
// flag_wait_kernel finished (stream order, start-of-kernel acquire); the flag only carries "has
// happened".  The spin gives up after minutes rather than hang the GPU for good, and a wait that gave
// up is LOUD: it bumps `timeouts`, which lives in host-visible (pinned, mapped) memory, so the library
// sees it with a plain load at its next entry point — no copy, no synchronisation — and fails that and
// every later call on the handle (mi_part_status, mi_part_spmv_dev, mi_part_destroy).
#pragma once
#include <hip/hip_runtime.h>

namespace mi355 {

__global__ void flag_set_kernel(unsigned* flag, unsigned value)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void flag_wait_kernel(const unsigned* flag, unsigned value, unsigned* timeouts /* host-visible */)
{
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
            // poll fast at first (the usual wait is a few microseconds), then back off; give up only after
            // minutes (a peer may be busy setting up its RCCL channels on the first steps) — never hang for good
            if (spins < 4096) __builtin_amdgcn_s_sleep(2);
            else __builtin_amdgcn_s_sleep(127);
            if (++spins > (1u << 24)) { // ~1 min
                __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
}

} // namespace mi355
