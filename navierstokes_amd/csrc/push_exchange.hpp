// push_exchange.hpp — halo exchange by PEER PUSH: each rank's kernel writes the x entries its neighbours
// need straight into a receive window in THEIR memory (mapped through HIP IPC; over xGMI between GPUs)
// and raises a flag there; the receiver's kernel waits for its neighbours' flags and moves the window
// into the halo part of x.  No RCCL call, no second stream, no event or flag-kernel hand-off: the step
// is four launches on the caller's stream (push, interior rows, wait+copy, boundary rows).
//
// New design (the reference has no distributed code).  Protocol, per partition handle:
//   * window = nranks flag slots (64 B apart, slot p written only by rank p) + 2 x n_halo doubles
//     (two parities), allocated uncached so that neither side's L2 can hold a stale line;
//   * step t (both sides count calls): the sender stores its entries into parity t & 1 of the peer's
//     window with WRITE-THROUGH stores (system-scope relaxed atomic stores = `global_store … sc0 sc1`: they never
//     sit dirty in the sender's L2, so no L2 write-back is needed — a release fence there costs microseconds once
//     the product's own y stores have dirtied the L2), every storing wave drains them (`s_waitcnt vmcnt(0)`), the
//     workgroup meets at a barrier, ONE lane then stores t into its flag slot (system-scope atomic store) — the
//     "write-through payload, drained, then flag" form of MI355X_MICROARCH.md § visibility;
//   * ONE PROCESS PER RANK: the receiver's wait kernel spins until another rank's push kernel has run; ranks that are
//     threads of one process share that process's few hardware queues, where a waiting kernel can be queued in front
//     of the very kernel it waits for (observed: 4 rank threads hang).  Separate processes have separate queues.
//   * the receiver spins (bounded; a give-up is counted in host-visible memory and is sticky) until
//     every NEIGHBOUR's slot shows >= t, fences (system acquire), and copies parity t & 1 into x;
//   * neighbours are made symmetric (a rank also flags peers it only receives from): a sender can then
//     only reach step t + 2 — and overwrite parity t & 1 — after it saw the receiver's flag of step
//     t + 1, which the receiver raises after its copy of step t in stream order.  Two parities suffice.
//     THE INVARIANT every form of the step keeps (tests/test_push_protocol.py is its executable model: random
//     interleavings of {push, flag, wait, read} per rank, the one-launch and four-launch forms mixed, asymmetric
//     couplings, ranks that send but receive nothing): a rank's kernels of step t + 1 start only after SOME kernel of
//     its step t has seen every neighbour's flag >= t.  In the four-launch form that is the wait + copy kernel (it runs
//     whenever the rank has a neighbour, halo or not); in the one-launch forms it is a ghost-reading run / workgroup —
//     and a rank WITHOUT ghosts (it only sends: an upwind coupling, the last rank of a triangular band) has none, so its
//     pushing workgroups themselves wait for flag >= t - 1 before they store step t (push_wait_flags with lag 1).
//     Without that a sender could run any number of steps ahead and overwrite a parity its peer was still reading.
//   * the two forms may be mixed freely across ranks (both raise and await the same flags; the model checks it).
//     DistCSR still makes the form collective — a matter of balanced step times, not of correctness.
#pragma once
#include <hip/hip_runtime.h>

namespace mi355 {

constexpr int kWinFlagStride = 16; // unsigneds: one 64-byte line per sender
constexpr int kPushChunk = 1024;    // doubles per {link, chunk} item of a chunked push (8 KB per workgroup: an FE slab's 150 KB plane goes out from ~19 CUs at once)

// payload store that leaves for the peer at once (sc0 sc1), and the drain behind a batch of them
__device__ __forceinline__ void push_store(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void push_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct PushLink {
    double* dst[2];  // where my entries go in the peer's window, per parity
    unsigned* flag;  // my slot in the peer's window
    int send_off;    // offset of this peer's list in send_idx
    int count;       // entries to push (0: flag only)
    int first;       // >= 0: the list is the contiguous slice x[first .. first+count)
};

// Polls before a wait gives up: 2^kPushSpinLog2Default (the back-off sleeps ~4 us per poll, so ~4 s; MI355_PUSH_SPIN_LOG2
// overrides, 8..30).  THE one place this default is written down.
constexpr int kPushSpinLog2Default = 20;

// Wait (threads tid, tid + nthreads, ... take one neighbour each) until every neighbour's flag in MY window shows
// >= step - lag.  Flags are monotone step numbers; "behind" is computed modulo 2^32, so a wrap after 4e9 steps is
// harmless.  Bounded: a give-up is counted in `timeouts` (host-visible, sticky, fails every later call on the handle) —
// and a wait that is not satisfied at once looks at that counter after a few dozen polls and gives up with it, so that the
// steps already queued behind a stalled peer cost microseconds each, not spin_max polls each (a healthy wait never
// reads host memory).  The caller follows with a barrier and a system-scope acquire fence before it reads the window.
__device__ __forceinline__ void push_wait_flags(const unsigned* flags, const int* __restrict__ nb, int n_nb, unsigned step, unsigned lag,
                                                unsigned* timeouts, unsigned spin_max, int tid, int nthreads)
{
    for (int j = tid; j < n_nb; j += nthreads) {
        const unsigned* f = flags + (size_t)nb[j] * kWinFlagStride;
        const unsigned want = step - lag;
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
            if (spins < 4096) __builtin_amdgcn_s_sleep(2);
            else __builtin_amdgcn_s_sleep(127);
            ++spins;
            if (spins == 64 && __hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break; // an earlier wait gave up already
            if (spins > spin_max) {
                __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
}

// (the stand-alone push and wait + copy kernels of the four-launch form live in push_kernels.hpp)

} // namespace mi355
