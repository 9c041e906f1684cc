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
#pragma once
#include <hip/hip_runtime.h>

namespace mi355 {

constexpr int kWinFlagStride = 16; // unsigneds: one 64-byte line per sender

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

// (the stand-alone push and wait + copy kernels of the four-launch form live in push_kernels.hpp)

} // namespace mi355
