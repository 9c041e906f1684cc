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

// The stand-alone push (four-launch form of the step).  A link's payload is cut into chunks of kPushChunk doubles, one
// workgroup each (work[w] = {link, chunk}), so that a large halo — an FE slab's boundary plane is 150 KB per neighbour — goes
// out from many CUs at once instead of through one workgroup's store queue.  Small payloads use the write-through 8-byte
// stores above; from kPushBigLink doubles on they use plain 16-byte stores and ONE system-scope release fence per workgroup
// (an L2 write-back: microseconds, but amortised over 32 KB, where the write-through form — one fabric write per 8 bytes —
// costs more).  The flag goes up when the link's last chunk is out: a ticket counter per link, bumped by every chunk's
// workgroup after its drain / fence; the last arriver resets it and stores the flag.
constexpr int kPushChunk = 1024; // 8 KB per workgroup: an FE slab's 150 KB plane goes out from ~19 CUs at once
constexpr int kPushBigLink = 8192;

__global__ __launch_bounds__(256) void halo_push_kernel(const PushLink* __restrict__ links, const int2* __restrict__ work,
                                                        const int* __restrict__ link_chunks, unsigned* __restrict__ tickets,
                                                        const int* __restrict__ send_idx, const double* __restrict__ x, unsigned step)
{
    const int2 w = work[blockIdx.x];
    const PushLink L = links[w.x];
    double* dst = L.dst[step & 1u];
    const int i0 = w.y * kPushChunk, i1 = min(L.count, i0 + kPushChunk);
    if (L.count >= kPushBigLink && L.first >= 0 && ((L.first | i0) & 1) == 0 && (((uintptr_t)dst | (uintptr_t)x) & 15) == 0) {
        const double2* s2 = reinterpret_cast<const double2*>(x + L.first + i0);
        double2* d2 = reinterpret_cast<double2*>(dst + i0);
        const int n2 = (i1 - i0) >> 1;
        for (int i = threadIdx.x; i < n2; i += 256) d2[i] = s2[i];
        if (((i1 - i0) & 1) && threadIdx.x == 0) dst[i1 - 1] = x[L.first + i1 - 1];
        __threadfence_system();
    } else {
        if (L.first >= 0) {
            for (int i = i0 + threadIdx.x; i < i1; i += 256) push_store(dst + i, x[L.first + i]);
        } else {
            for (int i = i0 + threadIdx.x; i < i1; i += 256) push_store(dst + i, x[send_idx[L.send_off + i]]);
        }
        push_drain(); // every storing wave: its stores have left for the peer
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(&tickets[w.x], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)done == link_chunks[w.x] - 1) { // the link's last chunk: everybody else's payload is out (their release, this acquire)
            __hip_atomic_store(&tickets[w.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(L.flag, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void halo_wait_copy_kernel(const unsigned* flags, const int* __restrict__ nb, int n_nb, unsigned step,
                                                             const double* src, double* __restrict__ dst, int n_halo,
                                                             unsigned* timeouts /* host-visible */, unsigned spin_max)
{
    for (int j = threadIdx.x; j < n_nb; j += 256) {
        const unsigned* f = flags + (size_t)nb[j] * kWinFlagStride;
        unsigned spins = 0;
        // flags are monotone step numbers; "behind" is computed modulo 2^32 so that a wrap after 4e9 steps is harmless
        while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - step) < 0) {
            if (spins < 4096) __builtin_amdgcn_s_sleep(2);
            else __builtin_amdgcn_s_sleep(127);
            if (++spins > spin_max) { // default 2^23: ~30 s
                __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // system scope: the data the flags announce
    const long long stride = (long long)gridDim.x * 256;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) { // 16 bytes per access: the window is uncached, every load is a trip to memory
        const long long n2 = n_halo >> 1;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride)
            reinterpret_cast<double2*>(dst)[i] = reinterpret_cast<const double2*>(src)[i];
        if ((n_halo & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n_halo - 1] = src[n_halo - 1];
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_halo; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
    }
}

} // namespace mi355
