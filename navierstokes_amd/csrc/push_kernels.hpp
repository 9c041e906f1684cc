// push_kernels.hpp — the two stand-alone kernels of the four-launch push step (protocol: push_exchange.hpp).
// Non-template kernels: include this in ONE translation unit (capi_part.hip).
#pragma once
#include "push_exchange.hpp"

namespace mi355 {

// The stand-alone push (four-launch form of the step).  A link's payload is cut into chunks of kPushChunk doubles, one
// workgroup each (work[w] = {link, chunk}), so that a large halo — an FE slab's boundary plane is 150 KB per neighbour — goes
// out from many CUs at once instead of through one workgroup's store queue.  Small payloads use the write-through 8-byte
// stores above; from kPushBigLink doubles on they use plain 16-byte stores and ONE system-scope release fence per workgroup
// (an L2 write-back: microseconds, but amortised over 32 KB, where the write-through form — one fabric write per 8 bytes —
// costs more).  The flag goes up when the link's last chunk is out: a ticket counter per link, bumped by every chunk's
// workgroup after its drain / fence; the last arriver resets it and stores the flag.
constexpr int kPushBigLink = 8192;

// one {link, chunk} item of the push, by a workgroup of 256 threads
__device__ __forceinline__ void push_chunk(const PushLink& L, int link, int chunk, const int* __restrict__ link_chunks, unsigned* __restrict__ tickets,
                                           const int* __restrict__ send_idx, const double* __restrict__ x, unsigned step)
{
    double* dst = (step & 1u) ? L.dst[1] : L.dst[0];
    const int i0 = chunk * kPushChunk, i1 = min(L.count, i0 + kPushChunk);
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
        const unsigned done = __hip_atomic_fetch_add(&tickets[link], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)done == link_chunks[link] - 1) { // the link's last chunk: everybody else's payload is out (their release, this acquire)
            __hip_atomic_store(&tickets[link], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(L.flag, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void halo_push_kernel(const PushLink* __restrict__ links, const int2* __restrict__ work,
                                                        const int* __restrict__ link_chunks, unsigned* __restrict__ tickets,
                                                        const int* __restrict__ send_idx, const double* __restrict__ x, unsigned step)
{
    const int2 w = work[blockIdx.x];
    const PushLink L = links[w.x];
    push_chunk(L, w.x, w.y, link_chunks, tickets, send_idx, x, step);
}

__global__ __launch_bounds__(256) void halo_wait_copy_kernel(const unsigned* flags, const int* __restrict__ nb, int n_nb, unsigned step,
                                                             const double* src, double* __restrict__ dst, int n_halo,
                                                             unsigned* timeouts /* host-visible */, unsigned spin_max)
{
    push_wait_flags(flags, nb, n_nb, step, 0u, timeouts, spin_max, threadIdx.x, 256);
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
