// launch_ring_impl.hpp — how one ring configuration <T, NNZB, RING, D> fans out into the kernel's instantiations (row map,
// non-temporal values, skewed staging, fused multi-GPU step, LEAN, dot epilogue).  Included by launch_ring_*.hip, one configuration
// group each: the 96 instantiations are most of the library's device-code compile time and build in parallel this way.
#pragma once
#include "capi_internal.hpp"

template <int T, int NNZB, int RING, int D, bool MAPPED, bool NT, bool SKEW, bool LEAN>
static void launch_ring3(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot)
{
    if constexpr (LEAN && !MAPPED) {
        if (dot && !comm) { // the dot epilogue (spmv_ring.hpp: RingDot); the caller checked ring_dot_eligible()
            hipLaunchKernelGGL((spmv_csr_ring<T, NNZB, RING, D, kRingMaxB, false, NT, SKEW, false, true, true>), dim3(A->ring.wgs), dim3(T), 0, s, V,
                               reinterpret_cast<const int4*>(A->ring.d_plan), A->ring.d_ok, A->ring.d_slots, d_x, d_y, reinterpret_cast<const int2*>(A->ring.d_rng), A->ring.uniform ? A->ring.bpw : 0, RingComm{}, *dot);
            return;
        }
    }
    if (!MAPPED && comm) { // the fused multi-GPU step: push workgroups in front of the grid (spmv_ring.hpp)
        hipLaunchKernelGGL((spmv_csr_ring<T, NNZB, RING, D, kRingMaxB, MAPPED, NT, SKEW, true, LEAN>), dim3(A->ring.wgs + comm->push_wgs), dim3(T), 0, s,
                           V, reinterpret_cast<const int4*>(A->ring.d_plan), A->ring.d_ok, A->ring.d_slots, d_x, d_y, reinterpret_cast<const int2*>(A->ring.d_rng), A->ring.uniform ? A->ring.bpw : 0, *comm);
        return;
    }
    hipLaunchKernelGGL((spmv_csr_ring<T, NNZB, RING, D, kRingMaxB, MAPPED, NT, SKEW, false, LEAN>), dim3(A->ring.wgs), dim3(T), 0, s, V,
                       reinterpret_cast<const int4*>(A->ring.d_plan), A->ring.d_ok, A->ring.d_slots, d_x, d_y, reinterpret_cast<const int2*>(A->ring.d_rng), A->ring.uniform ? A->ring.bpw : 0, RingComm{});
}

template <int T, int NNZB, int RING, int D, bool MAPPED, bool NT, bool SKEW>
static void launch_ring2(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot)
{
    // the LEAN instantiation exists for the configuration that runs in practice (4: 256 threads) at depths 2 and 4
    if (T == 256 && D != 3 && A->ring.lean) launch_ring3<T, NNZB, RING, D, MAPPED, NT, SKEW, (T == 256 && D != 3)>(A, V, d_x, d_y, s, comm, dot);
    else launch_ring3<T, NNZB, RING, D, MAPPED, NT, SKEW, false>(A, V, d_x, d_y, s, comm, dot);
}

template <int T, int NNZB, int RING, int D, bool MAPPED>
static void launch_ring1(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot)
{
    if (A->ring.nt) {
        if (A->ring.skew) launch_ring2<T, NNZB, RING, D, MAPPED, true, true>(A, V, d_x, d_y, s, comm, dot);
        else launch_ring2<T, NNZB, RING, D, MAPPED, true, false>(A, V, d_x, d_y, s, comm, dot);
    } else {
        if (A->ring.skew) launch_ring2<T, NNZB, RING, D, MAPPED, false, true>(A, V, d_x, d_y, s, comm, dot);
        else launch_ring2<T, NNZB, RING, D, MAPPED, false, false>(A, V, d_x, d_y, s, comm, dot);
    }
}

template <int T, int NNZB, int RING, int D>
static void launch_ring(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot)
{
    static_assert(NNZB <= kRingPadNnz && 2 * T + 1 <= kRingPadRows, "device arrays are padded for the kernel's unclamped loads");
    if (V.rowmap) launch_ring1<T, NNZB, RING, D, true>(A, V, d_x, d_y, s, comm, dot);
    else launch_ring1<T, NNZB, RING, D, false>(A, V, d_x, d_y, s, comm, dot);
}
