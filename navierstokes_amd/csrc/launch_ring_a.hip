// launch_ring_a.hip — ring kernel instantiations: configurations 1-3 (512 threads).  Part of libmi355spmv.so (capi_internal.hpp).
#include "launch_ring_impl.hpp"

void launch_ring_cfg123(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot)
{
    switch (A->ring.cfg.id) {
    case 1: launch_ring<512, 2048, 5120, 2>(A, V, d_x, d_y, s, comm, dot); break;
    case 2: launch_ring<512, 4096, 5120, 2>(A, V, d_x, d_y, s, comm, dot); break;
    default: launch_ring<512, 4096, 11264, 2>(A, V, d_x, d_y, s, comm, dot); break;
    }
}
