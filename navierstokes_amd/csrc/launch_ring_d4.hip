// launch_ring_d4.hip — ring kernel instantiations: configuration 4 (256 threads, 2048-nonzero blocks), 4 blocks of prefetch.  Part of libmi355spmv.so (capi_internal.hpp).
#include "launch_ring_impl.hpp"

void launch_ring_cfg4_d4(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot) { launch_ring<256, 2048, 5120, 4>(A, V, d_x, d_y, s, comm, dot); }
