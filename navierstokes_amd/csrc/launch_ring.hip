// launch_ring.hip — dispatch of a ring launch to its configuration group (launch_ring_a.hip, launch_ring_d2/3/4.hip).
#include "capi_internal.hpp"

void launch_ring_cfg123(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot);
void launch_ring_cfg4_d2(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot);
void launch_ring_cfg4_d3(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot);
void launch_ring_cfg4_d4(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot);

void launch_ring_cfg(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot)
{
    if (A->ring.cfg.id >= 1 && A->ring.cfg.id <= 3) return launch_ring_cfg123(A, V, d_x, d_y, s, comm, dot);
    int depth = A->ring.cfg.depth; // chosen at create (long runs: 4 blocks of prefetch); MI355_RING_DEPTH is read per
                                   // launch so that tools/depth_ab.py can compare the depths on one handle
    if (const char* e = getenv("MI355_RING_DEPTH")) depth = atoi(e);
    if (depth == 3) launch_ring_cfg4_d3(A, V, d_x, d_y, s, comm, dot);
    else if (depth == 4) launch_ring_cfg4_d4(A, V, d_x, d_y, s, comm, dot);
    else launch_ring_cfg4_d2(A, V, d_x, d_y, s, comm, dot);
}
