// spmv_rowpar.hpp — the baseline CSR kernel (a non-template kernel: include it in ONE translation unit).
#pragma once
#include "spmv_kernels.hpp"

namespace mi355 {

// ---------------------------------------------------------------------------
// rowpar kernel: one thread per row straight from global memory — the shape of
// the reference's CPU loop (mpk/SpMV.cpp:41-56).  Uncoalesced (lane stride =
// row length); kept as the simple always-valid baseline and for tiny matrices.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void spmv_csr_rowpar(CsrView A, const double* __restrict__ x,
                                                       double* __restrict__ y)
{
    const int r = blockIdx.x * kWG + threadIdx.x;
    if (r >= A.n) return;
    double s = 0.0;
    for (int k = A.ptrow[r]; k < A.ptrow[r + 1]; k++) s = fma(A.coef[k], x[A.indcol[k]], s);
    y[A.rowmap ? A.rowmap[r] : r] = s;
}

} // namespace mi355
