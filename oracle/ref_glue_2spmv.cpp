// oracle/ref_glue_2spmv.cpp — TEST INFRASTRUCTURE.
//
// extern "C" doors into the REAL "interleave" helpers of mpk/2SpMV.cpp:
// orthogonalize (in-place dot + AXPY, :3-11) and orthonormalize_against_basis
// (classical Gram-Schmidt against a list of vectors + norm, :13-28).  That
// translation unit carries its own main(); oracle/Makefile compiles it where it
// lies with -Dmain=ref_2spmv_main_unused and links it with this file,
// mpk/SpMV.cpp and mpk/utils.cpp into oracle/_ref/libref_2spmv.so.  No reference
// code here.
#include "SpMV.h" // the reference's mpk/SpMV.h

// defined in mpk/2SpMV.cpp (not declared in its header; the default argument lives on the definition)
void orthogonalize(int nrow, const std::vector<double>& x, std::vector<double>& y, double alpha);
void orthonormalize_against_basis(int nrow, std::vector<std::vector<double>>& basis, std::vector<double>& y);

extern "C" {

// y -= alpha * (x . y) * x, in place (mpk/2SpMV.cpp:3-11)
int ref_orthogonalize_inplace(int n, const double* x, double* y, double alpha)
{
    std::vector<double> xv(x, x + n), yv(y, y + n);
    orthogonalize(n, xv, yv, alpha);
    for (int i = 0; i < n; i++) y[i] = yv[i];
    return 0;
}

// y -= (y . v_m) v_m for each of the m basis vectors in turn, each dot taken on the UPDATED y
// (mpk/2SpMV.cpp:13-28); basis = m contiguous vectors of n
int ref_orthonormalize_against_basis(int n, int m, const double* basis, double* y)
{
    std::vector<std::vector<double>> B((size_t)m);
    for (int j = 0; j < m; j++) B[j].assign(basis + (size_t)j * n, basis + (size_t)(j + 1) * n);
    std::vector<double> yv(y, y + n);
    orthonormalize_against_basis(n, B, yv);
    for (int i = 0; i < n; i++) y[i] = yv[i];
    return 0;
}

} // extern "C"
