// oracle/ref_glue_multiold.cpp — TEST INFRASTRUCTURE.
//
// extern "C" door into the REAL three-vector orthogonalize of
// mpk/old/SpMVmulti.cpp:164-169 (the copy of mpk/SpMVmulti.cpp:146-151 that
// still carries its struct definitions and therefore compiles):
//   beta = std::inner_product(b, x1);  x3[i] = x1[i] - alpha * beta * b[i].
// oracle/Makefile compiles that self-contained file where it lies with
// -Dmain=ref_multiold_main_unused and links it with this file into
// oracle/_ref/libref_multiold.so.  No reference code here.
#include <cstddef>
#include <vector>

void orthogonalize(int nrow, const std::vector<double>& b, const std::vector<double>& x1, std::vector<double>& x3,
                   double alpha);

extern "C" {

int ref_orthogonalize3(int n, const double* b, const double* x1, double* x3, double alpha)
{
    std::vector<double> bv(b, b + n), xv(x1, x1 + n), out((size_t)n, 0.0);
    orthogonalize(n, bv, xv, out, alpha);
    for (int i = 0; i < n; i++) x3[i] = out[i];
    return 0;
}

} // extern "C"
