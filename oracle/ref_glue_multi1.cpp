// oracle/ref_glue_multi1.cpp — TEST INFRASTRUCTURE.
//
// extern "C" door into the REAL SpM4V_AVX2 of mpk/SpMVmulti-1.cpp:434-493 (the
// k = 4 first-touch traversal whose innermost row sum is a 4-wide AVX2 fma with
// a scalar remainder).  That file is self-contained (own structs, own
// Generate*layer, own main); oracle/Makefile compiles it where it lies with
// -Dmain=ref_multi1_main_unused and links it with this file into
// oracle/_ref/libref_multi1.so.  No reference code here: the struct only
// re-declares the layout of mpk/SpMVmulti-1.cpp:15-20.
#include <cstddef>
#include <vector>

struct csrmatrix {
    int n, nnz;
    std::vector<int> ptrow;
    std::vector<int> indcol;
    std::vector<double> coef;
};

void Generate1stlayer(std::vector<int>& ptrowend1, csrmatrix& A);
void Generate2ndlayer(std::vector<std::vector<int>>& ptrowend2, csrmatrix& A, std::vector<int>& ptrowend1);
void Generate3rdlayer(std::vector<std::vector<std::vector<int>>>& ptrowend3, csrmatrix& A,
                      std::vector<int>& ptrowend1, std::vector<std::vector<int>>& ptrowend2);
void SpM4V_AVX2(double* y4, double* y3, double* y2, double* y1, const double* x, const csrmatrix& A,
                const std::vector<int>& ptrowend1, const std::vector<std::vector<int>>& ptrowend2,
                const std::vector<std::vector<std::vector<int>>>& ptrowend3);

extern "C" {

// Y = 4 contiguous vectors of n: Y[0] = A x .. Y[3] = A^4 x
int ref_multi1_spm4v_avx2(int n, int nnz, const int* ptrow, const int* indcol, const double* coef, const double* x,
                          double* Y)
{
    csrmatrix a;
    a.n = n;
    a.nnz = nnz;
    a.ptrow.assign(ptrow, ptrow + n + 1);
    a.indcol.assign(indcol, indcol + nnz);
    a.coef.assign(coef, coef + nnz);
    std::vector<int> e1(nnz);
    std::vector<std::vector<int>> e2(nnz);
    std::vector<std::vector<std::vector<int>>> e3(nnz);
    Generate1stlayer(e1, a);
    Generate2ndlayer(e2, a, e1);
    Generate3rdlayer(e3, a, e1, e2);
    SpM4V_AVX2(Y + 3 * (size_t)n, Y + 2 * (size_t)n, Y + (size_t)n, Y, x, a, e1, e2, e3);
    return 0;
}

} // extern "C"
