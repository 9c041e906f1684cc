// oracle/ref_glue_spm2v.cpp — TEST INFRASTRUCTURE.
//
// extern "C" doors into the REAL fused 2-step kernels of mpk/SpM2V.cpp.  That
// translation unit carries its own main(); oracle/Makefile compiles it where it
// lies with -Dmain=ref_spm2v_main_unused and links it with this file and
// mpk/utils.cpp into oracle/_ref/libref_spm2v.so.  No reference code here.
#include "SpMV.h" // the reference's mpk/SpMV.h

// defined in mpk/SpM2V.cpp (not declared in its header)
void Generate1stlayer(std::vector<int>& ptrowend1, csrmatrix& A);
void Generate1stlayer_BCSR4(std::vector<int>& ptrowendB, const bcsr4x4_matrix& A);
void SpM2V_CSR(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& ptrowend1);
void SpM2V_CSR_OPT(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& ptrowend1);
void SpM2V_CSR_AVX2(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& ptrowend1);
void SpM2V_BCSR_OPT(double* z, double* y, double* x, bcsr4x4_matrix& A, std::vector<int>& ptrowendB);
void SpM2V_BCSR_FMA(double* z, double* y, double* x, bcsr4x4_matrix& A, std::vector<int>& ptrowendB);
void SpM2V_BCSR_AVX2(double* z, double* y, double* x, bcsr4x4_matrix& A, std::vector<int>& ptrowendB);

extern "C" {

// first-touch table; end1 has nnz entries
int ref_gen_layer1(int n, int nnz, const int* ptrow, const int* indcol, int* end1)
{
    csrmatrix a;
    a.n = n;
    a.nnz = nnz;
    a.ptrow.assign(ptrow, ptrow + n + 1);
    a.indcol.assign(indcol, indcol + nnz);
    a.coef.assign((size_t)nnz, 0.0);
    std::vector<int> t;
    Generate1stlayer(t, a);
    for (int k = 0; k < nnz; k++) end1[k] = t[k];
    return 0;
}

// variant: 0 SpM2V_CSR (x87), 1 SpM2V_CSR_OPT, 3 SpM2V_CSR_AVX2 (drops <4 remainders)
int ref_spm2v_csr(int variant, int n, int nnz, const int* ptrow, const int* indcol, const double* coef,
                  const double* x, double* y, double* z)
{
    csrmatrix a;
    a.n = n;
    a.nnz = nnz;
    a.ptrow.assign(ptrow, ptrow + n + 1);
    a.indcol.assign(indcol, indcol + nnz);
    a.coef.assign(coef, coef + nnz);
    std::vector<int> t;
    Generate1stlayer(t, a);
    double* xx = const_cast<double*>(x);
    switch (variant) {
    case 0: SpM2V_CSR(z, y, xx, a, t); break;
    case 1: SpM2V_CSR_OPT(z, y, xx, a, t); break;
    case 3: SpM2V_CSR_AVX2(z, y, xx, a, t); break;
    default: return -1;
    }
    return 0;
}

// variant: 1 SpM2V_BCSR_OPT, 2 _FMA, 3 _AVX2 (variant 0, the no-sse scalar one,
// miscompiles at -O3 with g++ 11.4 — SURVEY.md §8c "Hazard" — and is not exposed)
int ref_spm2v_bcsr(int variant, int nbrows, int nblocks, const int* ptrow, const int* indcol,
                   const double* coef, const double* x, double* y, double* z)
{
    bcsr4x4_matrix a;
    a.nrows = nbrows;
    a.nblocks = nblocks;
    a.ptrow.assign(ptrow, ptrow + nbrows + 1);
    a.indcol.assign(indcol, indcol + nblocks);
    a.coef.assign(coef, coef + 16 * (size_t)nblocks);
    std::vector<int> t;
    Generate1stlayer_BCSR4(t, a);
    double* xx = const_cast<double*>(x);
    switch (variant) {
    case 1: SpM2V_BCSR_OPT(z, y, xx, a, t); break;
    case 2: SpM2V_BCSR_FMA(z, y, xx, a, t); break;
    case 3: SpM2V_BCSR_AVX2(z, y, xx, a, t); break;
    default: return -1;
    }
    return 0;
}

// first-touch table of block rows (mpk/SpM2V.cpp:28-46); endB has nblocks entries
int ref_gen_layer1_bcsr4(int nbrows, int nblocks, const int* ptrow, const int* indcol, int* endB)
{
    bcsr4x4_matrix a;
    a.nrows = nbrows;
    a.nblocks = nblocks;
    a.ptrow.assign(ptrow, ptrow + nbrows + 1);
    a.indcol.assign(indcol, indcol + nblocks);
    std::vector<int> t;
    Generate1stlayer_BCSR4(t, a);
    if ((int)t.size() < nblocks) return -1;
    for (int k = 0; k < nblocks; k++) endB[k] = t[k];
    return 0;
}

} // extern "C"
