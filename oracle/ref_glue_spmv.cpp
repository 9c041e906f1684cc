// oracle/ref_glue_spmv.cpp — TEST INFRASTRUCTURE.
//
// extern "C" doors into the REAL reference kernels of mpk/SpMV.cpp and
// mpk/utils.cpp so that Python (ctypes) can drive them.  This file contains no
// reference code: it includes the reference's own header from where it lies
// (-I/root/reference/mpk) and is linked with the reference's own translation
// units by oracle/Makefile into oracle/_ref/libref_spmv.so.  It is only built
// when /root/reference is present.
#include "SpMV.h" // the reference's mpk/SpMV.h

namespace {
csrmatrix make_csr(int n, int nnz, const int* ptrow, const int* indcol, const double* coef)
{
    csrmatrix a;
    a.n = n;
    a.nnz = nnz;
    a.ptrow.assign(ptrow, ptrow + n + 1);
    a.indcol.assign(indcol, indcol + nnz);
    a.coef.assign(coef, coef + nnz);
    return a;
}
} // namespace

extern "C" {

// variant: 0 SpMV_CSR (x87 scalar), 1 _OPT, 2 _FMA, 3 _AVX2
int ref_spmv_csr(int variant, int n, int nnz, const int* ptrow, const int* indcol, const double* coef,
                 const double* x, double* y)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, coef);
    double* xx = const_cast<double*>(x);
    switch (variant) {
    case 0: SpMV_CSR(y, xx, a); break;
    case 1: SpMV_CSR_OPT(y, xx, a); break;
    case 2: SpMV_CSR_FMA(y, xx, a); break;
    case 3: SpMV_CSR_AVX2(y, xx, a); break;
    default: return -1;
    }
    return 0;
}

// variant: 0 SpMV_BCSR, 1 _OPT, 2 _FMA, 3 _AVX2
int ref_spmv_bcsr(int variant, int nbrows, int nblocks, const int* ptrow, const int* indcol,
                  const double* coef, const double* x, double* y)
{
    bcsr4x4_matrix a;
    a.nrows = nbrows;
    a.nblocks = nblocks;
    a.ptrow.assign(ptrow, ptrow + nbrows + 1);
    a.indcol.assign(indcol, indcol + nblocks);
    a.coef.assign(coef, coef + 16 * (size_t)nblocks);
    switch (variant) {
    case 0: SpMV_BCSR(y, x, a); break;
    case 1: SpMV_BCSR_OPT(y, x, a); break;
    case 2: SpMV_BCSR_FMA(y, x, a); break;
    case 3: SpMV_BCSR_AVX2(y, x, a); break;
    default: return -1;
    }
    return 0;
}

// COO2CSR; outputs sized by the caller to nrow+1 / nnz / nnz. Returns ptrow[nrow].
int ref_coo2csr(int nrow, int nnz, const int* irow, const int* jcol, const double* val, int* ptrow,
                int* indcol, double* coef)
{
    csrmatrix a;
    COO2CSR(a, nrow, nnz, const_cast<int*>(irow), const_cast<int*>(jcol), const_cast<double*>(val));
    for (int i = 0; i <= nrow; i++) ptrow[i] = a.ptrow[i];
    int stored = a.ptrow[nrow];
    for (int k = 0; k < stored; k++) {
        indcol[k] = a.indcol[k];
        coef[k] = a.coef[k];
    }
    return stored;
}

// generate_BCSR4; two-pass (indcol == NULL -> count only). Returns block count.
int ref_coo2bcsr4(int nrow, int nnz, const int* irow, const int* jcol, const double* val, int* ptrow,
                  int* indcol, double* coef)
{
    bcsr4x4_matrix a;
    std::vector<std::list<std::pair<int, std::array<double, 16>>>> block_rows((nrow + 3) / 4 + 1);
    generate_BCSR4(&block_rows[0], nrow, nnz, irow, jcol, val, a);
    int nb = (int)a.indcol.size();
    if (indcol) {
        for (int i = 0; i <= a.nrows; i++) ptrow[i] = a.ptrow[i];
        for (int k = 0; k < nb; k++) indcol[k] = a.indcol[k];
        for (size_t k = 0; k < a.coef.size(); k++) coef[k] = a.coef[k];
    }
    return nb;
}

double ref_norm2(int n, const double* x)
{
    std::vector<double> v(x, x + n);
    return norm2(v);
}

double ref_rel_error(int n, const double* ref, const double* test)
{
    std::vector<double> a(ref, ref + n), b(test, test + n);
    return rel_error(a, b);
}

// Timed reference SpMV for bench.py's cpu_baseline (kind "reference"):
// reps cold calls with the reference's own flush_cache() before each, best seconds.
double ref_time_spmv_csr(int variant, int n, int nnz, const int* ptrow, const int* indcol,
                         const double* coef, const double* x, double* y, int reps, int flush)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, coef);
    double* xx = const_cast<double*>(x);
    double best = 1e300;
    for (int r = 0; r < reps; r++) {
        if (flush) flush_cache();
        auto t0 = std::chrono::high_resolution_clock::now();
        switch (variant) {
        case 0: SpMV_CSR(y, xx, a); break;
        case 1: SpMV_CSR_OPT(y, xx, a); break;
        case 2: SpMV_CSR_FMA(y, xx, a); break;
        default: SpMV_CSR_AVX2(y, xx, a); break;
        }
        auto t1 = std::chrono::high_resolution_clock::now();
        double dt = std::chrono::duration<double>(t1 - t0).count();
        if (dt < best) best = dt;
    }
    return best;
}

} // extern "C"
