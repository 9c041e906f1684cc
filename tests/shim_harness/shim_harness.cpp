// tests/shim_harness/shim_harness.cpp — TEST INFRASTRUCTURE.
//
// extern "C" doors into the C++ symbols that libmpk_mi355.so EXPORTS UNDER THE REFERENCE'S NAMES
// (include/SpMV.h), so that Python tests can call the very functions a reference driver would bind —
// not a Python mirror of them — and compare their output bit for bit with the goldens produced by the
// reference's object code (tests/golden/).  The doors have the same shapes as oracle/ref_glue_*.cpp's,
// so one helper drives both sides.  The integer builders (COO2CSR, generate_BCSR4, Generate*layer*)
// run on the host and need no GPU; the compute doors do.
#include "SpMV.h" // include/SpMV.h of this repo

#include <cstddef>

namespace {
csrmatrix make_csr(int n, int nnz, const int* ptrow, const int* indcol, const double* coef)
{
    csrmatrix a;
    a.n = n;
    a.nnz = nnz;
    a.ptrow.assign(ptrow, ptrow + n + 1);
    a.indcol.assign(indcol, indcol + nnz);
    if (coef) a.coef.assign(coef, coef + nnz);
    else a.coef.assign((size_t)nnz, 0.0);
    return a;
}
bcsr4x4_matrix make_bcsr(int nbrows, int nblocks, const int* ptrow, const int* indcol, const double* coef)
{
    bcsr4x4_matrix a;
    a.nrows = nbrows;
    a.nblocks = nblocks;
    a.ptrow.assign(ptrow, ptrow + nbrows + 1);
    a.indcol.assign(indcol, indcol + nblocks);
    if (coef) a.coef.assign(coef, coef + 16 * (size_t)nblocks);
    return a;
}
} // namespace

extern "C" {

// ---- host-side integer work (no GPU) ------------------------------------------------------------

// COO2CSR; outputs sized by the caller to nrow+1 / nnz / nnz.  Returns ptrow[nrow]; *nnz_field = a.nnz.
int shim_coo2csr(int nrow, int nnz, const int* irow, const int* jcol, const double* val, int* ptrow, int* indcol,
                 double* coef, int* nnz_field)
{
    csrmatrix a;
    COO2CSR(a, nrow, nnz, const_cast<int*>(irow), const_cast<int*>(jcol), const_cast<double*>(val));
    for (int i = 0; i <= nrow; i++) ptrow[i] = a.ptrow[i];
    const int stored = a.ptrow[nrow];
    for (int k = 0; k < stored; k++) {
        indcol[k] = a.indcol[k];
        coef[k] = a.coef[k];
    }
    if (nnz_field) *nnz_field = a.nnz;
    return stored;
}

// generate_BCSR4; two-pass (indcol == NULL -> count only).  Returns the block count.
int shim_coo2bcsr4(int nrow, int nnz, const int* irow, const int* jcol, const double* val, int* ptrow, int* indcol,
                   double* coef)
{
    bcsr4x4_matrix a;
    std::vector<std::list<std::pair<int, std::array<double, 16> > > > block_rows((size_t)(nrow + 3) / 4 + 1);
    generate_BCSR4(&block_rows[0], nrow, nnz, irow, jcol, val, a);
    const int nb = (int)a.indcol.size();
    if (indcol) {
        for (int i = 0; i <= a.nrows; i++) ptrow[i] = a.ptrow[i];
        for (int k = 0; k < nb; k++) indcol[k] = a.indcol[k];
        for (size_t k = 0; k < a.coef.size(); k++) coef[k] = a.coef[k];
    }
    return nb;
}

int shim_gen_layer1(int n, int nnz, const int* ptrow, const int* indcol, int* end1)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, nullptr);
    std::vector<int> t;
    Generate1stlayer(t, a);
    if ((int)t.size() < nnz) return -1;
    for (int k = 0; k < nnz; k++) end1[k] = t[k];
    return 0;
}

int shim_gen_layer1_bcsr4(int nbrows, int nblocks, const int* ptrow, const int* indcol, int* endB)
{
    bcsr4x4_matrix a = make_bcsr(nbrows, nblocks, ptrow, indcol, nullptr);
    std::vector<int> t;
    Generate1stlayer_BCSR4(t, a);
    if ((int)t.size() < nblocks) return -1;
    for (int k = 0; k < nblocks; k++) endB[k] = t[k];
    return 0;
}

// nested tables flattened exactly like oracle/ref_glue_multi0.cpp:ref_multi0_layers
int shim_layers(int n, int nnz, const int* ptrow, const int* indcol, int* e1, int* len2, int* e2, int* len3, int* e3,
                long long* n2, long long* n3)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, nullptr);
    std::vector<int> t1(nnz);
    std::vector<std::vector<int> > t2(nnz);
    std::vector<std::vector<std::vector<int> > > t3(nnz);
    Generate1stlayer(t1, a);
    Generate2ndlayer(t2, a, t1);
    Generate3rdlayer(t3, a, t1, t2);
    long long c2 = 0, c3 = 0;
    for (int ia = 0; ia < nnz; ia++) {
        if (e1) e1[ia] = t1[ia];
        if (len2) len2[ia] = (int)t2[ia].size();
        for (size_t jjb = 0; jjb < t2[ia].size(); jjb++) {
            if (e2) e2[c2] = t2[ia][jjb];
            const std::vector<int>* v3 = jjb < t3[ia].size() ? &t3[ia][jjb] : nullptr;
            if (len3) len3[c2] = v3 ? (int)v3->size() : 0;
            c2++;
            if (v3)
                for (size_t kkc = 0; kkc < v3->size(); kkc++) {
                    if (e3) e3[c3] = (*v3)[kkc];
                    c3++;
                }
        }
    }
    if (n2) *n2 = c2;
    if (n3) *n3 = c3;
    return 0;
}

// ---- compute through the shim (GPU) ----------------------------------------------------------------

// variant: 0 SpMV_CSR, 1 _OPT, 2 _FMA, 3 _AVX2, 4 SpMV (mpk/SpMVmulti0.cpp's name)
int shim_spmv_csr(int variant, int n, int nnz, const int* ptrow, const int* indcol, const double* coef, const double* x,
                  double* y)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, coef);
    double* xx = const_cast<double*>(x);
    switch (variant) {
    case 0: SpMV_CSR(y, xx, a); break;
    case 1: SpMV_CSR_OPT(y, xx, a); break;
    case 2: SpMV_CSR_FMA(y, xx, a); break;
    case 3: SpMV_CSR_AVX2(y, xx, a); break;
    case 4: SpMV(y, xx, a); break;
    default: return -1;
    }
    mi355_invalidate(a);
    return 0;
}

// Seconds per SpMV_CSR call on one csrmatrix object kept alive across `reps` calls (first call excluded): what the reference's
// calling convention costs through the shim — full-content hash of the caller's arrays, x in and y out over PCIe, the kernel.
// trust != 0: with mi355_assume_unchanged(true), i.e. without the hash.
double shim_time_spmv_csr(int n, int nnz, const int* ptrow, const int* indcol, const double* coef, const double* x, double* y,
                          int reps, int trust)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, coef);
    double* xx = const_cast<double*>(x);
    SpMV_CSR(y, xx, a);
    mi355_assume_unchanged(trust != 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) SpMV_CSR(y, xx, a);
    const auto t1 = std::chrono::steady_clock::now();
    mi355_assume_unchanged(false);
    mi355_invalidate(a);
    return std::chrono::duration<double>(t1 - t0).count() / reps;
}

// The stale-copy scenario: product, then `nedit` coefficients rewritten IN PLACE in the same storage
// (same addresses, same pattern), product again.  y0 / y1 = results before / after the edit.
int shim_spmv_csr_inplace_edit(int n, int nnz, const int* ptrow, const int* indcol, const double* coef, const double* x,
                               int nedit, const int* edit_idx, const double* edit_val, double* y0, double* y1)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, coef);
    double* xx = const_cast<double*>(x);
    SpMV_CSR(y0, xx, a);
    for (int e = 0; e < nedit; e++) a.coef[edit_idx[e]] = edit_val[e];
    SpMV_CSR(y1, xx, a);
    mi355_invalidate(a);
    return 0;
}

int shim_spmv_bcsr_inplace_edit(int nbrows, int nblocks, const int* ptrow, const int* indcol, const double* coef,
                                const double* x, int nedit, const int* edit_idx, const double* edit_val, double* y0, double* y1)
{
    bcsr4x4_matrix a = make_bcsr(nbrows, nblocks, ptrow, indcol, coef);
    SpMV_BCSR(y0, x, a);
    for (int e = 0; e < nedit; e++) a.coef[edit_idx[e]] = edit_val[e];
    SpMV_BCSR(y1, x, a);
    mi355_invalidate(a);
    return 0;
}

// variant: 0 SpMV_BCSR, 1 _OPT, 2 _FMA, 3 _AVX2
int shim_spmv_bcsr(int variant, int nbrows, int nblocks, const int* ptrow, const int* indcol, const double* coef,
                   const double* x, double* y)
{
    bcsr4x4_matrix a = make_bcsr(nbrows, nblocks, ptrow, indcol, coef);
    switch (variant) {
    case 0: SpMV_BCSR(y, x, a); break;
    case 1: SpMV_BCSR_OPT(y, x, a); break;
    case 2: SpMV_BCSR_FMA(y, x, a); break;
    case 3: SpMV_BCSR_AVX2(y, x, a); break;
    default: return -1;
    }
    mi355_invalidate(a);
    return 0;
}

// variant: 0 SpM2V_CSR, 1 _OPT, 2 _FMA, 3 _AVX2, 4 SpM2V0, 5 SpM2V
int shim_spm2v_csr(int variant, int n, int nnz, const int* ptrow, const int* indcol, const double* coef, const double* x,
                   double* y, double* z)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, coef);
    std::vector<int> t;
    Generate1stlayer(t, a);
    double* xx = const_cast<double*>(x);
    switch (variant) {
    case 0: SpM2V_CSR(z, y, xx, a, t); break;
    case 1: SpM2V_CSR_OPT(z, y, xx, a, t); break;
    case 2: SpM2V_CSR_FMA(z, y, xx, a, t); break;
    case 3: SpM2V_CSR_AVX2(z, y, xx, a, t); break;
    case 4: SpM2V0(z, y, xx, a, t); break;
    case 5: SpM2V(z, y, xx, a, t); break;
    default: return -1;
    }
    mi355_invalidate(a);
    return 0;
}

// variant: 0 SpM2V_BCSR, 1 _OPT, 2 _FMA, 3 _AVX2
int shim_spm2v_bcsr(int variant, int nbrows, int nblocks, const int* ptrow, const int* indcol, const double* coef,
                    const double* x, double* y, double* z)
{
    bcsr4x4_matrix a = make_bcsr(nbrows, nblocks, ptrow, indcol, coef);
    std::vector<int> t;
    Generate1stlayer_BCSR4(t, a);
    double* xx = const_cast<double*>(x);
    switch (variant) {
    case 0: SpM2V_BCSR(z, y, xx, a, t); break;
    case 1: SpM2V_BCSR_OPT(z, y, xx, a, t); break;
    case 2: SpM2V_BCSR_FMA(z, y, xx, a, t); break;
    case 3: SpM2V_BCSR_AVX2(z, y, xx, a, t); break;
    default: return -1;
    }
    mi355_invalidate(a);
    return 0;
}

// which: 3 SpM3V, 4 SpM4V, 5 SpM4V_AVX2; Y = `k` contiguous vectors of n, Y[p] = A^(p+1) x
int shim_powers(int which, int n, int nnz, const int* ptrow, const int* indcol, const double* coef, const double* x, double* Y)
{
    csrmatrix a = make_csr(n, nnz, ptrow, indcol, coef);
    std::vector<int> e1(nnz);
    std::vector<std::vector<int> > e2(nnz);
    std::vector<std::vector<std::vector<int> > > e3(nnz);
    Generate1stlayer(e1, a);
    Generate2ndlayer(e2, a, e1);
    double* xx = const_cast<double*>(x);
    double *y1 = Y, *y2 = Y + (size_t)n, *y3 = Y + 2 * (size_t)n, *y4 = Y + 3 * (size_t)n;
    if (which == 3) SpM3V(y3, y2, y1, xx, a, e1, e2);
    else {
        Generate3rdlayer(e3, a, e1, e2);
        if (which == 4) SpM4V(y4, y3, y2, y1, xx, a, e1, e2, e3);
        else if (which == 5) SpM4V_AVX2(y4, y3, y2, y1, x, a, e1, e2, e3);
        else return -1;
    }
    mi355_invalidate(a);
    return 0;
}

int shim_orthogonalize3(int n, const double* b, const double* x1, double* x3, double alpha)
{
    std::vector<double> bv(b, b + n), xv(x1, x1 + n), out((size_t)n, 0.0);
    orthogonalize(n, bv, xv, out, alpha);
    for (int i = 0; i < n; i++) x3[i] = out[i];
    return 0;
}

int shim_orthogonalize_inplace(int n, const double* x, double* y, double alpha)
{
    std::vector<double> xv(x, x + n), yv(y, y + n);
    orthogonalize(n, xv, yv, alpha);
    for (int i = 0; i < n; i++) y[i] = yv[i];
    return 0;
}

int shim_orthonormalize_against_basis(int n, int m, const double* basis, double* y)
{
    std::vector<std::vector<double> > B((size_t)m);
    for (int j = 0; j < m; j++) B[j].assign(basis + (size_t)j * n, basis + (size_t)(j + 1) * n);
    std::vector<double> yv(y, y + n);
    orthonormalize_against_basis(n, B, yv);
    for (int i = 0; i < n; i++) y[i] = yv[i];
    return 0;
}

double shim_norm2(int n, const double* x)
{
    std::vector<double> v(x, x + n);
    return norm2(v);
}

double shim_rel_error(int n, const double* ref, const double* test)
{
    std::vector<double> a(ref, ref + n), b(test, test + n);
    return rel_error(a, b);
}

} // extern "C"
