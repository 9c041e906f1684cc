// oracle/ref_glue_multi0.cpp — TEST INFRASTRUCTURE.
//
// extern "C" doors into the REAL k = 2, 3, 4 matrix-powers kernels of
// mpk/SpMVmulti0.cpp (A. Suzuki's original).  That file is self-contained (own
// struct csrmatrix, own SpMV/COO2CSR, own main); oracle/Makefile compiles it
// where it lies with -Dmain=ref_multi0_main_unused and links it with this file
// into oracle/_ref/libref_multi0.so.  No reference code here: the struct below
// only re-declares the layout of mpk/SpMVmulti0.cpp:15-20 so that the
// prototypes can be written.
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
void SpMV(double* y, double* x, csrmatrix& a);
void SpM2V0(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& ptrowend1);
void SpM3V(double* w, double* z, double* y, double* x, csrmatrix& A, std::vector<int>& ptrowend1,
           std::vector<std::vector<int>>& ptrowend2);
void SpM4V(double* v, double* w, double* z, double* y, double* x, csrmatrix& A,
           std::vector<int>& ptrowend1, std::vector<std::vector<int>>& ptrowend2,
           std::vector<std::vector<std::vector<int>>>& ptrowend3);

extern "C" {

// Y = k contiguous vectors of n: Y[0]=Ax .. Y[k-1]=A^k x.
// fused = 0: k chained SpMV (mpk/SpMVmulti0.cpp:369-373 pattern)
// fused = 1: SpM2V0 / SpM3V / SpM4V with their Generate*layer tables
int ref_multi0_powers(int fused, int k, int n, int nnz, const int* ptrow, const int* indcol,
                      const double* coef, const double* x, double* Y)
{
    if (k < 1 || k > 4) return -1;
    csrmatrix a;
    a.n = n;
    a.nnz = nnz;
    a.ptrow.assign(ptrow, ptrow + n + 1);
    a.indcol.assign(indcol, indcol + nnz);
    a.coef.assign(coef, coef + nnz);
    double* xx = const_cast<double*>(x);
    double* y1 = Y;
    double* y2 = Y + (size_t)n;
    double* y3 = Y + 2 * (size_t)n;
    double* y4 = Y + 3 * (size_t)n;
    if (!fused || k == 1) {
        double* src = xx;
        for (int p = 0; p < k; p++) {
            SpMV(Y + (size_t)p * n, src, a);
            src = Y + (size_t)p * n;
        }
        return 0;
    }
    std::vector<int> e1(nnz);
    std::vector<std::vector<int>> e2(nnz);
    std::vector<std::vector<std::vector<int>>> e3(nnz);
    Generate1stlayer(e1, a);
    if (k == 2) { SpM2V0(y2, y1, xx, a, e1); return 0; }
    Generate2ndlayer(e2, a, e1);
    if (k == 3) { SpM3V(y3, y2, y1, xx, a, e1, e2); return 0; }
    Generate3rdlayer(e3, a, e1, e2);
    SpM4V(y4, y3, y2, y1, xx, a, e1, e2, e3);
    return 0;
}

// The nested first-touch tables of Generate1st/2nd/3rdlayer (mpk/SpMVmulti0.cpp:22-40, :106-130,
// :157-187), flattened in traversal order so that Python can compare them:
//   e1[ia]                                   nnz entries
//   len2[ia] = ptrowend2[ia].size(), e2 = concatenation over ia
//   len3[q]  = ptrowend3[ia][jjb].size() for q running over (ia, jjb) in order, e3 = concatenation
// Two-pass: call with e2 == NULL to get the counts (*n2 = total e2 entries = number of (ia,jjb)
// pairs, *n3 = total e3 entries), then again with buffers.
int ref_multi0_layers(int n, int nnz, const int* ptrow, const int* indcol, int* e1, int* len2, int* e2,
                      int* len3, int* e3, long long* n2, long long* n3)
{
    csrmatrix a;
    a.n = n;
    a.nnz = nnz;
    a.ptrow.assign(ptrow, ptrow + n + 1);
    a.indcol.assign(indcol, indcol + nnz);
    a.coef.assign((size_t)nnz, 0.0);
    std::vector<int> t1(nnz);
    std::vector<std::vector<int>> t2(nnz);
    std::vector<std::vector<std::vector<int>>> t3(nnz);
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

} // extern "C"
