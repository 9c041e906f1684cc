/*
 * oracle/cpu_ref.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's CSR SpMV / matrix-powers path
 * (aantoine890/navierstokes, mpk/).  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check and time
 * the HIP path against the reference's algorithm on machines where
 * /root/reference is absent.  Nothing under navierstokes_amd/ may import,
 * link or call it; the product path fails loudly without its HIP library.
 *
 * Parity pinning: every function below is compared against the reference's
 * own object code (oracle/_ref/, built from /root/reference/mpk by
 * oracle/Makefile) in tests/test_oracle_vs_reference.py when the reference is
 * present, and against the committed golden vectors tests/golden/ (generated
 * from that object code by tests/golden/make_golden.py) everywhere else.
 *   - orc_spmv_csr_fma     bit-equal to SpMV_CSR_OPT / SpMV_CSR_FMA
 *   - orc_spmv_csr_x87     bit-equal to SpMV_CSR (x87 build, per-term rounding)
 *   - orc_spm2v_fused      bit-equal to SpM2V_CSR_OPT, = 2 chained fma SpMVs
 *   - orc_spmkv_fused      k<=4 first-touch traversal; arith fma bit-equal to
 *                          SpM3V, arith x87 bit-equal to SpM2V0 / SpM4V, arith
 *                          avx2row bit-equal to SpM4V_AVX2 (mpk/SpMVmulti-1.cpp)
 *   - orc_gen_layers       bit-equal to Generate1st/2nd/3rdlayer (nested tables)
 *   - orc_orthogonalize    bit-equal to orthogonalize of mpk/old/SpMVmulti.cpp:164-169
 *   - orc_orthogonalize_inplace  bit-equal to orthogonalize of mpk/2SpMV.cpp:3-11
 *   - orc_mgs              bit-equal to orthonormalize_against_basis, mpk/2SpMV.cpp:13-28
 *
 * All citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ SpMV */

/* y = A x, each row a sequential fma chain in CSR order.
 * Follows mpk/SpMV.cpp:41-56 (SpMV_CSR_FMA: y[i] = __builtin_fma(coef, x[j], y[i]))
 * and :23-38 (SpMV_CSR_OPT, which GCC contracts to the same chain). */
void orc_spmv_csr_fma(int n, const int *ptrow, const int *indcol, const double *coef,
                      const double *x, double *y)
{
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int ia = ptrow[i]; ia < ptrow[i + 1]; ia++) s = fma(coef[ia], x[indcol[ia]], s);
        y[i] = s;
    }
}

/* y = A x with separate IEEE-double multiply and add (no contraction).
 * This is what mpk/SpMV.cpp:6-20 (SpMV_CSR) would compute on an SSE2 target;
 * the reference builds it for x87 instead, see orc_spmv_csr_x87. */
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
void orc_spmv_csr_muladd(int n, const int *ptrow, const int *indcol, const double *coef,
                         const double *x, double *y)
{
    for (int i = 0; i < n; i++) {
        volatile double s = 0.0;
        for (int ia = ptrow[i]; ia < ptrow[i + 1]; ia++) {
            volatile double p = coef[ia] * x[indcol[ia]];
            s = s + p;
        }
        y[i] = s;
    }
}
#pragma GCC pop_options

/* One accumulation step s + c*x in one of the reference's three arithmetics:
 *   ARITH_FMA    s = fma(c, x, s)               (_OPT/_FMA variants, SSE2+FMA)
 *   ARITH_X87    the "pure sequential" variants are compiled
 *                target("no-sse,no-avx2,no-fma") (mpk/SpMV.cpp:5, mpk/SpM2V.cpp:79,
 *                mpk/SpMVmulti0.cpp:42-43,189-190), i.e. x87: product and sum are
 *                formed with a 64-bit significand and the store to y[i] after
 *                EVERY term rounds to double.  long double is that x87 format
 *                on x86-64 Linux.  Verified bit-for-bit against the reference's
 *                object code in tests/test_oracle_vs_reference.py.
 *   ARITH_MULADD IEEE double multiply, then add (an SSE2 build without FMA). */
enum { ARITH_FMA = 0, ARITH_X87 = 1, ARITH_MULADD = 2, ARITH_AVX2ROW = 3 };
/* ARITH_AVX2ROW (orc_spmkv_fused only): SpM4V_AVX2, mpk/SpMVmulti-1.cpp:434-493 — the innermost
 * row sum runs four interleaved fma chains (one per AVX lane) over the groups of four terms, folds
 * them as (l0 + l2) + (l1 + l3), continues with a scalar fma chain over the <4 remainder and is
 * then ADDED to y1[l] (0.0 at first touch); the upper levels are plain fma updates. */

static inline double acc_term(int arith, double c, double x, double s)
{
    if (arith == ARITH_FMA || arith == ARITH_AVX2ROW) return fma(c, x, s);
    if (arith == ARITH_X87) return (double)((long double)s + (long double)c * (long double)x);
    volatile double p = c * x;
    volatile double r = s + p;
    return r;
}

/* y = A x exactly as mpk/SpMV.cpp:5-20 (SpMV_CSR, x87 build) evaluates it. */
void orc_spmv_csr_x87(int n, const int *ptrow, const int *indcol, const double *coef,
                      const double *x, double *y)
{
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int ia = ptrow[i]; ia < ptrow[i + 1]; ia++) s = acc_term(ARITH_X87, coef[ia], x[indcol[ia]], s);
        y[i] = s;
    }
}

/* Y[p] = A^(p+1) x for p = 0..k-1 by k chained SpMVs — the "SpMV chain" column
 * of mpk/SpMVmulti0.cpp:369-373.  Y is k contiguous vectors of n. */
void orc_spmk_chain(int k, int n, const int *ptrow, const int *indcol, const double *coef,
                    const double *x, double *Y)
{
    const double *src = x;
    for (int p = 0; p < k; p++) {
        orc_spmv_csr_fma(n, ptrow, indcol, coef, src, Y + (size_t)p * n);
        src = Y + (size_t)p * n;
    }
}

/* ------------------------------------------------- fused matrix powers (CPU) */

/* First-touch table of the fused 2-step kernel, mpk/SpM2V.cpp:5-26
 * (= mpk/SpMVmulti0.cpp:22-40): for nonzero ia=(i,j) in CSR traversal order,
 * end1[ia] = ptrow[j+1] the first time column j is met, else ptrow[j]. */
void orc_gen_layer1(int n, const int *ptrow, const int *indcol, int *end1)
{
    unsigned char *seen = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; i++)
        for (int ia = ptrow[i]; ia < ptrow[i + 1]; ia++) {
            int j = indcol[ia];
            if (seen[j]) end1[ia] = ptrow[j];
            else { end1[ia] = ptrow[j + 1]; seen[j] = 1; }
        }
    free(seen);
}

/* z = A(Ax), y = Ax in one traversal, mpk/SpM2V.cpp:135-169 (SpM2V_CSR_OPT;
 * scalar twin :79-112): y[j] is accumulated lazily, in CSR order, the first
 * time row i references column j, then used in z[i].  Rows of y that no row
 * references as a column stay 0 (SURVEY.md §8a-10 caveat). */
void orc_spm2v_fused(int n, const int *ptrow, const int *indcol, const double *coef,
                     const int *end1, const double *x, double *y, double *z)
{
    for (int i = 0; i < n; i++) y[i] = z[i] = 0.0;
    for (int i = 0; i < n; i++)
        for (int ia = ptrow[i]; ia < ptrow[i + 1]; ia++) {
            int j = indcol[ia];
            for (int jb = ptrow[j]; jb < end1[ia]; jb++) y[j] = fma(coef[jb], x[indcol[jb]], y[j]);
            z[i] = fma(coef[ia], y[j], z[i]);
        }
}

/* The same traversal as the x87 object code of mpk/SpM2V.cpp:79-112 (SpM2V_CSR,
 * target("no-sse,no-avx2,no-fma")) evaluates it with g++ 11.4 -O3: the inner
 * jb loop keeps y[j] in an 80-bit register for the whole row and rounds once
 * in the store, and when the row was just computed the z update multiplies by
 * that still-extended value; z[i] itself is rounded after every term.  This is
 * a compiler artefact, restated only to pin the oracle bit-for-bit to every
 * variant the reference ships (tests/test_oracle_golden.py); it also shows why
 * parity with the x87 variants can only be asked to ~1e-16, not bitwise. */
void orc_spm2v_fused_x87(int n, const int *ptrow, const int *indcol, const double *coef,
                         const int *end1, const double *x, double *y, double *z)
{
    for (int i = 0; i < n; i++) y[i] = z[i] = 0.0;
    for (int i = 0; i < n; i++)
        for (int ia = ptrow[i]; ia < ptrow[i + 1]; ia++) {
            int j = indcol[ia];
            long double yj = (long double)y[j];
            if (end1[ia] > ptrow[j]) {
                for (int jb = ptrow[j]; jb < end1[ia]; jb++)
                    yj += (long double)coef[jb] * (long double)x[indcol[jb]];
                y[j] = (double)yj;
            }
            z[i] = (double)((long double)z[i] + (long double)coef[ia] * yj);
        }
}

/* k-level (k = 2, 3, 4) first-touch traversal: Y[0]=Ax ... Y[k-1]=A^k x.
 * Restates SpM3V / SpM4V with Generate2ndlayer / Generate3rdlayer,
 * mpk/SpMVmulti0.cpp:106-130, :132-155, :157-187, :189-221.  The reference
 * materialises the first-touch decisions as nested tables ptrowend2[ia][jjb],
 * ptrowend3[ia][jjb][kkc]; each table entry is "full row" exactly when the
 * index is met for the first time at that depth of this very traversal
 * (independent masks mask2, mask3 per level), so evaluating the masks on the
 * fly visits the same (row, range) pairs in the same order. */
static void level_visit(int arith, int depth, int k, int row, const int *ptrow, const int *indcol,
                        const double *coef, const double *x, double *Y, int n,
                        unsigned char *seen)
{
    /* computes Y[depth][row] assuming it is visited for the first time */
    double *out = Y + (size_t)depth * n;
    if (depth == 0 && arith == ARITH_AVX2ROW) {
        double l[4] = {0.0, 0.0, 0.0, 0.0};
        int ld = ptrow[row];
        const int end = ptrow[row + 1];
        for (; ld <= end - 4; ld += 4)
            for (int t = 0; t < 4; t++) l[t] = fma(coef[ld + t], x[indcol[ld + t]], l[t]);
        double sum = 0.0;
        sum += (l[0] + l[2]) + (l[1] + l[3]);
        for (; ld < end; ld++) sum = fma(coef[ld], x[indcol[ld]], sum);
        out[row] += sum;
        return;
    }
    for (int ia = ptrow[row]; ia < ptrow[row + 1]; ia++) {
        int j = indcol[ia];
        if (depth == 0) {
            out[row] = acc_term(arith, coef[ia], x[j], out[row]);
        } else {
            unsigned char *sn = seen + (size_t)(depth - 1) * n;
            if (!sn[j]) {
                sn[j] = 1;
                level_visit(arith, depth - 1, k, j, ptrow, indcol, coef, x, Y, n, seen);
            }
            out[row] = acc_term(arith, coef[ia], (Y + (size_t)(depth - 1) * n)[j], out[row]);
        }
    }
}

/* arith: 0 fma (SpM2V_CSR_OPT, SpM3V), 1 x87 (SpM2V_CSR, SpM2V0, SpM4V), 2 mul+add, 3 SpM4V_AVX2 */
void orc_spmkv_fused(int arith, int k, int n, const int *ptrow, const int *indcol, const double *coef,
                     const double *x, double *Y)
{
    memset(Y, 0, (size_t)k * n * sizeof(double));
    unsigned char *seen = (unsigned char *)calloc((size_t)(k > 1 ? k - 1 : 1) * (n > 0 ? n : 1), 1);
    for (int i = 0; i < n; i++) level_visit(arith, k - 1, k, i, ptrow, indcol, coef, x, Y, n, seen);
    free(seen);
}

/* The nested first-touch tables of the k = 3, 4 traversals, mpk/SpMVmulti0.cpp:106-130
 * (Generate2ndlayer) and :157-187 (Generate3rdlayer), flattened in traversal order:
 *   e1[ia]                               (Generate1stlayer, :22-40)
 *   len2[ia] = size of ptrowend2[ia] = e1[ia] - ptrow[j];  e2 = their concatenation
 *   len3[q]  = size of ptrowend3[ia][jjb], q over (ia, jjb) in order; e3 = their concatenation
 * Each level keeps its own mask: an index is "first met" at level L the first time the level-L
 * loop of THIS traversal reaches it.  Two-pass: buffers may be NULL to obtain *n2 (entries of e2)
 * and *n3 (entries of e3). */
void orc_gen_layers(int n, const int *ptrow, const int *indcol, int *e1, int *len2, int *e2, int *len3,
                    int *e3, long long *n2, long long *n3)
{
    unsigned char *m1 = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    unsigned char *m2 = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    unsigned char *m3 = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    long long c2 = 0, c3 = 0;
    for (int i = 0; i < n; i++)
        for (int ia = ptrow[i]; ia < ptrow[i + 1]; ia++) {
            const int j = indcol[ia];
            int end1 = ptrow[j];
            if (!m1[j]) { end1 = ptrow[j + 1]; m1[j] = 1; }
            if (e1) e1[ia] = end1;
            if (len2) len2[ia] = end1 - ptrow[j];
            for (int jb = ptrow[j]; jb < end1; jb++) {
                const int k = indcol[jb];
                int end2 = ptrow[k];
                if (!m2[k]) { end2 = ptrow[k + 1]; m2[k] = 1; }
                if (e2) e2[c2] = end2;
                if (len3) len3[c2] = end2 - ptrow[k];
                c2++;
                for (int kc = ptrow[k]; kc < end2; kc++) {
                    const int l = indcol[kc];
                    int end3 = ptrow[l];
                    if (!m3[l]) { end3 = ptrow[l + 1]; m3[l] = 1; }
                    if (e3) e3[c3] = end3;
                    c3++;
                }
            }
        }
    if (n2) *n2 = c2;
    if (n3) *n3 = c3;
    free(m1); free(m2); free(m3);
}

/* First-touch table of block rows, mpk/SpM2V.cpp:28-46 (Generate1stlayer_BCSR4): block m = (bi, bj)
 * gets the full block row bj the first time block column bj is met, an empty range afterwards. */
void orc_gen_layer1_bcsr4(int nbrows, const int *ptrow, const int *indcol, int *endB)
{
    unsigned char *seen = (unsigned char *)calloc((size_t)(nbrows > 0 ? nbrows : 1), 1);
    for (int bi = 0; bi < nbrows; bi++)
        for (int m = ptrow[bi]; m < ptrow[bi + 1]; m++) {
            const int bj = indcol[m];
            if (seen[bj]) endB[m] = ptrow[bj];
            else { endB[m] = ptrow[bj + 1]; seen[bj] = 1; }
        }
    free(seen);
}

/* -------------------------------------------------------------- BCSR 4x4 */

/* y = A x for row-major 4x4 blocks, fma order of mpk/SpMV.cpp:150-178
 * (SpMV_BCSR_FMA): per block, for i, for j: y[4bi+i] = fma(blk[4i+j], x[4bj+j], y[4bi+i]). */
void orc_spmv_bcsr4_fma(int nbrows, const int *ptrow, const int *indcol, const double *coef,
                        const double *x, double *y)
{
    for (int bi = 0; bi < nbrows; bi++) {
        double acc[4] = {0, 0, 0, 0};
        for (int ia = ptrow[bi]; ia < ptrow[bi + 1]; ia++) {
            const double *blk = coef + 16 * (size_t)ia;
            const double *xb = x + 4 * (size_t)indcol[ia];
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) acc[i] = fma(blk[4 * i + j], xb[j], acc[i]);
        }
        for (int i = 0; i < 4; i++) y[4 * bi + i] = acc[i];
    }
}

/* ------------------------------------------------------------ BLAS-1 bits */

/* sqrt(sum x^2), sequential — mpk/utils.cpp:131-136 */
double orc_norm2(int n, const double *x)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s = fma(x[i], x[i], s);
    return sqrt(s);
}

/* ||ref - test||_2 / ||ref||_2 — mpk/utils.cpp:138-143, THE parity metric */
double orc_rel_error(int n, const double *ref, const double *test)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) {
        double d = ref[i] - test[i];
        s = fma(d, d, s);
    }
    return sqrt(s) / orc_norm2(n, ref);
}

/* sequential fma dot: beta = fma(a[i], b[i], beta) — what the volatile accumulation loop of
 * mpk/2SpMV.cpp:5-7 compiles to (g++ 11.4 -O3, FMA target: one vfmadd231sd per element) */
double orc_dot(int n, const double *a, const double *b)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s = fma(a[i], b[i], s);
    return s;
}

/* std::inner_product(b, x1) of mpk/SpMVmulti.cpp:147 (= mpk/old/SpMVmulti.cpp:165) and the dot loops of
 * orthonormalize_against_basis, mpk/2SpMV.cpp:15-17, as g++ 11.4 -O3 vectorises them for an FMA
 * target: the PRODUCTS are formed four at a time and rounded (vmulpd), then added to the running sum
 * one by one in index order (an in-order reduction: no reassociation without -ffast-math); a
 * remaining pair is handled the same way, and a last odd element by one scalar fma.  Pinned bitwise
 * against the reference's object code (tests/golden/blas1_*.npz). */
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
double orc_dot_gccvec(int n, const double *a, const double *b)
{
    volatile double s = 0.0;
    int i = 0;
    for (; i + 4 <= n; i += 4)
        for (int t = 0; t < 4; t++) { volatile double p = a[i + t] * b[i + t]; s = s + p; }
    if (n - i >= 2) {
        for (int t = 0; t < 2; t++) { volatile double p = a[i + t] * b[i + t]; s = s + p; }
        i += 2;
    }
    double r = s;
    if (i < n) r = fma(a[i], b[i], r);
    return r;
}
#pragma GCC pop_options

/* out[i] = fma(-ab, b[i], x1[i]): the update half of both orthogonalize forms as compiled
 * (vmulsd alpha*beta once per element, then vfnmadd132sd: mpk/2SpMV.cpp:9-10, mpk/SpMVmulti.cpp:148-150) */
void orc_ortho_update(int n, double ab, const double *b, const double *x1, double *out)
{
    for (int i = 0; i < n; i++) out[i] = fma(-ab, b[i], x1[i]);
}

/* out = x1 - alpha*beta*b with beta = b.x1 — orthogonalize(nrow, b, x1, x3, alpha),
 * mpk/SpMVmulti.cpp:146-151 (compilable copy: mpk/old/SpMVmulti.cpp:164-169); returns beta. */
double orc_orthogonalize(int n, const double *b, const double *x1, double *out, double alpha)
{
    double beta = orc_dot_gccvec(n, b, x1);
    orc_ortho_update(n, alpha * beta, b, x1, out);
    return beta;
}

/* y -= alpha*(x.y)*x in place — orthogonalize(nrow, x, y, alpha), mpk/2SpMV.cpp:3-11; returns beta. */
double orc_orthogonalize_inplace(int n, const double *x, double *y, double alpha)
{
    double beta = orc_dot(n, x, y);
    orc_ortho_update(n, alpha * beta, x, y, y);
    return beta;
}

/* orthonormalize_against_basis(nrow, basis, y), mpk/2SpMV.cpp:13-28: for each basis vector in turn
 * dot = y.v (on the y updated so far), y -= dot*v (compiled to vfnmadd: y = fma(-dot, v, y)); the
 * norm computed at the end is discarded by the reference and nothing is normalised.  basis = m
 * contiguous vectors of n; dots (may be NULL) receives the m coefficients. */
void orc_mgs(int n, int m, const double *basis, double *y, double *dots)
{
    for (int j = 0; j < m; j++) {
        const double *v = basis + (size_t)j * n;
        const double d = orc_dot_gccvec(n, y, v);
        orc_ortho_update(n, d, v, y, y);
        if (dots) dots[j] = d;
    }
}

/* y += a x */
void orc_axpy(int n, double a, const double *x, double *y)
{
    for (int i = 0; i < n; i++) y[i] = fma(a, x[i], y[i]);
}

/* ------------------------------------------------------- format builders */

typedef struct { int r, c, k; double v; } orc_ent;

static int ent_cmp(const void *pa, const void *pb)
{
    const orc_ent *a = (const orc_ent *)pa, *b = (const orc_ent *)pb;
    if (a->r != b->r) return a->r < b->r ? -1 : 1;
    if (a->c != b->c) return a->c < b->c ? -1 : 1;
    return a->k < b->k ? -1 : (a->k > b->k);
}

/* COO -> CSR with the reference's rules, mpk/utils.cpp:5-43 + :97-127:
 * columns ascending per row; for a duplicated (i,j) the FIRST value in COO
 * order is kept and later ones are dropped.  Returns the number of stored
 * nonzeros (= ptrow[nrow]); the reference leaves a.nnz at the COO count. */
int orc_coo2csr(int nrow, int nnz, const int *irow, const int *jcol, const double *val,
                int *ptrow, int *indcol, double *coef)
{
    orc_ent *e = (orc_ent *)malloc(sizeof(orc_ent) * (size_t)(nnz > 0 ? nnz : 1));
    for (int k = 0; k < nnz; k++) { e[k].r = irow[k]; e[k].c = jcol[k]; e[k].k = k; e[k].v = val[k]; }
    qsort(e, (size_t)nnz, sizeof(orc_ent), ent_cmp);
    int out = 0, row = 0;
    ptrow[0] = 0;
    for (int k = 0; k < nnz; k++) {
        if (k > 0 && e[k].r == e[k - 1].r && e[k].c == e[k - 1].c) continue; /* first wins */
        while (row < e[k].r) ptrow[++row] = out;
        indcol[out] = e[k].c;
        coef[out] = e[k].v;
        out++;
    }
    while (row < nrow) ptrow[++row] = out;
    free(e);
    return out;
}

/* COO -> BCSR 4x4 with the reference's rules, mpk/utils.cpp:45-95: block rows
 * = nrow/4 (truncating), blocks of a block row in order of FIRST APPEARANCE
 * in the COO stream (not sorted), 16 values row-major, duplicate (i,j)
 * OVERWRITES (last wins).  Two-pass: call with indcol == NULL to get the
 * block count.  Returns the number of blocks. */
int orc_coo2bcsr4(int nrow, int nnz, const int *irow, const int *jcol, const double *val,
                  int *ptrow, int *indcol, double *coef)
{
    int nb = nrow / 4;
    /* appearance-ordered block list per block row, as singly linked chains */
    int cap = nnz > 0 ? nnz : 1;
    int *head = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
    int *tail = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
    int *next = (int *)malloc(sizeof(int) * (size_t)cap);
    int *bcol = (int *)malloc(sizeof(int) * (size_t)cap);
    double *bval = (double *)calloc((size_t)cap * 16, sizeof(double));
    for (int b = 0; b <= nb; b++) head[b] = tail[b] = -1;
    int nblk = 0;
    for (int k = 0; k < nnz; k++) {
        int bi = irow[k] / 4, bj = jcol[k] / 4;
        if (bi >= nb) continue; /* rows beyond 4*(nrow/4) are dropped in the flattening loop */
        int p = head[bi];
        while (p >= 0 && bcol[p] != bj) p = next[p];
        if (p < 0) {
            p = nblk++;
            bcol[p] = bj;
            next[p] = -1;
            if (tail[bi] >= 0) next[tail[bi]] = p; else head[bi] = p;
            tail[bi] = p;
        }
        bval[(size_t)p * 16 + 4 * (irow[k] % 4) + (jcol[k] % 4)] = val[k];
    }
    if (indcol) {
        int out = 0;
        ptrow[0] = 0;
        for (int b = 0; b < nb; b++) {
            for (int p = head[b]; p >= 0; p = next[p]) {
                indcol[out] = bcol[p];
                memcpy(coef + (size_t)out * 16, bval + (size_t)p * 16, 16 * sizeof(double));
                out++;
            }
            ptrow[b + 1] = out;
        }
    }
    free(head); free(tail); free(next); free(bcol); free(bval);
    return nblk;
}

/* MatrixMarket reader with the reference's quirks, mpk/SpM2V.cpp:815-852 /
 * mpk/2SpMV.cpp:55-92: line 1 skipped unconditionally, then lines starting
 * with '%', then "nrow ncol nnz"; each entry is read with "%d %d %f" into a
 * FLOAT and widened, so every coefficient is rounded to binary32.
 * Two-pass: call with irow == NULL to read the header only. */
int orc_read_mtx(const char *path, int *nrow, int *nnz, int *irow, int *jcol, double *val)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    char buf[1024];
    int a = 0, b = 0, c = 0;
    if (!fgets(buf, sizeof buf, fp)) { fclose(fp); return -2; }
    for (;;) {
        if (!fgets(buf, sizeof buf, fp)) { fclose(fp); return -2; }
        if (buf[0] != '%') { sscanf(buf, "%d %d %d", &a, &b, &c); break; }
    }
    *nrow = a; *nnz = c;
    if (irow) {
        for (int k = 0; k < c; k++) {
            int i, j; float v;
            if (fscanf(fp, "%d %d %f", &i, &j, &v) != 3) { fclose(fp); return -3; }
            irow[k] = i - 1; jcol[k] = j - 1; val[k] = (double)v;
        }
    }
    fclose(fp);
    return 0;
}

/* Cache eviction sweep of the reference's timing protocol, mpk/utils.cpp:146-154
 * (300 MiB written then read).  Used only by bench.py's cpu_baseline leg. */
void orc_flush_cache(void)
{
    static unsigned char *buf = NULL;
    const size_t sz = (size_t)300 * 1024 * 1024;
    if (!buf) buf = (unsigned char *)malloc(sz);
    volatile unsigned char sink = 0;
    for (size_t i = 0; i < sz; i++) { buf[i] = (unsigned char)i; sink ^= buf[i]; }
    (void)sink;
}

/* Timed single-thread SpMV for the cpu_baseline leg: `reps` cold calls
 * (flush before each, as mpk/SpM2V.cpp:895-904), best time in seconds. */
#include <time.h>
double orc_time_spmv(int n, const int *ptrow, const int *indcol, const double *coef,
                     const double *x, double *y, int reps, int flush)
{
    double best = 1e300;
    for (int r = 0; r < reps; r++) {
        if (flush) orc_flush_cache();
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC_RAW, &t0);
        orc_spmv_csr_fma(n, ptrow, indcol, coef, x, y);
        clock_gettime(CLOCK_MONOTONIC_RAW, &t1);
        double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (dt < best) best = dt;
    }
    return best;
}

/* y = A x for row-major 4x4 blocks with PER-BLOCK partial sums: for every block the four products of a
 * row are chained from zero (p = B[4r]*x0; p = fma(B[4r+c], xc, p)) and the block's partial is then
 * ADDED to the running row value.  This is the arithmetic of the fused BCSR kernels SpM2V_BCSR_OPT /
 * _FMA (mpk/SpM2V.cpp:473-525, :566-618: "sum += B[4*r+c]*xk[c]" per block, then "yj[r] += sum") and of
 * the s-step multi-vector product MatMatMult_SeqBAIJ_4_AVX2 (src/kernels/spmm_avx2.c:77-88: acc per
 * block, then sum += acc) — NOT that of SpMV_BCSR_FMA, whose row is one continuous chain. */
void orc_spmv_bcsr4_blockacc(int nbrows, const int *ptrow, const int *indcol, const double *coef,
                             const double *x, double *y)
{
    for (int bi = 0; bi < nbrows; bi++) {
        double acc[4] = {0, 0, 0, 0};
        for (int ia = ptrow[bi]; ia < ptrow[bi + 1]; ia++) {
            const double *blk = coef + 16 * (size_t)ia;
            const double *xb = x + 4 * (size_t)indcol[ia];
            for (int i = 0; i < 4; i++) {
                double p = 0.0;
                for (int j = 0; j < 4; j++) p = fma(blk[4 * i + j], xb[j], p);
                volatile double t = acc[i] + p;
                acc[i] = t;
            }
        }
        for (int i = 0; i < 4; i++) y[4 * bi + i] = acc[i];
    }
}


/* Row-parallel form of orc_spmv_csr_fma for the bench's all-cores CPU column (SURVEY.md §8d "CPU baseline
 * beside it" (2)): rows are independent, each still ONE sequential fma chain, so the bits are those of the
 * single-thread function.  The reference itself is single-threaded (mpk/Makefile:10 has no -fopenmp);
 * this is the fair upper bound for its algorithm on the host's cores, not something the reference ships. */
#include <omp.h>
void orc_spmv_csr_fma_omp(int n, const int *ptrow, const int *indcol, const double *coef,
                          const double *x, double *y, int nthreads)
{
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int ia = ptrow[i]; ia < ptrow[i + 1]; ia++) s = fma(coef[ia], x[indcol[ia]], s);
        y[i] = s;
    }
}

/* best-of-reps seconds, warm (repeated products, as a solver would run them) */
double orc_time_spmv_omp(int n, const int *ptrow, const int *indcol, const double *coef,
                         const double *x, double *y, int reps, int nthreads)
{
    double best = 1e300;
    orc_spmv_csr_fma_omp(n, ptrow, indcol, coef, x, y, nthreads); /* first touch / page-in */
    for (int r = 0; r < reps; r++) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC_RAW, &t0);
        orc_spmv_csr_fma_omp(n, ptrow, indcol, coef, x, y, nthreads);
        clock_gettime(CLOCK_MONOTONIC_RAW, &t1);
        double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (dt < best) best = dt;
    }
    return best;
}
