/*
 * synth_csr.c — seeded synthetic CSR generators for the SpMV hot path.
 *
 * The reference ships no matrix (mat/*.mtx and mmesh.tar.gz are
 * missing blobs), so every workload is generated.  The generators are
 * counter-based: row i depends only on (kind, seed, n, w, i), so any rank can
 * generate exactly its own row range and all ranks agree on the global matrix.
 *
 * Kinds (SURVEY.md §8d "Synthetic inputs"):
 *   S15  (kind 0): exactly 15 distinct columns per row = diagonal + 14 draws
 *                  i + U[-w, w]; ascending; diag 1.0, off-diag U(-1,1)/15.
 *   SVAR (kind 1): row length U{8..22} (mean 15), same band; off-diag U(-1,1)/len.
 *   SFE  (kind 2): 4x4-block FE-like rows (mpk matrices have 44-58 nnz/row, all
 *                  row lengths = 0 mod 4, mpk/log/log_SPMV.txt:1,82): 14 block
 *                  columns per block row (diag block + 13 in bi +- w/4), dense
 *                  4x4 blocks; 56 nnz/row.  n must be a multiple of 4.
 *
 * Column ordering inside a row is ascending, as mpk/utils.cpp:5-43
 * (generate_CSR) guarantees for the reference's csrmatrix.
 *
 * Plain C, no dependencies; built by gcc into libsynthcsr.so.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SYNTH_S15 0
#define SYNTH_SVAR 1
#define SYNTH_SFE 2

static inline uint64_t sm64_next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline double sm64_unit(uint64_t *s) /* U[0,1) with 53 bits */
{
    return (double)(sm64_next(s) >> 11) * (1.0 / 9007199254740992.0);
}

static inline uint64_t row_seed(uint64_t seed, uint64_t salt, int64_t i)
{
    uint64_t s = seed * 1000003ull + (uint64_t)i + salt * 0xD1B54A32D192ED03ull;
    /* one scrambling step so that neighbouring rows start far apart */
    sm64_next(&s);
    return s;
}

/* insertion of c into the ascending array cols[0..m); returns 0 if present */
static inline int sorted_insert(int *cols, int m, int c)
{
    int lo = 0;
    while (lo < m && cols[lo] < c) lo++;
    if (lo < m && cols[lo] == c) return 0;
    for (int k = m; k > lo; k--) cols[k] = cols[k - 1];
    cols[lo] = c;
    return 1;
}

static int row_len_of(int kind, uint64_t seed, int n, int64_t i)
{
    int len;
    if (kind == SYNTH_S15) len = 15;
    else if (kind == SYNTH_SVAR) {
        uint64_t s = row_seed(seed, 7, i);
        len = 8 + (int)(sm64_next(&s) % 15ull);
    } else len = 56;
    if (kind != SYNTH_SFE && len > n) len = n;
    return len;
}

/* number of nonzeros in rows [rb, re) */
long long synth_count(int kind, unsigned long long seed, int n, int w, long long rb, long long re)
{
    (void)w;
    if (kind == SYNTH_S15) return (long long)row_len_of(kind, seed, n, 0) * (re - rb);
    if (kind == SYNTH_SFE) {
        int nb = n / 4;
        int bl = nb < 14 ? nb : 14;
        return (long long)bl * 4 * (re - rb);
    }
    long long t = 0;
    for (long long i = rb; i < re; i++) t += row_len_of(kind, seed, n, i);
    return t;
}

static void gen_scalar_row(int kind, uint64_t seed, int n, int w, int64_t i, int len, int *cols, double *vals)
{
    uint64_t s = row_seed(seed, 1, i);
    int m = 0;
    cols[m++] = (int)i;
    uint64_t span = 2ull * (uint64_t)w + 1ull;
    int tries = 0;
    while (m < len) {
        int64_t c = i + (int64_t)(sm64_next(&s) % span) - (int64_t)w;
        if (++tries > 4096) { /* degenerate tiny n: fall back to a linear fill */
            for (int cc = 0; cc < n && m < len; cc++) m += sorted_insert(cols, m, cc);
            break;
        }
        if (c < 0 || c >= n) continue;
        m += sorted_insert(cols, m, (int)c);
    }
    for (int k = 0; k < len; k++) {
        double u = sm64_unit(&s);
        vals[k] = (cols[k] == (int)i) ? 1.0 : (2.0 * u - 1.0) / (double)len;
    }
}

/*
 * Generate rows [rb, re).  ptrow has (re-rb)+1 entries, RELATIVE to the first
 * generated nonzero (ptrow[0] == 0).  indcol/coef must hold synth_count()
 * entries.  Column indices are GLOBAL.  Returns 0, or -1 on bad arguments.
 */
int synth_rows(int kind, unsigned long long seed, int n, int w, long long rb, long long re,
               int *ptrow, int *indcol, double *coef)
{
    if (n <= 0 || rb < 0 || re > n || rb > re || w < 1) return -1;
    if (kind == SYNTH_SFE) {
        if (n % 4) return -1;
        int nb = n / 4, wb = w / 4 < 1 ? 1 : w / 4;
        int bl = nb < 14 ? nb : 14;
        long long pos = 0, out_r = 0;
        ptrow[0] = 0;
        for (long long bi = rb / 4; bi <= (re - 1) / 4 && rb < re; bi++) {
            uint64_t s = row_seed(seed, 3, bi);
            int bcols[16];
            int m = 0, tries = 0;
            bcols[m++] = (int)bi;
            uint64_t span = 2ull * (uint64_t)wb + 1ull;
            while (m < bl) {
                int64_t c = bi + (int64_t)(sm64_next(&s) % span) - (int64_t)wb;
                if (++tries > 4096) {
                    for (int cc = 0; cc < nb && m < bl; cc++) m += sorted_insert(bcols, m, cc);
                    break;
                }
                if (c < 0 || c >= nb) continue;
                m += sorted_insert(bcols, m, (int)c);
            }
            for (int r = 0; r < 4; r++) {
                long long row = bi * 4 + r;
                int emit = (row >= rb && row < re);
                for (int k = 0; k < bl; k++)
                    for (int c = 0; c < 4; c++) {
                        double u = sm64_unit(&s);
                        if (!emit) continue;
                        int col = bcols[k] * 4 + c;
                        indcol[pos] = col;
                        coef[pos] = (col == (int)row) ? 1.0 : (2.0 * u - 1.0) / 56.0;
                        pos++;
                    }
                if (emit) ptrow[++out_r] = (int)pos;
            }
        }
        return 0;
    }
    if (kind != SYNTH_S15 && kind != SYNTH_SVAR) return -1;
    long long pos = 0;
    ptrow[0] = 0;
    int cols[32];
    double vals[32];
    for (long long i = rb; i < re; i++) {
        int len = row_len_of(kind, seed, n, i);
        gen_scalar_row(kind, seed, n, w, i, len, cols, vals);
        memcpy(indcol + pos, cols, (size_t)len * sizeof(int));
        memcpy(coef + pos, vals, (size_t)len * sizeof(double));
        pos += len;
        ptrow[i - rb + 1] = (int)pos;
    }
    return 0;
}

/* x_j = sin(0.001 j) (mpk/2SpMV.cpp:114) written without libm dependence on
 * the caller side; j is the GLOBAL index. */
#include <math.h>
void synth_x_sin(long long jb, long long je, double *x)
{
    for (long long j = jb; j < je; j++) x[j - jb] = sin(0.001 * (double)j);
}

/* ---- unstructured node numbering ---------------------------------------------------------------------
 * The reference's matrices come from gmsh meshes (src/solve_newton.c:91-197 reads them, src/benchmark_spmv.c:
 * 76-123 assembles): node numbers follow the mesher, not geometry.  synth_node_permutation draws a seeded
 * random numbering of the nn nodes (Fisher-Yates on splitmix64) and synth_permute_sym_sorted applies it
 * symmetrically, B = P A P^T, delivering every row with ASCENDING columns — what MatView + COO2CSR
 * (mpk/utils.cpp:5-43) hand to the kernels for such a mesh.  block = 4 keeps the four dofs of a node
 * together (new row = 4 * perm[row / 4] + row % 4), so the 4x4 node-block structure survives. */
void synth_node_permutation(unsigned long long seed, int nn, int *perm)
{
    uint64_t s = row_seed(seed, 0x9e3779b9u, nn);
    for (int i = 0; i < nn; i++) perm[i] = i;
    for (int i = nn - 1; i > 0; i--) {
        const int j = (int)(sm64_next(&s) % (uint64_t)(i + 1));
        const int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
}

/* p2/c2/v2 sized like the input; returns 0, or -1 on bad arguments */
int synth_permute_sym_sorted(int n, int block, const int *p, const int *c, const double *v, const int *perm_nodes,
                             int *p2, int *c2, double *v2)
{
    if (n < 0 || block < 1 || n % block) return -1;
    const int nn = n / block;
    int *inv = (int *)malloc(sizeof(int) * (size_t)(nn > 0 ? nn : 1));
    if (!inv) return -1;
    for (int i = 0; i < nn; i++) inv[perm_nodes[i]] = i;
    p2[0] = 0;
    for (int rn = 0; rn < n; rn++) {
        const int ro = inv[rn / block] * block + rn % block;
        p2[rn + 1] = p2[rn] + (p[ro + 1] - p[ro]);
    }
    for (int rn = 0; rn < n; rn++) {
        const int ro = inv[rn / block] * block + rn % block;
        const int a = p[ro], len = p[ro + 1] - p[ro], b = p2[rn];
        for (int k = 0; k < len; k++) { /* insertion sort by new column: rows are short */
            const int cn = perm_nodes[c[a + k] / block] * block + c[a + k] % block;
            const double val = v[a + k];
            int m = k;
            while (m > 0 && c2[b + m - 1] > cn) { c2[b + m] = c2[b + m - 1]; v2[b + m] = v2[b + m - 1]; m--; }
            c2[b + m] = cn;
            v2[b + m] = val;
        }
    }
    free(inv);
    return 0;
}
