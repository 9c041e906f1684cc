/*
 * fe_matrix.c — PETSc-free generator of the reference's Navier–Stokes FE matrices
 * (SURVEY.md §8 f-3): stabilised P1–P1 on tetrahedra, 4 dofs per node
 * (ux, uy, uz, p), assembled in 4x4 node blocks.
 *
 * What it reproduces
 *   - the element matrices of src/integration.c: lumped-form mass M (:91-109),
 *     symmetric-gradient diffusion A0 (:112-164), divergence B (:212-221) and the
 *     pressure stabilisation D (:224-238);
 *   - the block rule of assemble_ns_matrix, src/benchmark_spmv.c:104-118 — for the
 *     node pair (i, j) of an element the 4x4 block
 *         [ A0+M (3x3)        B[j][3i+a] (column 3) ]
 *         [ -B[i][3j+b] (row 3)            D[i][j]  ]
 *     is ADDED into block (node_i, node_j), every entry of a touched block being
 *     stored (PETSc BAIJ -> AIJ keeps explicit zeros), which is what makes the
 *     reference's mat/matrixN_aij.mtx rows 44-58 long and a multiple of 4.
 * The mesh the reference reads (gmsh .msh, missing: SURVEY.md F1) is replaced by a
 * structured box split into 6 Kuhn tetrahedra per cell: 15 block columns for an
 * interior node -> 60 nonzeros per row, inside the reference's range.
 *
 * The formulas are derived here, not transcribed: with g_i = grad(phi_i) and V the
 * tet volume,  A0[3i+a][3j+b] = (V/Re) (delta_ab g_i.g_j + g_i[b] g_j[a])  is the
 * closed form of the reference's 6-component D(u):D(v) loop; results agree with the
 * reference's object code to rounding (tests/test_fe_matrix.py).
 *
 * Plain C (gcc), part of libsynthcsr.so.  Setup-time code, not on the hot path.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* signed volume of the tet (a0,a1,a2,a3) and gradients of its four P1 hat functions */
static double tet_geometry(const double a[4][3], double g[4][3])
{
    double e[3][3]; /* edge vectors from vertex 0 */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) e[r][c] = a[r + 1][c] - a[0][c];
    /* cofactors: rows of inverse(E)^T up to 1/det */
    double cof[3][3];
    cof[0][0] = e[1][1] * e[2][2] - e[1][2] * e[2][1];
    cof[0][1] = e[1][2] * e[2][0] - e[1][0] * e[2][2];
    cof[0][2] = e[1][0] * e[2][1] - e[1][1] * e[2][0];
    cof[1][0] = e[2][1] * e[0][2] - e[2][2] * e[0][1];
    cof[1][1] = e[2][2] * e[0][0] - e[2][0] * e[0][2];
    cof[1][2] = e[2][0] * e[0][1] - e[2][1] * e[0][0];
    cof[2][0] = e[0][1] * e[1][2] - e[0][2] * e[1][1];
    cof[2][1] = e[0][2] * e[1][0] - e[0][0] * e[1][2];
    cof[2][2] = e[0][0] * e[1][1] - e[0][1] * e[1][0];
    const double det = e[0][0] * cof[0][0] + e[0][1] * cof[0][1] + e[0][2] * cof[0][2];
    /* phi_{r+1}(x) = row r of inverse(E) applied to (x - a0); phi_0 = 1 - sum */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) g[r + 1][c] = cof[r][c] / det;
    for (int c = 0; c < 3; c++) g[0][c] = -(g[1][c] + g[2][c] + g[3][c]);
    return det / 6.0;
}

static double tet_longest_edge(const double a[4][3])
{
    double m = 0.0;
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++) {
            double d2 = 0.0;
            for (int c = 0; c < 3; c++) d2 += (a[i][c] - a[j][c]) * (a[i][c] - a[j][c]);
            if (d2 > m) m = d2;
        }
    return sqrt(m);
}

/*
 * The sixteen 4x4 node blocks of one element, blk[i][j][16] row-major:
 *   rows/cols 0..2 = velocity components, 3 = pressure.
 */
void fe_element_blocks(const double a[4][3], double Re, double delta, double blk[4][4][16])
{
    double g[4][3];
    const double vol = tet_geometry(a, g);
    const double h = tet_longest_edge(a);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double* b = blk[i][j];
            const double gg = g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2];
            const double mass = (i == j) ? vol / 10.0 : vol / 20.0;
            for (int p = 0; p < 3; p++)
                for (int q = 0; q < 3; q++)
                    b[4 * p + q] = (vol / Re) * ((p == q ? gg : 0.0) + g[i][q] * g[j][p]) + (p == q ? mass : 0.0);
            /* divergence: B[L][3m+c] = (vol/4) G_m[c] (src/integration.c:212-221), where G is what the
               reference's tet_gradients returns: for a positively oriented tet that is MINUS the true
               gradient (its face normals point outward, :39-57; unit tet: G_0 = (+1,+1,+1)).  A0 and D
               are quadratic in G and do not notice; B carries the sign, and it is reproduced here so
               that the assembled matrix equals the reference's. */
            for (int p = 0; p < 3; p++) b[4 * p + 3] = -(vol / 4.0) * g[i][p];  /*  B[j][3i+p] */
            for (int q = 0; q < 3; q++) b[12 + q] = (vol / 4.0) * g[j][q];      /* -B[i][3j+q] */
            b[15] = delta * h * h * vol * gg;
        }
}

/* node offsets a node of the Kuhn triangulation is connected to (besides itself) */
static const int kNbr[15][3] = {
    {0, 0, 0},   {1, 0, 0},  {-1, 0, 0},  {0, 1, 0},   {0, -1, 0}, {0, 0, 1},  {0, 0, -1}, {1, 1, 0},
    {-1, -1, 0}, {1, 0, 1},  {-1, 0, -1}, {0, 1, 1},   {0, -1, -1}, {1, 1, 1},  {-1, -1, -1},
};

static int nbr_slot(int dx, int dy, int dz)
{
    for (int s = 0; s < 15; s++)
        if (kNbr[s][0] == dx && kNbr[s][1] == dy && kNbr[s][2] == dz) return s;
    return -1;
}

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* node coordinates: unit spacing, each interior node displaced by a seeded jitter < jitter/2 per axis */
static void node_xyz(int ix, int iy, int iz, int nx, int ny, int nz, double jitter, uint64_t seed, double out[3])
{
    out[0] = ix; out[1] = iy; out[2] = iz;
    if (jitter > 0.0 && ix > 0 && ix < nx && iy > 0 && iy < ny && iz > 0 && iz < nz) {
        uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(ix + (nx + 1) * (iy + (ny + 1) * iz)));
        for (int c = 0; c < 3; c++) {
            h = mix64(h + 0x9E3779B97F4A7C15ull);
            out[c] += jitter * ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5);
        }
    }
}

/* dimension helpers: rows = 4 * nodes; nnz counted exactly by fe_matrix_count */
long long fe_matrix_rows(int nx, int ny, int nz) { return 4ll * (nx + 1) * (ny + 1) * (nz + 1); }

long long fe_matrix_count(int nx, int ny, int nz)
{
    long long blocks = 0;
    for (int iz = 0; iz <= nz; iz++)
        for (int iy = 0; iy <= ny; iy++)
            for (int ix = 0; ix <= nx; ix++)
                for (int s = 0; s < 15; s++) {
                    const int jx = ix + kNbr[s][0], jy = iy + kNbr[s][1], jz = iz + kNbr[s][2];
                    if (jx >= 0 && jx <= nx && jy >= 0 && jy <= ny && jz >= 0 && jz <= nz) blocks++;
                }
    return blocks * 16;
}

/*
 * Assemble the matrix of the (nx x ny x nz)-cell box.  Node id = ix + (nx+1)(iy + (ny+1) iz),
 * dof = 4*node + component.  Output CSR with ascending columns; ptrow has rows+1 entries.
 * Returns 0, -1 on bad arguments, -2 on allocation failure.
 */
static int fe_assemble_impl(int nx, int ny, int nz, double Re, double delta, double jitter, unsigned long long seed,
                            int* ptrow, int* indcol, double* coef, int scalar)
{
    if (nx < 1 || ny < 1 || nz < 1 || Re <= 0.0 || jitter < 0.0 || jitter > 0.4) return -1;
    const long long nn = (long long)(nx + 1) * (ny + 1) * (nz + 1);
    if (nn * (scalar ? 1 : 4) > 0x7fffffffll) return -1;
    const int W = scalar ? 1 : 16; /* scalar: only the pressure-pressure entry [3][3] of every node block */
    double* acc = (double*)calloc((size_t)nn * 15 * W, sizeof(double)); /* [node][slot][W] */
    if (!acc) return -2;
    /* Kuhn split of the unit cube along the (1,1,1) diagonal: one tet per permutation of the axes */
    static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    for (int cz = 0; cz < nz; cz++)
        for (int cy = 0; cy < ny; cy++)
            for (int cx = 0; cx < nx; cx++)
                for (int t = 0; t < 6; t++) {
                    int v[4][3];
                    v[0][0] = cx; v[0][1] = cy; v[0][2] = cz;
                    for (int s = 0; s < 3; s++) {
                        v[s + 1][0] = v[s][0]; v[s + 1][1] = v[s][1]; v[s + 1][2] = v[s][2];
                        v[s + 1][perm[t][s]] += 1;
                    }
                    double a[4][3];
                    for (int m = 0; m < 4; m++) node_xyz(v[m][0], v[m][1], v[m][2], nx, ny, nz, jitter, seed, a[m]);
                    /* orientation: swap two vertices of odd permutations so that the volume is positive
                       (the reference warns on vol <= 0, src/integration.c:34-36) */
                    double g[4][3];
                    if (tet_geometry(a, g) < 0.0) {
                        for (int c = 0; c < 3; c++) {
                            const int ti = v[2][c]; v[2][c] = v[3][c]; v[3][c] = ti;
                            const double td = a[2][c]; a[2][c] = a[3][c]; a[3][c] = td;
                        }
                    }
                    double blk[4][4][16];
                    fe_element_blocks(a, Re, delta, blk);
                    for (int i = 0; i < 4; i++) {
                        const long long ni = v[i][0] + (long long)(nx + 1) * (v[i][1] + (long long)(ny + 1) * v[i][2]);
                        for (int j = 0; j < 4; j++) {
                            const int s = nbr_slot(v[j][0] - v[i][0], v[j][1] - v[i][1], v[j][2] - v[i][2]);
                            if (s < 0) { free(acc); return -1; }
                            double* d = acc + ((size_t)ni * 15 + s) * W;
                            if (scalar) d[0] += blk[i][j][15];
                            else
                                for (int q = 0; q < 16; q++) d[q] += blk[i][j][q];
                        }
                    }
                }
    /* slots in ascending order of the neighbour's node id */
    int order[15];
    long long delta_id[15];
    for (int s = 0; s < 15; s++) {
        order[s] = s;
        delta_id[s] = kNbr[s][0] + (long long)(nx + 1) * (kNbr[s][1] + (long long)(ny + 1) * kNbr[s][2]);
    }
    for (int p = 1; p < 15; p++)
        for (int q = p; q > 0 && delta_id[order[q]] < delta_id[order[q - 1]]; q--) {
            const int t = order[q]; order[q] = order[q - 1]; order[q - 1] = t;
        }
    long long pos = 0;
    ptrow[0] = 0;
    for (int iz = 0; iz <= nz; iz++)
        for (int iy = 0; iy <= ny; iy++)
            for (int ix = 0; ix <= nx; ix++) {
                const long long ni = ix + (long long)(nx + 1) * (iy + (long long)(ny + 1) * iz);
                if (scalar) {
                    for (int k = 0; k < 15; k++) {
                        const int s = order[k];
                        const int jx = ix + kNbr[s][0], jy = iy + kNbr[s][1], jz = iz + kNbr[s][2];
                        if (jx < 0 || jx > nx || jy < 0 || jy > ny || jz < 0 || jz > nz) continue;
                        indcol[pos] = (int)(ni + delta_id[s]);
                        coef[pos] = acc[(size_t)ni * 15 + s];
                        pos++;
                    }
                    ptrow[ni + 1] = (int)pos;
                    continue;
                }
                for (int r = 0; r < 4; r++) {
                    for (int k = 0; k < 15; k++) {
                        const int s = order[k];
                        const int jx = ix + kNbr[s][0], jy = iy + kNbr[s][1], jz = iz + kNbr[s][2];
                        if (jx < 0 || jx > nx || jy < 0 || jy > ny || jz < 0 || jz > nz) continue;
                        const long long nj = ni + delta_id[s];
                        const double* d = acc + ((size_t)ni * 15 + s) * 16 + 4 * r;
                        for (int c = 0; c < 4; c++) {
                            indcol[pos] = (int)(4 * nj + c);
                            coef[pos] = d[c];
                            pos++;
                        }
                    }
                    ptrow[4 * ni + r + 1] = (int)pos;
                }
            }
    free(acc);
    return 0;
}

int fe_matrix_assemble(int nx, int ny, int nz, double Re, double delta, double jitter, unsigned long long seed,
                       int* ptrow, int* indcol, double* coef)
{
    return fe_assemble_impl(nx, ny, nz, Re, delta, jitter, seed, ptrow, indcol, coef, 0);
}

/*
 * The pressure-pressure part of the same matrix: one row per node, entry (i, j) = block (i, j)[3][3] — the stabilisation
 * term delta h^2 (grad phi_i, grad phi_j) of src/integration.c, i.e. a P1 Laplacian on the jittered Kuhn mesh, the shape of
 * the reference's pressure Poisson operator: 15 nonzeros per interior row (7-15 on the boundary), ascending columns.
 * rows = fe_matrix_rows / 4, nonzeros = fe_matrix_count / 16.
 */
int fe_pressure_matrix_assemble(int nx, int ny, int nz, double Re, double delta, double jitter, unsigned long long seed,
                                int* ptrow, int* indcol, double* coef)
{
    return fe_assemble_impl(nx, ny, nz, Re, delta, jitter, seed, ptrow, indcol, coef, 1);
}
