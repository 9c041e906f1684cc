// mpk_shim.cpp — bodies of include/SpMV.h: the reference's mpk/ interface
// (aantoine890/navierstokes mpk/SpMV.h:37-66 and the per-file kernels) expressed
// over the C-ABI of include/mi355_spmv.h.  Host-only C++ (g++), links
// libmi355spmv.so; exports the same mangled symbols as mpk/SpMV.cpp +
// mpk/utils.cpp so the reference's drivers link against it unchanged
// (INTEGRATION.md).  Compute always goes to the GPU; the only host-side work is
// the COO->CSR/BCSR format building, which is setup-time integer work.
#include "SpMV.h"

#include <cstdint>
#include <cstdlib>
#include <map>

#include "mi355_spmv.h"

namespace {

[[noreturn]] void die(const char* where, int status)
{
    std::fprintf(stderr, "libmpk_mi355: %s failed: %s (%s)\n", where, mi_strerror(status), mi_last_error());
    std::abort();
}

#define MI_CALL(expr)                     \
    do {                                  \
        int st_ = (expr);                 \
        if (st_ != MI_OK) die(#expr, st_); \
    } while (0)

// Cheap content fingerprint so that a matrix rebuilt in the same storage is
// re-uploaded: size, extent and up to 64 evenly spaced (column, value) samples.
uint64_t fingerprint(int n, const int* ptrow, const int* indcol, const double* coef, size_t per_entry)
{
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; };
    const long long m = n > 0 ? ptrow[n] : 0;
    mix((uint64_t)n);
    mix((uint64_t)m);
    const long long step = m > 64 ? m / 64 : 1;
    for (long long k = 0; k < m; k += step) {
        mix((uint64_t)indcol[k]);
        uint64_t bits;
        std::memcpy(&bits, coef + (size_t)k * per_entry, sizeof bits);
        mix(bits);
    }
    return h;
}

struct Key {
    const void* p0;
    const void* p1;
    const void* p2;
    bool operator<(const Key& o) const
    {
        if (p0 != o.p0) return p0 < o.p0;
        if (p1 != o.p1) return p1 < o.p1;
        return p2 < o.p2;
    }
};

template <class H>
struct Slot {
    H handle;
    uint64_t fp;
};

std::map<Key, Slot<mi_csr_t>> g_csr;
std::map<Key, Slot<mi_bcsr4_t>> g_bcsr;

mi_csr_t device_csr(csrmatrix& A)
{
    const Key k{A.ptrow.data(), A.indcol.data(), A.coef.data()};
    const uint64_t fp = fingerprint(A.n, A.ptrow.data(), A.indcol.data(), A.coef.data(), 1);
    auto it = g_csr.find(k);
    if (it != g_csr.end() && it->second.fp == fp) return it->second.handle;
    if (it != g_csr.end()) {
        mi_csr_destroy(it->second.handle);
        g_csr.erase(it);
    }
    if (g_csr.size() >= 16) { // bounded: drop everything rather than grow without limit
        for (auto& kv : g_csr) mi_csr_destroy(kv.second.handle);
        g_csr.clear();
    }
    mi_csr_t h = nullptr;
    MI_CALL(mi_csr_create(A.n, A.n, A.ptrow.data(), A.indcol.data(), A.coef.data(), &h));
    g_csr[k] = Slot<mi_csr_t>{h, fp};
    return h;
}

mi_bcsr4_t device_bcsr(const bcsr4x4_matrix& A)
{
    const Key k{A.ptrow.data(), A.indcol.data(), A.coef.data()};
    const uint64_t fp = fingerprint(A.nrows, A.ptrow.data(), A.indcol.data(), A.coef.data(), 16);
    auto it = g_bcsr.find(k);
    if (it != g_bcsr.end() && it->second.fp == fp) return it->second.handle;
    if (it != g_bcsr.end()) {
        mi_bcsr4_destroy(it->second.handle);
        g_bcsr.erase(it);
    }
    if (g_bcsr.size() >= 16) {
        for (auto& kv : g_bcsr) mi_bcsr4_destroy(kv.second.handle);
        g_bcsr.clear();
    }
    // x is indexed by block column; the reference's callers pass vectors of
    // 4*nrows entries, so that is how much of x is transferred.
    mi_bcsr4_t h = nullptr;
    MI_CALL(mi_bcsr4_create(A.nrows, A.nrows, A.ptrow.data(), A.indcol.data(), A.coef.data(), &h));
    g_bcsr[k] = Slot<mi_bcsr4_t>{h, fp};
    return h;
}

void powers(int k, double* const* outs, double* x, csrmatrix& A)
{
    MI_CALL(mi_spmk(device_csr(A), k, x, outs));
}

} // namespace

// ---- y = A x ------------------------------------------------------------------

void SpMV_CSR(double* y, double* x, csrmatrix& A) { MI_CALL(mi_spmv(device_csr(A), x, y)); }
void SpMV_CSR_OPT(double* y, double* x, csrmatrix& A) { SpMV_CSR(y, x, A); }
void SpMV_CSR_FMA(double* y, double* x, csrmatrix& A) { SpMV_CSR(y, x, A); }
void SpMV_CSR_AVX2(double* y, double* x, csrmatrix& A) { SpMV_CSR(y, x, A); }

void SpMV_BCSR(double* y, const double* x, const bcsr4x4_matrix& A) { MI_CALL(mi_bcsr4_spmv(device_bcsr(A), x, y)); }
void SpMV_BCSR_OPT(double* y, const double* x, const bcsr4x4_matrix& A) { SpMV_BCSR(y, x, A); }
void SpMV_BCSR_FMA(double* y, const double* x, const bcsr4x4_matrix& A) { SpMV_BCSR(y, x, A); }
void SpMV_BCSR_AVX2(double* y, const double* x, const bcsr4x4_matrix& A) { SpMV_BCSR(y, x, A); }

// ---- matrix powers ---------------------------------------------------------------

void Generate1stlayer(std::vector<int>& ptrowend1, csrmatrix& A)
{
    // entry ia=(i,j): whole row j on the first meeting of column j, empty range afterwards
    std::vector<char> met((size_t)(A.n > 0 ? A.n : 0), 0);
    const int stored = A.n > 0 ? A.ptrow[A.n] : 0;
    ptrowend1.assign((size_t)std::max(A.nnz, stored), 0);
    for (int ia = 0; ia < stored; ia++) {
        const int j = A.indcol[ia];
        ptrowend1[ia] = met[j] ? A.ptrow[j] : A.ptrow[j + 1];
        met[j] = 1;
    }
}

void SpM2V_CSR(double* z, double* y, double* x, csrmatrix& A, std::vector<int>&)
{
    double* outs[2] = {y, z};
    powers(2, outs, x, A);
}
void SpM2V_CSR_OPT(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& t) { SpM2V_CSR(z, y, x, A, t); }
void SpM2V_CSR_FMA(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& t) { SpM2V_CSR(z, y, x, A, t); }
void SpM2V_CSR_AVX2(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& t) { SpM2V_CSR(z, y, x, A, t); }

// first-touch table of block rows, filled as the reference fills it (mpk/SpM2V.cpp:28-46): the
// block row of a block column's first appearance gets its full range, later appearances an empty one
void Generate1stlayer_BCSR4(std::vector<int>& ptrowendB, const bcsr4x4_matrix& A)
{
    std::vector<char> seen((size_t)A.nrows, 0);
    ptrowendB.resize(A.indcol.size());
    for (int bi = 0; bi < A.nrows; bi++)
        for (int m = A.ptrow[bi]; m < A.ptrow[bi + 1]; m++) {
            const int bj = A.indcol[m];
            ptrowendB[m] = seen[bj] ? A.ptrow[bj] : A.ptrow[bj + 1];
            seen[bj] = 1;
        }
}

void SpM2V_BCSR(double* z, double* y, double* x, bcsr4x4_matrix& A, std::vector<int>&)
{
    double* outs[2] = {y, z};
    MI_CALL(mi_bcsr4_spmk(device_bcsr(A), 2, x, outs));
}
void SpM2V_BCSR_OPT(double* z, double* y, double* x, bcsr4x4_matrix& A, std::vector<int>& t) { SpM2V_BCSR(z, y, x, A, t); }
void SpM2V_BCSR_FMA(double* z, double* y, double* x, bcsr4x4_matrix& A, std::vector<int>& t) { SpM2V_BCSR(z, y, x, A, t); }
void SpM2V_BCSR_AVX2(double* z, double* y, double* x, bcsr4x4_matrix& A, std::vector<int>& t) { SpM2V_BCSR(z, y, x, A, t); }

void SpM3V(double* w, double* z, double* y, double* x, csrmatrix& A, std::vector<int>&,
           std::vector<std::vector<int> >&)
{
    double* outs[3] = {y, z, w};
    powers(3, outs, x, A);
}

void SpM4V(double* v, double* w, double* z, double* y, double* x, csrmatrix& A, std::vector<int>&,
           std::vector<std::vector<int> >&, std::vector<std::vector<std::vector<int> > >&)
{
    double* outs[4] = {y, z, w, v};
    powers(4, outs, x, A);
}

// ---- BLAS-1 ----------------------------------------------------------------------

void orthogonalize(int nrow, const std::vector<double>& b, const std::vector<double>& x1,
                   std::vector<double>& x3, double alpha)
{
    double beta = 0.0;
    MI_CALL(mi_orthogonalize(nrow, b.data(), x1.data(), x3.data(), alpha, &beta));
}

void orthogonalize(int nrow, const std::vector<double>& x, std::vector<double>& y, double alpha)
{
    double beta = 0.0;
    std::vector<double> out((size_t)nrow);
    MI_CALL(mi_orthogonalize(nrow, x.data(), y.data(), out.data(), alpha, &beta));
    std::copy(out.begin(), out.end(), y.begin());
}

double norm2(const std::vector<double>& x)
{
    double r = 0.0;
    MI_CALL(mi_norm2((int)x.size(), x.data(), &r));
    return r;
}

double rel_error(const std::vector<double>& ref, const std::vector<double>& test)
{
    double r = 0.0;
    MI_CALL(mi_rel_error((int)ref.size(), ref.data(), test.data(), &r));
    return r;
}

void flush_cache() { MI_CALL(mi_flush_cache()); }

// ---- format builders ---------------------------------------------------------------

namespace {
struct Ent {
    int col;
    int seq;
    double v;
};
} // namespace

void generate_CSR(std::list<int>* ind_cols_tmp, std::list<double>* val_tmp, int nrow, int nnz, int* irow,
                  int* jcol, double* val)
{
    // bucket by row, order each bucket by (column, arrival), keep the first of equal columns
    std::vector<std::vector<Ent> > rows((size_t)nrow);
    for (int k = 0; k < nnz; k++) rows[irow[k]].push_back(Ent{jcol[k], k, val[k]});
    for (int i = 0; i < nrow; i++) {
        std::vector<Ent>& r = rows[i];
        std::sort(r.begin(), r.end(), [](const Ent& a, const Ent& b) { return a.col != b.col ? a.col < b.col : a.seq < b.seq; });
        // entries already present in the caller's lists take part too (the reference appends to them)
        for (size_t t = 0; t < r.size(); t++) {
            if (t > 0 && r[t].col == r[t - 1].col) continue;
            std::list<int>::iterator ic = ind_cols_tmp[i].begin();
            std::list<double>::iterator iv = val_tmp[i].begin();
            while (ic != ind_cols_tmp[i].end() && *ic < r[t].col) { ++ic; ++iv; }
            if (ic != ind_cols_tmp[i].end() && *ic == r[t].col) continue;
            ind_cols_tmp[i].insert(ic, r[t].col);
            val_tmp[i].insert(iv, r[t].v);
        }
    }
}

void COO2CSR(csrmatrix& a, int nrow, int nnz, int* irow, int* jcol, double* val)
{
    a.n = nrow;
    a.nnz = nnz; // the COO count, even when duplicates are dropped (mpk/utils.cpp:100)
    a.ptrow.assign((size_t)nrow + 1, 0);
    a.indcol.assign((size_t)nnz, 0);
    a.coef.assign((size_t)nnz, 0.0);
    std::vector<std::list<int> > cols((size_t)nrow);
    std::vector<std::list<double> > vals((size_t)nrow);
    generate_CSR(cols.data(), vals.data(), nrow, nnz, irow, jcol, val);
    int k = 0;
    for (int i = 0; i < nrow; i++) {
        std::list<double>::const_iterator iv = vals[i].begin();
        for (std::list<int>::const_iterator ic = cols[i].begin(); ic != cols[i].end(); ++ic, ++iv) {
            a.indcol[k] = *ic;
            a.coef[k] = *iv;
            k++;
        }
        a.ptrow[i + 1] = k;
    }
}

void generate_BCSR4(std::list<std::pair<int, std::array<double, 16> > >* block_rows, int nrow, int nnz,
                    const int* irow, const int* jcol, const double* val, bcsr4x4_matrix& A)
{
    typedef std::list<std::pair<int, std::array<double, 16> > > BlockList;
    const int nbr = nrow / 4;
    // per block row: block column -> position in the appearance-ordered list
    std::vector<std::map<int, BlockList::iterator> > where((size_t)(nrow + 3) / 4 + 1);
    for (int k = 0; k < nnz; k++) {
        const int bi = irow[k] / 4, bj = jcol[k] / 4;
        BlockList& L = block_rows[bi];
        std::map<int, BlockList::iterator>& W = where[bi];
        if (W.empty() && !L.empty()) // caller-provided content: index it once
            for (BlockList::iterator it = L.begin(); it != L.end(); ++it) W.insert(std::make_pair(it->first, it));
        std::map<int, BlockList::iterator>::iterator f = W.find(bj);
        if (f == W.end()) {
            std::array<double, 16> zero = {};
            L.push_back(std::make_pair(bj, zero));
            BlockList::iterator last = L.end();
            --last;
            f = W.insert(std::make_pair(bj, last)).first;
        }
        f->second->second[4 * (irow[k] % 4) + (jcol[k] % 4)] = val[k]; // last duplicate wins
    }
    A.nrows = nbr;
    A.ptrow.assign((size_t)nbr + 1, 0);
    A.indcol.clear();
    A.coef.clear();
    for (int bi = 0; bi < nbr; bi++) {
        for (BlockList::const_iterator it = block_rows[bi].begin(); it != block_rows[bi].end(); ++it) {
            A.indcol.push_back(it->first);
            A.coef.insert(A.coef.end(), it->second.begin(), it->second.end());
        }
        A.ptrow[bi + 1] = (int)A.indcol.size();
    }
    A.nblocks = 0; // the reference resets it and never sets it again (mpk/utils.cpp:78)
}
