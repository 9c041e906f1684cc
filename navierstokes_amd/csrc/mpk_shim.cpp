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
#include <thread>

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

// Content hashes of the caller's arrays.  The reference's functions read the live arrays on every
// call; this library computes from a device copy, so before every product the shim must know whether
// the caller changed anything since the copy was made — including in-place edits of a few
// coefficients (boundary-condition rows, the Newton loop's Jacobian update,
// src/solve_newton.c:1245-1247).  EVERY byte is hashed (a sampled fingerprint would miss exactly
// those edits): 8 independent multiply-xor lanes per 4 MiB chunk, chunks hashed by a few threads and
// combined in order, so the value does not depend on the thread count.  Measured: 6.4 ms for the 920 MB of a
// 75 M-nonzero matrix (144 GB/s, 16 threads), a whole call 8.0 ms against 190 ms+ for the reference's CPU product of that size; callers that keep
// their matrix fixed can switch the check off (mi355_assume_unchanged).
uint64_t hash_chunk(const unsigned char* p, size_t bytes)
{
    static const uint64_t K[8] = {0x9e3779b97f4a7c15ull, 0xbf58476d1ce4e5b9ull, 0x94d049bb133111ebull, 0xd6e8feb86659fd93ull,
                                  0xca5a826395121157ull, 0x9fb21c651e98df25ull, 0xa0761d6478bd642full, 0xe7037ed1a0b428dbull};
    uint64_t lane[8];
    for (int i = 0; i < 8; i++) lane[i] = K[i] ^ (uint64_t)bytes;
    size_t k = 0;
    for (; k + 64 <= bytes; k += 64) {
        uint64_t w[8];
        std::memcpy(w, p + k, 64);
        for (int i = 0; i < 8; i++) lane[i] = (lane[i] ^ w[i]) * K[i] + (lane[i] >> 29);
    }
    uint64_t tail[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bytes > k) std::memcpy(tail, p + k, bytes - k);
    uint64_t h = 0xcbf29ce484222325ull;
    for (int i = 0; i < 8; i++) {
        lane[i] = (lane[i] ^ tail[i]) * K[i] + (lane[i] >> 29);
        h = (h ^ lane[i]) * 0x100000001b3ull + (h >> 31);
    }
    return h;
}

uint64_t hash_bytes(const void* data, size_t bytes)
{
    const size_t chunk = (size_t)4 << 20;
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    std::vector<uint64_t> part(nchunks ? nchunks : 1, 0);
    auto work = [&](size_t c0, size_t c1) {
        for (size_t c = c0; c < c1; c++) part[c] = hash_chunk(p + c * chunk, std::min(chunk, bytes - c * chunk));
    };
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt > 16 ? 16 : (nt < 1 ? 1 : nt);
    if (nchunks < 4 || nt == 1) work(0, nchunks);
    else {
        if (nt > nchunks) nt = (unsigned)nchunks;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, nchunks * t / nt, nchunks * (t + 1) / nt);
        for (std::thread& t : th) t.join();
    }
    uint64_t h = 0x84222325cbf29ce4ull ^ (uint64_t)bytes;
    for (size_t c = 0; c < nchunks; c++) h = (h ^ part[c]) * 0x100000001b3ull + (h >> 31);
    return h;
}

bool g_assume_unchanged = false; // mi355_assume_unchanged(true): the caller promises to call mi355_invalidate after edits

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
    int n;
    long long stored;     // ptrow[n] when the copy was made
    uint64_t pattern_fp;  // ptrow + indcol
    uint64_t values_fp;   // coef
};

std::map<Key, Slot<mi_csr_t>> g_csr;
std::map<Key, Slot<mi_bcsr4_t>> g_bcsr;
std::map<Key, Slot<mi_dist_t>> g_dist;

// MI355_NGPUS=N (N >= 2): the CSR entry points of this header — SpMV_CSR*, SpM2V_CSR*, SpM3V, SpM4V*, orthogonalize — run on
// a mi_dist handle (include/mi355_spmv.h: the matrix row-partitioned over N GPUs of this ONE process) instead of a one-GPU
// handle; same bits.  Unset / 1: one GPU.  (The blocked entry points stay on one GPU.)
int ngpus()
{
    static const int n = [] {
        const char* e = std::getenv("MI355_NGPUS");
        const int v = e ? std::atoi(e) : 0;
        return v >= 2 && v <= 64 ? v : 0;
    }();
    return n;
}
mi_dist_t g_last_dist = nullptr; // the handle orthogonalize(nrow, ...) distributes its vectors like (it takes no matrix)
int g_last_dist_n = -1;

// the device copy of A, brought up to date with the caller's arrays
template <class H, class Create, class Update, class Destroy>
H device_copy(std::map<Key, Slot<H>>& cache, const Key& k, int n, const int* ptrow, const int* indcol, const double* coef,
              size_t per_entry, Create create, Update update, Destroy destroy)
{
    const long long stored = n > 0 ? ptrow[n] : 0;
    typename std::map<Key, Slot<H>>::iterator it = cache.find(k);
    const bool cached = it != cache.end() && it->second.n == n && it->second.stored == stored;
    if (cached && g_assume_unchanged) return it->second.handle;
    // (an empty matrix may come with empty vectors: nothing to read then)
    const uint64_t pfp = hash_bytes(ptrow, n > 0 ? sizeof(int) * ((size_t)n + 1) : 0) ^ (hash_bytes(indcol, sizeof(int) * (size_t)stored) * 3);
    const uint64_t vfp = hash_bytes(coef, sizeof(double) * per_entry * (size_t)stored);
    if (cached && it->second.pattern_fp == pfp) {
        if (it->second.values_fp != vfp) { // same pattern, new coefficients: refresh the values only
            update(it->second.handle);
            it->second.values_fp = vfp;
        }
        return it->second.handle;
    }
    if (it != cache.end()) {
        destroy(it->second.handle);
        cache.erase(it);
    }
    if (cache.size() >= 16) { // bounded: drop everything rather than grow without limit
        for (typename std::map<Key, Slot<H>>::iterator q = cache.begin(); q != cache.end(); ++q) destroy(q->second.handle);
        cache.clear();
    }
    H h = create();
    cache[k] = Slot<H>{h, n, stored, pfp, vfp};
    return h;
}

mi_csr_t device_csr(csrmatrix& A)
{
    const Key k{A.ptrow.data(), A.indcol.data(), A.coef.data()};
    return device_copy<mi_csr_t>(
        g_csr, k, A.n, A.ptrow.data(), A.indcol.data(), A.coef.data(), 1,
        [&]() {
            mi_csr_t h = nullptr;
            MI_CALL(mi_csr_create(A.n, A.n, A.ptrow.data(), A.indcol.data(), A.coef.data(), &h));
            return h;
        },
        [&](mi_csr_t h) { MI_CALL(mi_csr_update_values(h, A.coef.data())); }, [](mi_csr_t h) { mi_csr_destroy(h); });
}

mi_dist_t device_dist(csrmatrix& A)
{
    const Key k{A.ptrow.data(), A.indcol.data(), A.coef.data()};
    mi_dist_t h = device_copy<mi_dist_t>(
        g_dist, k, A.n, A.ptrow.data(), A.indcol.data(), A.coef.data(), 1,
        [&]() {
            mi_dist_t d = nullptr;
            MI_CALL(mi_dist_create(ngpus(), A.n, A.ptrow.data(), A.indcol.data(), A.coef.data(), &d));
            return d;
        },
        [&](mi_dist_t d) { MI_CALL(mi_dist_update_values(d, A.coef.data())); },
        [](mi_dist_t d) {
            if (d == g_last_dist) g_last_dist = nullptr;
            mi_dist_destroy(d);
        });
    g_last_dist = h;
    g_last_dist_n = A.n;
    return h;
}

mi_bcsr4_t device_bcsr(const bcsr4x4_matrix& A)
{
    const Key k{A.ptrow.data(), A.indcol.data(), A.coef.data()};
    // x is indexed by block column: the reference reads x[4*bj .. 4*bj+3] for every block column bj that
    // occurs (mpk/SpMV.cpp:104-113), which for a matrix whose row count is not a multiple of 4 reaches one
    // block past 4*nrows (generate_BCSR4 truncates the ROWS only, mpk/utils.cpp:49).  The same range of the
    // caller's x is transferred here: 4 * (largest block column + 1) entries, at least 4*nrows.
    return device_copy<mi_bcsr4_t>(
        g_bcsr, k, A.nrows, A.ptrow.data(), A.indcol.data(), A.coef.data(), 16,
        [&]() {
            int nbcols = A.nrows;
            const int nb = A.nrows > 0 ? A.ptrow[A.nrows] : 0;
            for (int m = 0; m < nb; m++) nbcols = std::max(nbcols, A.indcol[m] + 1);
            mi_bcsr4_t h = nullptr;
            MI_CALL(mi_bcsr4_create(A.nrows, nbcols, A.ptrow.data(), A.indcol.data(), A.coef.data(), &h));
            return h;
        },
        [&](mi_bcsr4_t h) { MI_CALL(mi_bcsr4_update_values(h, A.coef.data())); }, [](mi_bcsr4_t h) { mi_bcsr4_destroy(h); });
}

void powers(int k, double* const* outs, double* x, csrmatrix& A)
{
    if (ngpus()) MI_CALL(mi_dist_spmk(device_dist(A), k, x, outs));
    else MI_CALL(mi_spmk(device_csr(A), k, x, outs));
}

} // namespace

// ---- device-copy cache control (extensions; see include/SpMV.h) ---------------------------------

void mi355_assume_unchanged(bool on) { g_assume_unchanged = on; }

void mi355_invalidate(csrmatrix& A)
{
    std::map<Key, Slot<mi_csr_t> >::iterator it = g_csr.find(Key{A.ptrow.data(), A.indcol.data(), A.coef.data()});
    if (it != g_csr.end()) {
        mi_csr_destroy(it->second.handle);
        g_csr.erase(it);
    }
    std::map<Key, Slot<mi_dist_t> >::iterator jt = g_dist.find(Key{A.ptrow.data(), A.indcol.data(), A.coef.data()});
    if (jt != g_dist.end()) {
        if (jt->second.handle == g_last_dist) g_last_dist = nullptr;
        mi_dist_destroy(jt->second.handle);
        g_dist.erase(jt);
    }
}

void mi355_invalidate(const bcsr4x4_matrix& A)
{
    std::map<Key, Slot<mi_bcsr4_t> >::iterator it = g_bcsr.find(Key{A.ptrow.data(), A.indcol.data(), A.coef.data()});
    if (it != g_bcsr.end()) {
        mi_bcsr4_destroy(it->second.handle);
        g_bcsr.erase(it);
    }
}

// ---- y = A x ------------------------------------------------------------------

void SpMV_CSR(double* y, double* x, csrmatrix& A)
{
    if (ngpus()) MI_CALL(mi_dist_spmv(device_dist(A), x, y));
    else MI_CALL(mi_spmv(device_csr(A), x, y));
}
void SpMV_CSR_OPT(double* y, double* x, csrmatrix& A) { SpMV_CSR(y, x, A); }
void SpMV_CSR_FMA(double* y, double* x, csrmatrix& A) { SpMV_CSR(y, x, A); }
void SpMV_CSR_AVX2(double* y, double* x, csrmatrix& A) { SpMV_CSR(y, x, A); }
void SpMV(double* y, double* x, csrmatrix& A) { SpMV_CSR(y, x, A); } // the name in mpk/SpMVmulti0.cpp:223-236

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
    ptrowend1.resize((size_t)std::max(A.nnz, stored)); // resize(A.nnz) in the reference: entries behind ptrow[n] keep their content
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

// Nested first-touch tables of the k = 3, 4 CPU traversals (mpk/SpMVmulti0.cpp:106-130, :157-187).
// The GPU kernels do not use them; they are filled exactly as the reference fills them for callers
// that build, print or reuse them.  Level L has its own "already met" set, updated in the order in
// which the level-L loop of the traversal reaches an index.
void Generate2ndlayer(std::vector<std::vector<int> >& ptrowend2, csrmatrix& A, std::vector<int>& ptrowend1)
{
    std::vector<char> met((size_t)(A.n > 0 ? A.n : 0), 0);
    const int stored = A.n > 0 ? A.ptrow[A.n] : 0;
    ptrowend2.resize((size_t)std::max(A.nnz, stored));
    for (int ia = 0; ia < stored; ia++) {
        const int first = A.ptrow[A.indcol[ia]], last = ptrowend1[ia];
        std::vector<int>& t = ptrowend2[ia];
        t.resize((size_t)(last - first));
        for (int jb = first; jb < last; jb++) {
            const int k = A.indcol[jb];
            t[jb - first] = met[k] ? A.ptrow[k] : A.ptrow[k + 1];
            met[k] = 1;
        }
    }
}

void Generate3rdlayer(std::vector<std::vector<std::vector<int> > >& ptrowend3, csrmatrix& A, std::vector<int>& ptrowend1,
                      std::vector<std::vector<int> >& ptrowend2)
{
    std::vector<char> met((size_t)(A.n > 0 ? A.n : 0), 0);
    const int stored = A.n > 0 ? A.ptrow[A.n] : 0;
    ptrowend3.resize((size_t)std::max(A.nnz, stored));
    for (int ia = 0; ia < stored; ia++) {
        const int first = A.ptrow[A.indcol[ia]], last = ptrowend1[ia];
        std::vector<std::vector<int> >& t2 = ptrowend3[ia];
        t2.resize((size_t)(last - first));
        for (int jb = first; jb < last; jb++) {
            const int k = A.indcol[jb];
            const int kfirst = A.ptrow[k], klast = ptrowend2[ia][jb - first];
            if (klast <= kfirst) continue; // the reference sizes this level inside its (empty) kc loop
            std::vector<int>& t3 = t2[jb - first];
            t3.resize((size_t)(klast - kfirst));
            for (int kc = kfirst; kc < klast; kc++) {
                const int l = A.indcol[kc];
                t3[kc - kfirst] = met[l] ? A.ptrow[l] : A.ptrow[l + 1];
                met[l] = 1;
            }
        }
    }
}

// mpk/SpMVmulti0.cpp:44-61 (SpM2V0) and :65-104 (SpM2V): the k = 2 kernels under their other names
void SpM2V0(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& t) { SpM2V_CSR(z, y, x, A, t); }
void SpM2V(double* z, double* y, double* x, csrmatrix& A, std::vector<int>& t) { SpM2V_CSR(z, y, x, A, t); }

// mpk/SpMVmulti-1.cpp:434-493: outputs first (y4 = A^4 x ... y1 = A x), const inputs
void SpM4V_AVX2(double* y4, double* y3, double* y2, double* y1, const double* x, const csrmatrix& A, const std::vector<int>&,
                const std::vector<std::vector<int> >&, const std::vector<std::vector<std::vector<int> > >&)
{
    double* outs[4] = {y1, y2, y3, y4};
    powers(4, outs, const_cast<double*>(x), const_cast<csrmatrix&>(A));
}

// ---- BLAS-1 ----------------------------------------------------------------------

void orthogonalize(int nrow, const std::vector<double>& b, const std::vector<double>& x1,
                   std::vector<double>& x3, double alpha)
{
    double beta = 0.0;
    // N GPUs: the vectors are distributed like the rows of the matrix last multiplied with (the call names no matrix)
    if (ngpus() && g_last_dist && g_last_dist_n == nrow) MI_CALL(mi_dist_orthogonalize(g_last_dist, b.data(), x1.data(), x3.data(), alpha, &beta));
    else MI_CALL(mi_orthogonalize(nrow, b.data(), x1.data(), x3.data(), alpha, &beta));
}

void orthogonalize(int nrow, const std::vector<double>& x, std::vector<double>& y, double alpha)
{
    double beta = 0.0;
    if (ngpus() && g_last_dist && g_last_dist_n == nrow) MI_CALL(mi_dist_orthogonalize(g_last_dist, x.data(), y.data(), y.data(), alpha, &beta));
    else MI_CALL(mi_orthogonalize(nrow, x.data(), y.data(), y.data(), alpha, &beta)); // in place: x3 == x1
}

void orthonormalize_against_basis(int nrow, std::vector<std::vector<double> >& basis, std::vector<double>& y)
{
    std::vector<const double*> ptrs(basis.size());
    for (size_t j = 0; j < basis.size(); j++) ptrs[j] = basis[j].data();
    MI_CALL(mi_orthonormalize_against_basis(nrow, (int)basis.size(), ptrs.data(), y.data(), nullptr));
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
