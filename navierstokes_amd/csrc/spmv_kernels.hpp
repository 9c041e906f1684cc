// spmv_kernels.hpp — fp64 CSR SpMV kernels for gfx950 (MI355X, CDNA4).
//
// The operation is y = A x for the reference's csrmatrix (mpk/SpMV.h:18-24),
// i.e. SpMV_CSR* of mpk/SpMV.cpp:6-85.  The path is HBM-bound (12 B of matrix
// per 2 flop), so everything here is about streaming coef/indcol with fully
// coalesced reads and keeping the x gather and the per-row reduction off the
// critical path.  No MFMA: there is no dense contraction in a 15-nnz row.
//
// Numerics contract (all kernels): each row is ONE sequential fma chain in CSR
// order — s = fma(coef[k], x[indcol[k]], s) — bit-identical to the reference's
// SpMV_CSR_OPT / SpMV_CSR_FMA object code (mpk/SpMV.cpp:23-56).  A tree
// reduction would be a few % cheaper in LDS traffic but would give up
// bit-parity with the reference; the row chain is hidden under the HBM stream.
//
// Kernel "stream" (the CSR-stream shape): a 256-thread workgroup owns a block
// of consecutive rows holding <= NNZB nonzeros (row-block table built once at
// mi_csr_create).  Phase 1: all threads stream the block's coef/indcol range
// with unit-stride loads (the range is contiguous in CSR), gather x, and park
// {coef, x[col]} in LDS.  Phase 2: one thread per row walks its LDS segment
// with the fma chain and stores y (coalesced, rows are consecutive).
// XCD-aware block order: blockIdx -> (xcd = b & 7, slot = b >> 3) -> row block
// xcd * per_xcd + slot, so each XCD (own 4 MiB L2) sweeps one contiguous 1/8 of
// the rows and the x window it gathers from stays in ITS L2 instead of being
// pulled into all eight.
//
// Kernel "stream_xlds": same, plus the x window [cmin, cmax] of the row block
// is first staged into LDS with coalesced 16-byte loads and the gather is served
// by ds_read_b64 instead of 64 divergent global requests per wave instruction.
// Used when the block's column span fits the LDS budget (banded / FE-ordered
// matrices); blocks whose span is too wide take the global gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

constexpr int kWG = 256;        // threads per workgroup (4 waves of 64)
constexpr int kNXCD = 8;        // XCDs on MI355X, each with a private L2

// Device view of a CSR matrix plus its row-block table.
struct CsrView {
    int n;                 // rows
    int ncols;             // length of x
    const int* ptrow;      // [n+1]
    const int* indcol;     // [nnz]
    const double* coef;    // [nnz]
    const int* rowmap;     // [n] or nullptr: row r writes y[rowmap[r]]
    const int2* blk;       // [nblk+1] {first row, first nnz}; blk[nblk] = {n, nnz}
    const int2* blk_span;  // [nblk] {min col, max col} of the block (stream_xlds)
    int nblk;
};

// LDS index skew: one extra slot every 32 entries, so that per-row walks with a
// power-of-two stride (FE rows are 16/32/48/56 long) do not land on one bank pair.
__device__ __forceinline__ int sk(int k) { return k + (k >> 5); }

__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    const int per = (nblk + kNXCD - 1) / kNXCD;
    return (bid & (kNXCD - 1)) * per + (bid >> 3);
}

// ---------------------------------------------------------------------------
// stream kernel.  NNZB: nonzeros per row block (LDS = 2 * 8 B * sk(NNZB)).
// XLDS: stage the block's x window in LDS when it fits XWIN doubles.
// ---------------------------------------------------------------------------
template <int NNZB, bool XLDS, int XWIN>
__global__ __launch_bounds__(kWG) void spmv_csr_stream(CsrView A, const double* __restrict__ x,
                                                       double* __restrict__ y)
{
    constexpr int PER = NNZB / kWG; // nonzeros per thread in phase 1
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_win[XLDS ? XWIN : 1];

    const int b = xcd_remap(blockIdx.x, A.nblk);
    if (b >= A.nblk) return;
    const int tid = threadIdx.x;
    const int2 d0 = A.blk[b];
    const int2 d1 = A.blk[b + 1];
    const int r0 = d0.x, p0 = d0.y, r1 = d1.x, p1 = d1.y;
    const int nn = p1 - p0;

    if (nn <= NNZB) {
        // row extents for phase 2, requested early so they are in flight with the stream
        int ra = 0, re = 0;
        const int myrow = r0 + tid;
        if (myrow < r1) {
            ra = A.ptrow[myrow] - p0;
            re = A.ptrow[myrow + 1] - p0;
        }
        double c[PER];
        int j[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = tid + i * kWG;
            if (k < nn) {
                c[i] = A.coef[p0 + k];
                j[i] = A.indcol[p0 + k];
            }
        }
        bool use_win = false;
        int cmin = 0;
        if (XLDS) {
            const int2 sp = A.blk_span[b];
            cmin = sp.x & ~1; // keep 16-byte alignment of the staged window
            const int wlen = sp.y - cmin + 1;
            use_win = (wlen <= XWIN);
            if (use_win) {
                // coalesced 16-B loads of x[cmin .. cmin+wlen)
                const double2* src = reinterpret_cast<const double2*>(x + cmin);
                const int n2 = wlen >> 1;
                for (int t = tid; t < n2; t += kWG) {
                    const double2 v = src[t];
                    s_win[2 * t] = v.x;
                    s_win[2 * t + 1] = v.y;
                }
                if ((wlen & 1) && tid == 0) s_win[wlen - 1] = x[cmin + wlen - 1];
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = tid + i * kWG;
            if (k < nn) {
                const double xv = (XLDS && use_win) ? s_win[j[i] - cmin] : x[j[i]];
                s_c[sk(k)] = c[i];
                s_x[sk(k)] = xv;
            }
        }
        __syncthreads();
        for (int r = myrow; r < r1; r += kWG) {
            if (r != myrow) {
                ra = A.ptrow[r] - p0;
                re = A.ptrow[r + 1] - p0;
            }
            double s = 0.0;
            for (int k = ra; k < re; k++) s = fma(s_c[sk(k)], s_x[sk(k)], s);
            y[A.rowmap ? A.rowmap[r] : r] = s;
        }
    } else {
        // one row longer than a block: stream it chunk by chunk; the chain itself
        // is inherently serial, thread 0 carries it (exactness over speed: FE rows
        // never get here, their length is bounded by the mesh valence).
        double s = 0.0;
        for (int base = p0; base < p1; base += NNZB) {
            const int m = min(NNZB, p1 - base);
            for (int k = tid; k < m; k += kWG) {
                s_c[sk(k)] = A.coef[base + k];
                s_x[sk(k)] = x[A.indcol[base + k]];
            }
            __syncthreads();
            if (tid == 0)
                for (int k = 0; k < m; k++) s = fma(s_c[sk(k)], s_x[sk(k)], s);
            __syncthreads();
        }
        if (tid == 0) y[A.rowmap ? A.rowmap[r0] : r0] = s;
    }
}

// ---------------------------------------------------------------------------
// rowpar kernel: one thread per row straight from global memory — the shape of
// the reference's CPU loop (mpk/SpMV.cpp:41-56).  Uncoalesced (lane stride =
// row length); kept as the simple always-valid baseline and for tiny matrices.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void spmv_csr_rowpar(CsrView A, const double* __restrict__ x,
                                                       double* __restrict__ y)
{
    const int r = blockIdx.x * kWG + threadIdx.x;
    if (r >= A.n) return;
    double s = 0.0;
    for (int k = A.ptrow[r]; k < A.ptrow[r + 1]; k++) s = fma(A.coef[k], x[A.indcol[k]], s);
    y[A.rowmap ? A.rowmap[r] : r] = s;
}

// ---------------------------------------------------------------------------
// BCSR 4x4, row-major blocks (mpk/SpMV.h:26-33, fill order mpk/utils.cpp:83-94).
// Four lanes per block row (lane q owns row 4*bi+q); per block, lane q reads its
// 4 coefficients as two 16-B loads (the 4 lanes together read the block's 128 B
// contiguously) and the 4 x values of the block column.  fma order = for block,
// for j — identical to SpMV_BCSR_FMA (mpk/SpMV.cpp:150-178).
// ---------------------------------------------------------------------------
struct Bcsr4View {
    int nbrows, nbcols;
    const int* ptrow;
    const int* indcol;
    const double* coef; // 16 per block
};

__global__ __launch_bounds__(kWG) void spmv_bcsr4(Bcsr4View A, const double* __restrict__ x,
                                                  double* __restrict__ y)
{
    const int g = blockIdx.x * kWG + threadIdx.x;
    const int bi = g >> 2, q = g & 3;
    if (bi >= A.nbrows) return;
    double s = 0.0;
    for (int ia = A.ptrow[bi]; ia < A.ptrow[bi + 1]; ia++) {
        const double2* row = reinterpret_cast<const double2*>(A.coef + 16 * (size_t)ia + 4 * q);
        const double2 a01 = row[0], a23 = row[1];
        const double2* xb = reinterpret_cast<const double2*>(x + 4 * (size_t)A.indcol[ia]);
        const double2 x01 = xb[0], x23 = xb[1];
        s = fma(a01.x, x01.x, s);
        s = fma(a01.y, x01.y, s);
        s = fma(a23.x, x23.x, s);
        s = fma(a23.y, x23.y, s);
    }
    y[4 * (size_t)bi + q] = s;
}

} // namespace mi355
