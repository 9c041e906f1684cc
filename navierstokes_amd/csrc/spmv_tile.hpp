// spmv_tile.hpp — the "tile" fp64 CSR SpMV kernel for gfx950 (MI355X): the kernel of wide-band matrices whose rows
// share columns — P1 operators on unstructured 3-D meshes after the create-time relabelling (reorder.hpp), the
// reference's scalar workload (pressure Poisson, src/solve_newton.c) — which neither the ring kernel (the column span
// of a row block, ~n^(2/3), outgrows any contiguous LDS window) nor the stream kernel (one L1 gather per nonzero)
// serves well.  Plan and rationale: tile_plan.hpp.
//
// One 256-thread workgroup per row block of <= NNZB nonzeros:
//   A. every global load of the block is issued up front, oldest first where it is needed first: the block's list
//      of DISTINCT columns (4 B each, coalesced), then the 16-bit slot stream (one 16-byte load per thread), then the
//      values (8 B, tid-strided, non-temporal for matrices beyond the Infinity Cache); as soon as the list has arrived
//      — the slot and value loads still in flight behind it — x is gathered ONCE per distinct column (ascending
//      columns: neighbouring lanes share lines);
//   B. values, slots and the x tile are parked in LDS;
//   C. one thread per row walks its segment: s = fma(coef[k], tile[slot[k]], s) in CSR order — the same sequential
//      chain as every other kernel here, bit-equal to the reference's SpMV_CSR_OPT/_FMA (mpk/SpMV.cpp:23-56).
// Per nonzero 8 + 2 + 4u bytes of matrix (u = distinct columns per nonzero of the block, 0.25-0.35 on meshes) instead
// of 12, and u instead of 1 global gathers.  XCD-aware block order as in the stream kernel, so that the x lines one
// block gathers are L2 hits for its neighbours.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmv_kernels.hpp"
#include "tile_plan.hpp"

namespace mi355 {

// row chain over {s_c[k], s_xs[s_j[k]]}, operands fetched U at a time (cf. row_chain, spmv_kernels.hpp)
template <int U, bool SKEW>
__device__ __forceinline__ double tile_row_chain(const double* s_c, const unsigned short* s_j, const double* s_xs, int ra, int re)
{
    double s = 0.0;
    for (int k0 = ra; k0 < re; k0 += U) {
        double cc[U], xx[U];
        unsigned jj[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = min(k0 + u, re - 1);
            jj[u] = s_j[k];
            cc[u] = s_c[SKEW ? sk(k) : k];
        }
#pragma unroll
        for (int u = 0; u < U; u++) xx[u] = s_xs[jj[u]];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (k0 + u < re) s = fma(cc[u], xx[u], s);
    }
    return s;
}

// One listed block, R = ceil(U / T) rounds of distinct columns (uniform per workgroup; one straight-line instantiation
// per R, so that every wait is counted: the list loads are the oldest, waiting for them leaves slots and values in flight).
template <int NNZB, bool NT, bool SKEW, int R>
__device__ __forceinline__ void tile_block(const CsrView& A, const int4 d0, const int4 d1, const unsigned* __restrict__ ulist,
                                           const unsigned short* __restrict__ slots, const double* __restrict__ x,
                                           double* __restrict__ y, double* s_c, double* s_xs, unsigned short* s_j)
{
    constexpr int T = kTileThreads, PER = NNZB / T;
    const int tid = threadIdx.x;
    const int r0 = d0.x, p0 = d0.y, u0 = d0.z, s0 = d0.w, r1 = d1.x;
    const int U = d1.z - u0;
    // ---- A: the distinct-column list first (the x gather waits for it alone) ...
    unsigned uid[R];
    const unsigned* ul = ulist + u0;
    const int ulast = U - 1;
#pragma unroll
    for (int i = 0; i < R; i++) uid[i] = ul[min(tid + i * T, ulast)];
    // ... then slots, row extents and values
    const uint4 sv = *reinterpret_cast<const uint4*>(slots + (size_t)s0 + 8 * tid);
    const int rowc = min(r0 + tid, r1 - 1);
    const int pa = A.ptrow[rowc], pe = A.ptrow[rowc + 1];
    double c[PER];
    const double* cb = A.coef + p0 + tid; // unclamped: lanes past the block's end read what lies behind (padded array), never used
#pragma unroll
    for (int i = 0; i < PER; i++) {
        if (NT) c[i] = __builtin_nontemporal_load(&cb[i * T]);
        else c[i] = cb[i * T];
    }
    __builtin_amdgcn_sched_barrier(0);
    double xv[R];
#pragma unroll
    for (int i = 0; i < R; i++) xv[i] = x[uid[i]];
    __builtin_amdgcn_sched_barrier(0);
    // ---- B: park the x tile, values and slots
#pragma unroll
    for (int i = 0; i < R; i++)
        if (tid + i * T < U) s_xs[tid + i * T] = xv[i];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int k = tid + i * T;
        s_c[SKEW ? sk(k) : k] = c[i];
    }
    *reinterpret_cast<uint4*>(s_j + 8 * tid) = sv;
    __syncthreads();
    // ---- C: row chains
    if (r0 + tid < r1) y[A.rowmap ? A.rowmap[r0 + tid] : r0 + tid] = tile_row_chain<8, SKEW>(s_c, s_j, s_xs, pa - p0, pe - p0);
    for (int r = r0 + tid + T; r < r1; r += T) { // blocks of very short rows hold more than T rows
        const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
        y[A.rowmap ? A.rowmap[r] : r] = tile_row_chain<8, SKEW>(s_c, s_j, s_xs, a, e);
    }
}

template <int NNZB, bool NT, bool SKEW>
__global__ __launch_bounds__(kTileThreads) void spmv_csr_tile(CsrView A, const int4* __restrict__ desc, int nblk,
                                                             const unsigned* __restrict__ ulist,
                                                             const unsigned short* __restrict__ slots,
                                                             const double* __restrict__ x, double* __restrict__ y)
{
    constexpr int T = kTileThreads;
    static_assert(NNZB / T == 8, "a thread's slots are one 16-byte load");
    constexpr int LDSN = SKEW ? NNZB + NNZB / 32 + 1 : NNZB;
    __shared__ double s_c[LDSN];
    __shared__ double s_xs[NNZB];
    __shared__ __attribute__((aligned(16))) unsigned short s_j[NNZB];

    const int b = xcd_remap(blockIdx.x, nblk);
    if (b >= nblk) return;
    const int tid = threadIdx.x;
    const int4 d0 = desc[b], d1 = desc[b + 1];
    const int r0 = d0.x, p0 = d0.y, r1 = d1.x;
    const int nn = d1.y - p0, U = d1.z - d0.z;

    if (nn == 0) { // a block of empty rows
        for (int r = r0 + tid; r < r1; r += T) y[A.rowmap ? A.rowmap[r] : r] = 0.0;
        return;
    }
    if (nn > NNZB) { // one row longer than a block: chunk by chunk, the chain carried by thread 0 (as in the stream kernel)
        double s = 0.0;
        for (int base = p0; base < p0 + nn; base += NNZB) {
            const int m = min(NNZB, p0 + nn - base);
            for (int k = tid; k < m; k += T) {
                s_c[k] = A.coef[base + k];
                s_xs[k] = x[A.indcol[base + k]];
            }
            __syncthreads();
            if (tid == 0)
                for (int k = 0; k < m; k++) s = fma(s_c[k], s_xs[k], s);
            __syncthreads();
        }
        if (tid == 0) y[A.rowmap ? A.rowmap[r0] : r0] = s;
        return;
    }
    switch ((U + T - 1) / T) {
    case 1: tile_block<NNZB, NT, SKEW, 1>(A, d0, d1, ulist, slots, x, y, s_c, s_xs, s_j); break;
    case 2: tile_block<NNZB, NT, SKEW, 2>(A, d0, d1, ulist, slots, x, y, s_c, s_xs, s_j); break;
    case 3: tile_block<NNZB, NT, SKEW, 3>(A, d0, d1, ulist, slots, x, y, s_c, s_xs, s_j); break;
    case 4: tile_block<NNZB, NT, SKEW, 4>(A, d0, d1, ulist, slots, x, y, s_c, s_xs, s_j); break;
    case 5: tile_block<NNZB, NT, SKEW, 5>(A, d0, d1, ulist, slots, x, y, s_c, s_xs, s_j); break;
    case 6: tile_block<NNZB, NT, SKEW, 6>(A, d0, d1, ulist, slots, x, y, s_c, s_xs, s_j); break;
    case 7: tile_block<NNZB, NT, SKEW, 7>(A, d0, d1, ulist, slots, x, y, s_c, s_xs, s_j); break;
    default: tile_block<NNZB, NT, SKEW, 8>(A, d0, d1, ulist, slots, x, y, s_c, s_xs, s_j); break;
    }
}

} // namespace mi355
