// spmm_tile.hpp — the multi-vector product Y[:, j] = A X[:, j], j < S, on the blocked matrix with each workgroup's x blocks
// gathered ONCE into LDS (MatMatMult_SeqBAIJ_4_AVX2, src/kernels/spmm_avx2.c:7-109; the tile idea of spmv_bcsr4_tile at S columns).
//
// Why: in spmm_bcsr4 / spmm_bcsr4_quad (spmv_kernels.hpp) every block costs its lanes a dependent chain — block column from
// memory, THEN the S x blocks through L1/L2 — and S x blocks of 32 bytes from S different arrays per block: 4.6 M blocks x S
// scattered 32-byte reads.  The kernel sat at 45-54 % of its byte model and fetched 1.31x the model's bytes (x once per XCD).
// Here a workgroup of 512 threads owns a tile of up to 128 block rows; the host lists the distinct block columns they touch (~320
// for the FE matrix) and gives every block the 16-bit position of its column in that list.  The workgroup gathers the S
// columns of those nodes into LDS once (node-major records of 4 S + 2 doubles: the two pad doubles spread neighbouring
// records over the banks), and the inner loop waits for COEFFICIENTS only — a pure stream, P blocks deep in registers — while
// x comes from LDS (2 S ds_read_b128 per block and lane, quads broadcast).
// The groups are CLUSTERS of the block graph (capi_bcsr.hip: build_spmm_tile_plan — breadth-first balls), not runs of consecutive
// rows: a ball of 128 nodes of a 3-D mesh touches ~2.5 distinct columns per row where 128 consecutive nodes touch 5.1, which
// halves the gather and the LDS footprint (three workgroups per CU at four columns).
// Measured (FE matrix, 1.31 M rows; profiles/NOTES.md §4.8, profiles/r03_spmm_tile_ablation.txt): four columns 168-179 -> 148-157 us in every
// tile form built; eight columns 227-237 -> 180-192 us with eight lanes per block row and non-temporal coefficient loads
// (spmm_bcsr4_otile below).  Compiling pieces out of this kernel shows where the rest goes: no gather 133 us, no Y stores 128-131,
// neither 114 (= the single-vector kernel's coefficient stream) — about 20 us each, additive, neither explained by bytes (PMC: the
// stores add no fetch traffic; the gathers 157 MB of 809): the stores are the memory-side read/write turnaround of profiles/NOTES.md §4.4, the
// gather two dependent round trips per workgroup.  Persistent workgroups with the next tile's x blocks staged in registers were
// built twice and measured slower (231 / 164-170 us at four columns: vmcnt counts in issue order, and the staging costs registers).
// Arithmetic: exactly spmm_bcsr4's (ARITH 0: one fma chain per row and column, bit-equal to SpMV_BCSR_FMA; ARITH 1: per-block
// partial sums added to the row value) — same blocks in the same order.
#pragma once
#include <hip/hip_runtime.h>

#include "spmv_kernels.hpp"

namespace mi355 {

constexpr int kSpmmTileThreads = 512;              // 128 block rows per workgroup
constexpr int kSpmmTileRows = kSpmmTileThreads / 4;

// The tile's x blocks into LDS: thread u takes node u of the tile's list in all S columns — one index load, then 2 S 16-byte loads
// in flight at once, then the LDS writes.  (Round 5: the earlier form — elements (column, node) dealt over all threads, four per
// thread, the writes under `if (e < total)` — compiled into four dependent index -> x -> write sequences per thread, each waiting
// for the one before (the loads had been sunk into the conditional writes), plus a software integer division per element.)
template <int S, int T>
__device__ __forceinline__ void spmm_gather_nodes(const unsigned* __restrict__ nodes, int U, const double* __restrict__ X, long long ldx, double* s_xt, int tid)
{
    constexpr int REC = 4 * S + 2;
    for (int u = tid; u < U; u += T) {
        const unsigned node = nodes[u];
        double2 v0[S], v1[S];
#pragma unroll
        for (int j = 0; j < S; j++) {
            const double2* xb = reinterpret_cast<const double2*>(X + (size_t)j * ldx + 4 * (size_t)node);
            v0[j] = xb[0];
            v1[j] = xb[1];
        }
        __builtin_amdgcn_sched_barrier(0); // every load is out before the first write waits for one
#pragma unroll
        for (int j = 0; j < S; j++) {
            double2* d = reinterpret_cast<double2*>(s_xt + (size_t)u * REC + 4 * j);
            d[0] = v0[j];
            d[1] = v1[j];
        }
    }
}

template <int S, int ARITH, int P>
__global__ __launch_bounds__(kSpmmTileThreads) void spmm_bcsr4_tile(Bcsr4View A, Bcsr4Tile Tl, const double* __restrict__ X, long long ldx,
                                                                     double* __restrict__ Y, long long ldy, int nwg, int xcd_chunk)
{
    extern __shared__ __attribute__((aligned(16))) double s_xt[];
    constexpr int REC = 4 * S + 2; // doubles per node record (16-byte aligned; the pad spreads records over the LDS banks)
    constexpr int T = kSpmmTileThreads;
    // xcd_chunk > 0: tiles in XCD-chunked order (spmv_kernels.hpp: xcd_remap_chunked) — neighbouring tiles, which share most of
    // their nodes, then run on ONE XCD and find each other's x blocks in its L2
    const int wg = xcd_chunk > 0 ? xcd_remap_chunked((int)blockIdx.x, nwg, xcd_chunk) : (int)blockIdx.x;
    if (wg >= nwg) return;
    const int tid = threadIdx.x;
    const int q = tid & 3;
    const int rr = Tl.rows[wg * kSpmmTileRows + (tid >> 2)]; // the tile's rows are a cluster of the block graph, not a range
    const bool live = rr >= 0;
    const int bi = live ? rr : -1 - rr;                       // (unused places shadow a row of the tile and store nothing)
    const int u0 = Tl.wg_ptr[wg], U = Tl.wg_ptr[wg + 1] - u0;
    const int ia0 = A.ptrow[bi], ia1 = A.ptrow[bi + 1];
    const int last = max(ia1 - 1, ia0);
    const double* cq = A.coef + 4 * q;
    double2 a01[P], a23[P];
    unsigned sl[P];
#pragma unroll
    for (int t = 0; t < P; t++) { // the first coefficient stages are in flight across the gather and its barrier
        const int blk = min(ia0 + t, last);
        const double2* row = reinterpret_cast<const double2*>(cq + 16 * (size_t)blk);
        a01[t] = row[0];
        a23[t] = row[1];
        sl[t] = Tl.slots[blk];
    }
    // the tile: element e = (column j, list position u), u fastest — neighbouring threads read neighbouring nodes of one
    // column (ascending, often adjacent in memory); four elements per thread in flight at a time
    spmm_gather_nodes<S, T>(Tl.nodes + u0, U, X, ldx, s_xt, tid);
    __syncthreads();
    double acc[S];
#pragma unroll
    for (int j = 0; j < S; j++) acc[j] = 0.0;
    for (int ia = ia0; ia < ia1; ia += P) {
#pragma unroll
        for (int t = 0; t < P; t++) {
            const double2 c01 = a01[t], c23 = a23[t];
            const double2* xs = reinterpret_cast<const double2*>(s_xt + (size_t)sl[t] * REC);
            const int nb = min(ia + t + P, last);
            const double2* nrow = reinterpret_cast<const double2*>(cq + 16 * (size_t)nb);
            a01[t] = nrow[0];
            a23[t] = nrow[1];
            sl[t] = Tl.slots[nb];
            if (ia + t < ia1) {
#pragma unroll
                for (int j = 0; j < S; j++) {
                    const double2 v01 = xs[2 * j], v23 = xs[2 * j + 1];
                    if (ARITH == 0) {
                        double s = acc[j];
                        s = fma(c01.x, v01.x, s);
                        s = fma(c01.y, v01.y, s);
                        s = fma(c23.x, v23.x, s);
                        s = fma(c23.y, v23.y, s);
                        acc[j] = s;
                    } else {
                        double p = fma(c01.x, v01.x, 0.0);
                        p = fma(c01.y, v01.y, p);
                        p = fma(c23.x, v23.x, p);
                        p = fma(c23.y, v23.y, p);
                        acc[j] = __dadd_rn(acc[j], p);
                    }
                }
            }
        }
    }
    if (live) {
        const size_t orow = 4 * (size_t)(A.browmap ? A.browmap[bi] : bi) + q;
#pragma unroll
        for (int j = 0; j < S; j++) Y[(size_t)j * ldy + orow] = acc[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// EIGHT lanes per block row (S even): lane l of a row's octet loads doubles [2l, 2l + 1] of every block — ONE 16-byte load per lane
// and block, the octet reading the block's 128 bytes as one contiguous line (the quad forms read it as two instructions of four
// strided 16-byte pieces each).  Lanes 2q and 2q + 1 hold the two halves of row q and swap them through DPP (quad_perm [1,0,3,2]),
// after which both hold the whole row; lane (q, h = l & 1) runs the fma chains of row q for the S / 2 columns of group h — so every
// (row, column) chain is still ONE lane's sequential chain over the row's blocks, bit for bit spmm_bcsr4's.  Half the load
// instructions per block, whole lines per instruction (which is what lets the coefficient stream be loaded non-temporally without
// being fetched twice, NT), half the LDS reads and accumulators per lane, and twice the waves per block row: 64 block rows per
// 512-thread tile.
// ---------------------------------------------------------------------------------------------------------------------------
template <int S, int ARITH, int P, bool NT>
__global__ __launch_bounds__(kSpmmTileThreads) void spmm_bcsr4_otile(Bcsr4View A, Bcsr4Tile Tl, const double* __restrict__ X, long long ldx,
                                                                      double* __restrict__ Y, long long ldy, int nwg)
{
    static_assert(S % 2 == 0, "two column groups");
    extern __shared__ __attribute__((aligned(16))) double s_xt[];
    typedef double dvec2 __attribute__((ext_vector_type(2)));
    constexpr int REC = 4 * S + 2, SL = S / 2, T = kSpmmTileThreads, ROWS = T / 8;
    const int wg = (int)blockIdx.x;
    if (wg >= nwg) return;
    const int tid = threadIdx.x;
    const int l = tid & 7, q = l >> 1, h = l & 1;
    const int rr = Tl.rows[wg * ROWS + (tid >> 3)];
    const bool live = rr >= 0;
    const int bi = live ? rr : -1 - rr; // (unused places shadow a row of the tile and store nothing)
    const int u0 = Tl.wg_ptr[wg], U = Tl.wg_ptr[wg + 1] - u0;
    const int ia0 = A.ptrow[bi], ia1 = A.ptrow[bi + 1];
    const int last = max(ia1 - 1, ia0);
    const double* cl = A.coef + 2 * l;
    dvec2 a[P];
    unsigned sl[P];
    auto ldc = [&](int blk) -> dvec2 {
        const dvec2* p = reinterpret_cast<const dvec2*>(cl + 16 * (size_t)blk);
        if (NT) return __builtin_nontemporal_load(p);
        return *p;
    };
#pragma unroll
    for (int t = 0; t < P; t++) {
        const int blk = min(ia0 + t, last);
        a[t] = ldc(blk);
        sl[t] = Tl.slots[blk];
    }
    spmm_gather_nodes<S, T>(Tl.nodes + u0, U, X, ldx, s_xt, tid);
    __syncthreads();
    double acc[SL];
#pragma unroll
    for (int j = 0; j < SL; j++) acc[j] = 0.0;
    for (int ia = ia0; ia < ia1; ia += P) {
#pragma unroll
        for (int t = 0; t < P; t++) {
            const dvec2 mine = a[t];
            const double2* xs = reinterpret_cast<const double2*>(s_xt + (size_t)sl[t] * REC + 4 * SL * h);
            const int nb = min(ia + t + P, last);
            a[t] = ldc(nb);
            sl[t] = Tl.slots[nb];
            // the partner lane (l ^ 1) holds the other half of this row: swap neighbours inside the quad
            constexpr int kSwap = 0xB1; // quad_perm [1, 0, 3, 2]
            const double ox = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(mine.x), kSwap, 0xf, 0xf, true),
                                               __builtin_amdgcn_mov_dpp(__double2loint(mine.x), kSwap, 0xf, 0xf, true));
            const double oy = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(mine.y), kSwap, 0xf, 0xf, true),
                                               __builtin_amdgcn_mov_dpp(__double2loint(mine.y), kSwap, 0xf, 0xf, true));
            const double c0 = h ? ox : mine.x, c1 = h ? oy : mine.y, c2 = h ? mine.x : ox, c3 = h ? mine.y : oy;
            if (ia + t < ia1) { // uniform within the octet
#pragma unroll
                for (int j = 0; j < SL; j++) {
                    const double2 v01 = xs[2 * j], v23 = xs[2 * j + 1];
                    if (ARITH == 0) {
                        double sacc = acc[j];
                        sacc = fma(c0, v01.x, sacc);
                        sacc = fma(c1, v01.y, sacc);
                        sacc = fma(c2, v23.x, sacc);
                        sacc = fma(c3, v23.y, sacc);
                        acc[j] = sacc;
                    } else {
                        double p = fma(c0, v01.x, 0.0);
                        p = fma(c1, v01.y, p);
                        p = fma(c2, v23.x, p);
                        p = fma(c3, v23.y, p);
                        acc[j] = __dadd_rn(acc[j], p);
                    }
                }
            }
        }
    }
    if (live) {
        const size_t orow = 4 * (size_t)(A.browmap ? A.browmap[bi] : bi) + q;
#pragma unroll
        for (int j = 0; j < SL; j++) Y[(size_t)(SL * h + j) * ldy + orow] = acc[j];
    }
}

} // namespace mi355
