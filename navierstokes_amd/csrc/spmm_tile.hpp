// spmm_tile.hpp — the multi-vector product Y[:, j] = A X[:, j], j < S, on the blocked matrix with each workgroup's x blocks
// gathered ONCE into LDS (MatMatMult_SeqBAIJ_4_AVX2, src/kernels/spmm_avx2.c:7-109; the tile idea of spmv_bcsr4_tile at S columns).
//
// Why: in spmm_bcsr4 / spmm_bcsr4_quad (spmv_kernels.hpp) every block costs its lanes a dependent chain — block column from
// memory, THEN the S x blocks through L1/L2 — and S x blocks of 32 bytes from S different arrays per block: 4.6 M blocks x S
// scattered 32-byte reads.  The kernel sat at 45-54 % of its byte model and fetched 1.31x the model's bytes (x once per XCD).
// Here a workgroup of 512 threads owns 128 block rows; the host lists the distinct block columns they touch (~660 for the
// FE matrix: a 2.8x reuse) and gives every block the 16-bit position of its column in that list.  The workgroup gathers the S
// columns of those nodes into LDS once (node-major records of 4 S + 2 doubles: the two pad doubles spread neighbouring
// records over the banks), and the inner loop waits for COEFFICIENTS only — a pure stream, P blocks deep in registers — while
// x comes from LDS (2 S ds_read_b128 per block and lane, quads broadcast).
// What it buys and what it does not (FE matrix, 1.31 M rows, bench.py fe_spmm4 / fe_spmm8; profiles/r03_spmm_tile_ablation.txt):
// four columns 168-174 -> 152-156 us (this kernel, 128 block rows per workgroup).  The remaining gap to the coefficient
// stream's own time (113 us for the single-vector kernel) is the GATHER at the top of each workgroup: two dependent round
// trips (list entry, then the node's x block) during which the CU idles — the tile's LDS footprint (95 KB at four columns)
// admits ONE workgroup per CU.  Compiled out, the kernel runs 130 us (162 -> 130 on the box of that run; at eight columns with
// two quads per block row 263 -> 165); LDS bank conflicts cost nothing (every lane reading slot 0: 160 us).  Two remedies were
// built and measured slower: smaller tiles for two workgroups per CU (64 rows: 164 us — more distinct columns per row, more
// gather work) and persistent workgroups that fetch tile i + 1's x blocks into registers while tile i is multiplied (231 us /
// 414 us: vmcnt counts loads in issue order, so the first coefficient wait of a tile also waits for every x block issued in
// front of it — the latency is not hidden, and the staging costs registers and a second barrier).  At eight columns the
// gather forms stay ahead (223-236 us against 263).  What would hide it is a second set of waves that only gathers, with a
// second LDS tile — which 160 KB do not hold at these tile sizes.
// Arithmetic: exactly spmm_bcsr4's (ARITH 0: one fma chain per row and column, bit-equal to SpMV_BCSR_FMA; ARITH 1: per-block
// partial sums added to the row value) — same blocks in the same order.
#pragma once
#include <hip/hip_runtime.h>

#include "spmv_kernels.hpp"

namespace mi355 {

constexpr int kSpmmTileThreads = 512;              // 128 block rows per workgroup
constexpr int kSpmmTileRows = kSpmmTileThreads / 4;

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
    const int g = wg * T + tid;
    const int bi = min(g >> 2, A.nbrows - 1), q = g & 3; // (lanes past the last block row shadow it and store nothing)
    const bool live = (g >> 2) < A.nbrows;
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
    const int total = U * S;
    for (int e0 = tid; e0 < total; e0 += 4 * T) {
        double2 v0[4], v1[4];
        int jj[4], uu[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int e = min(e0 + r * T, total - 1);
            jj[r] = e / U;
            uu[r] = e - jj[r] * U;
            const unsigned node = Tl.nodes[u0 + uu[r]];
            const double2* xb = reinterpret_cast<const double2*>(X + (size_t)jj[r] * ldx + 4 * (size_t)node);
            v0[r] = xb[0];
            v1[r] = xb[1];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (e0 + r * T < total) {
                double2* d = reinterpret_cast<double2*>(s_xt + (size_t)uu[r] * REC + 4 * jj[r]);
                d[0] = v0[r];
                d[1] = v1[r];
            }
    }
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

} // namespace mi355
