// spmv_bcsr_sell.hpp — the blocked product as ONE contiguous stream per wave (round 4).
//
// SpMV_BCSR{,_OPT,_FMA,_AVX2}(y, x, A), mpk/SpMV.cpp:90-219 — and, through the blocked copy mi_csr_create makes of an FE matrix,
// SpMV_CSR of the reference's own matrix family (mpk/log/log_SPMV.txt:1-90: 44-58 nonzeros per row in 4x4 node blocks).
// Arithmetic unchanged: lane (r, q) owns row 4*bi + q of block row bi and runs ONE fma chain over the row's blocks in storage
// order, columns 0..3 inside a block — the bits of SpMV_BCSR_FMA (mpk/SpMV.cpp:150-178).
//
// What changed is where the bytes lie and how they are asked for.  spmv_bcsr4 (spmv_kernels.hpp) gives every block row to a quad
// that walks its ~15 blocks with a two-deep pipeline of its own: each load instruction of a wave touches sixteen 128-byte lines
// at 64 bytes each (the second instruction touches the same sixteen again, which is why non-temporal loads lose 30 % there), a
// wave lives for ~8 round trips of which two are its start (row pointers, then the first blocks) and one its drain, and the
// launch is 5 133 workgroups coming and going.  It streams its 660 MB at 5.85 TB/s whatever the order of its rows or the
// source of x — the rate of TEMPORAL loads on this part (profiles/NOTES.md) — where a non-temporal sweep reads 6.4-7.0.
//
// Here the library keeps a SLICED copy of the block values (mi_bcsr4_create; a format of its own making, like the CSR ring's
// 16-bit column stream): a slice is 16 consecutive block rows — one wave: 16 quads — padded to the slice's longest row (0.9 % on
// the 68^3 FE matrix), and step j of a slice holds the j-th block of each of its rows as TWO contiguous kilobytes:
//     val[step][0][lane] = {a(q,0), a(q,1)}      val[step][1][lane] = {a(q,2), a(q,3)}      lane = 4*r + q
// so that each of the wave's two 16-byte loads per step reads one contiguous KiB — whole lines, once — and may be non-temporal.
// A wave is PERSISTENT and owns a contiguous range of slices: its whole input is one contiguous stream (values) beside a second
// (block columns, 64 B per step), prefetched D steps ahead with unconditional, counted loads that never look at a row or slice
// boundary; the boundary (every ~15 steps, wave-uniform since slices are padded) only decides when the accumulators are set aside.
// Tried and dropped: a POOL of the last 2-12 % of the slices taken dynamically (one returning atomic per grab, issued when the wave's
// pipeline is empty) to even out the 15 us over which the waves end — two atomics per wave on one line (the failed grab, the "I am
// through" count that re-arms the pool) are 2 048 serialized atomics at the end of the launch: 140 us with an EMPTY pool.  Which
// waves end late is only half systematic (correlation 0.5-0.7 between launches; it follows the physical XCD placement, which
// alternates), so static re-weighting has nothing stable to hold on to.
// What tools/sell_bench.hip measured on the way (FE-shaped pattern, 328 509 block rows; profiles/r04_sell_bench.txt):
//   * the product with each slice's 512 bytes of y stored as they complete: 117 us — and 95.5 us with the stores compiled out: 10 MB of
//     stores among the loads cost 21 us, whatever their flavour (non-temporal 117.8, write-through 115.9).  PARKED in LDS and stored
//     behind the wave's last load: 105 us.  (A store among streaming reads costs the read stream many times its bytes: the CSR ring
//     kernel met the same effect in round 1.)
//   * of two workgroups on a CU — and of two waves on a SIMD — the one dispatched first takes the memory pipeline and ends ~40 us before
//     the other, which then runs alone and under-subscribed (trace: 1 024 waves end within 15 us, the other 1 024 another 35 us later).
//     ONE wave per SIMD with a deeper pipeline (D = 8) ends within 15 us: 100.6 us (0.83 of 8 TB/s on the 132 B / block model).
// x is gathered through L1 / L2 as in spmv_bcsr4 (columns one round ahead of x): measured there, its source does not matter.
// Padding steps are never multiplied (a lane past its row's end keeps its sum: fma(0, x, s) could flip a -0).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "spmv_kernels.hpp" // quad_bcast, spmm_block_update (the multi-vector form below)

namespace mi355 {

constexpr int kSellRows = 16;          // block rows per slice = quads per wave
constexpr int kSellStepDoubles = 256;  // 16 blocks x 16 values
constexpr int kSellPadSteps = 48;      // steps of padding behind the last one: the stream's loads run ahead unclamped (3 * D <= 48)
constexpr unsigned kSellPadCol = 0x80000000u;   // column entry of a padding place (a row shorter than its slice): bit 31; x is read at node 0, nothing is multiplied
constexpr unsigned kSellFirstCol = 0x40000000u; // set on all 16 entries of the first step of every slice: where the sums of one slice end
constexpr unsigned kSellColMask = 0x3fffffffu;  // (block columns stay below 2^30: 4 * 8 bytes per node already make that 32 GB of x)

struct SellView {
    const double* val;     // [nsteps + kSellPadSteps][2][64][2]
    const unsigned* col;   // [nsteps + kSellPadSteps][16]; bit 31 (kSellPadCol) where a row has no block at that step, bit 30 on a slice's first step
    const int* sptr;       // [nslices + 3] first step of each slice (three terminators, so that look-ahead reads stay in bounds)
    const int* wrng;       // [nwaves + 1] slices of each wave
    int nslices, nbrows;
    const int* browmap = nullptr; // nullptr, or block row bi writes y[4 * browmap[bi] + q] (the blocked copy of a relabelled matrix, reorder.hpp)
};

// host: the slice table of a block pattern — sptr (with its three terminators), the sliced column stream, the wave ranges
// balanced by steps
struct SellPlanHost {
    int nslices = 0, nwaves = 0;
    long long nsteps = 0;
    std::vector<int> sptr, wrng;
    std::vector<unsigned> col;
};

// waves: contiguous slice ranges of (nearly) equal step counts; a multiple of 32 (8 XCDs x 4 waves per workgroup)
inline void build_sell_wave_ranges(const SellPlanHost& P, int nwaves_max, std::vector<int>& wrng, int& nwaves, int nslices_static = -1);

inline void build_sell_plan(int nbrows, const int* ptrow, const int* indcol, int nwaves_max, SellPlanHost& P)
{
    P.nslices = (nbrows + kSellRows - 1) / kSellRows;
    P.sptr.assign((size_t)P.nslices + 3, 0);
    long long t = 0;
    for (int s = 0; s < P.nslices; s++) {
        int L = 1; // a slice of empty rows still takes one (padding) step: a boundary is then always a step apart from the next
        for (int r = 0; r < kSellRows && kSellRows * s + r < nbrows; r++) L = std::max(L, ptrow[kSellRows * s + r + 1] - ptrow[kSellRows * s + r]);
        P.sptr[s] = (int)t;
        t += L;
    }
    P.col.assign((size_t)(t + kSellPadSteps) * kSellRows, kSellPadCol);
    for (int s = 0; s < P.nslices; s++)
        for (int r = 0; r < kSellRows && kSellRows * s + r < nbrows; r++) {
            const int b0 = ptrow[kSellRows * s + r], n = ptrow[kSellRows * s + r + 1] - b0;
            for (int j = 0; j < n; j++) P.col[((size_t)P.sptr[s] + j) * kSellRows + r] = (unsigned)indcol[b0 + j];
        }
    for (int s = 0; s < P.nslices; s++)
        for (int r = 0; r < kSellRows; r++) P.col[(size_t)P.sptr[s] * kSellRows + r] |= kSellFirstCol;
    for (int r = 0; r < kSellRows; r++) P.col[(size_t)t * kSellRows + r] |= kSellFirstCol; // (the step behind the last slice: never consumed)
    P.nsteps = t;
    P.sptr[P.nslices] = P.sptr[P.nslices + 1] = P.sptr[P.nslices + 2] = (int)t;
    build_sell_wave_ranges(P, nwaves_max, P.wrng, P.nwaves);
}

// nslices_static >= 0: the ranges cover the slices [0, nslices_static) only (the rest is the pool of the POOL form)
inline void build_sell_wave_ranges(const SellPlanHost& P, int nwaves_max, std::vector<int>& wrng, int& nwaves, int nslices_static)
{
    const int NS = nslices_static >= 0 ? std::min(nslices_static, P.nslices) : P.nslices;
    const int W = std::min(nwaves_max, std::max(32, (P.nslices / 2 + 31) / 32 * 32));
    nwaves = W;
    wrng.assign((size_t)W + 1, NS);
    wrng[0] = 0;
    const long long steps = P.sptr[NS];
    int s = 0;
    for (int w = 1; w < W; w++) {
        const long long target = steps * w / W;
        while (s < NS && P.sptr[s] < target) s++;
        wrng[w] = s;
    }
}

typedef double sell_v2d __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ sell_v2d sell_ld(const sell_v2d* p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

// fills the sliced copy from the row-major blocks: one wave per slice (setup and value refreshes; never per product)
__global__ __launch_bounds__(64) void bcsr4_to_sell_kernel(int nslices, int nbrows, const int* __restrict__ ptrow, const double* __restrict__ coef,
                                                           const int* __restrict__ sptr, double* __restrict__ val)
{
    const int lane = threadIdx.x, r = lane >> 2, q = lane & 3;
    for (int s = blockIdx.x; s < nslices; s += gridDim.x) {
        const int bi = kSellRows * s + r;
        const int b0 = bi < nbrows ? ptrow[bi] : 0, n = bi < nbrows ? ptrow[bi + 1] - b0 : 0;
        const int t0 = sptr[s], L = sptr[s + 1] - t0;
        for (int j = 0; j < L; j++) {
            sell_v2d lo = {0.0, 0.0}, hi = {0.0, 0.0};
            if (j < n) {
                const sell_v2d* src = reinterpret_cast<const sell_v2d*>(coef + 16 * (size_t)(b0 + j) + 4 * q);
                lo = src[0];
                hi = src[1];
            }
            sell_v2d* dst = reinterpret_cast<sell_v2d*>(val + (size_t)(t0 + j) * kSellStepDoubles);
            dst[lane] = lo;
            dst[64 + lane] = hi;
        }
    }
}

// A CSR handle's blocked copy refreshed from (new) CSR values in ONE pass (round 5; mi_csr_update_values*: a Newton loop's Jacobian,
// src/solve_newton.c:1245-1247): a workgroup per slice of 16 block rows reads the slice's CSR segment (64 consecutive rows: contiguous)
// once, coalesced, into LDS and writes from there (a) the handle's CSR values when csr_out is given, (b) the row-major 4x4 blocks, (c) the
// sliced values — where round 4 made three passes (device-to-device copy, blocks gathered lane by lane from the CSR rows, sliced values
// gathered lane by lane from the blocks): 888 us per update of the FE matrix, nine products.  CAP = LDS doubles (0: read straight from src).
template <int CAP>
__global__ __launch_bounds__(256) void bcsr4_refresh_kernel(int nslices, int nbrows, const int* __restrict__ csr_ptrow, const double* __restrict__ src,
                                                            double* __restrict__ csr_out, const int* __restrict__ bptr, double* __restrict__ bval,
                                                            const int* __restrict__ sptr, double* __restrict__ sell_val)
{
    __shared__ double buf[CAP > 0 ? CAP : 1];
    __shared__ int rp[4 * kSellRows + 1]; // CSR row pointers of the slice's 64 rows
    __shared__ int bp[kSellRows + 1];     // block pointers of its 16 block rows
    const int tid = threadIdx.x;
    for (int s = blockIdx.x; s < nslices; s += gridDim.x) {
        if (tid <= 4 * kSellRows) rp[tid] = csr_ptrow[min(4 * kSellRows * s + tid, 4 * nbrows)];
        if (tid <= kSellRows) bp[tid] = bptr[min(kSellRows * s + tid, nbrows)];
        __syncthreads();
        const int base = rp[0], seg = rp[4 * kSellRows] - base;
        const bool staged = CAP > 0 && seg <= CAP; // (workgroup-uniform)
        if (staged) {
            for (int k0 = tid; k0 < seg; k0 += 4 * 256) { // four loads in flight per thread
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = src[base + min(k0 + 256 * u, seg - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (k0 + 256 * u < seg) {
                        buf[k0 + 256 * u] = v[u];
                        if (csr_out) csr_out[base + k0 + 256 * u] = v[u];
                    }
            }
            __syncthreads();
        } else if (csr_out) {
            for (int k = tid; k < seg; k += 256) csr_out[base + k] = src[base + k];
        }
        auto at = [&](int row, int pos) -> double { return staged ? buf[rp[row] - base + pos] : src[rp[row] + pos]; }; // value `pos` of the slice's CSR row `row`
        // (b) the blocks, block row by block row (16 x 15 = 240 values per FE block row: one turn of the 256 threads, contiguous in bval)
        for (int r = 0; r < kSellRows; r++) {
            const int nb = bp[r + 1] - bp[r];
            for (int k = tid; k < 16 * nb; k += 256) bval[16 * (size_t)bp[r] + k] = at(4 * r + ((k >> 2) & 3), 4 * (k >> 4) + (k & 3));
        }
        // (c) the sliced values: step j = block j of every row as [half][lane][2]
        if (sell_val) {
            const int t0 = sptr[s], L = sptr[s + 1] - t0;
            for (int k = tid; k < L * 128; k += 256) {
                const int j = k >> 7, h = (k >> 6) & 1, lane = k & 63, r = lane >> 2, q = lane & 3;
                sell_v2d v = {0.0, 0.0};
                if (j < bp[r + 1] - bp[r]) {
                    v.x = at(4 * r + q, 4 * j + 2 * h);
                    v.y = at(4 * r + q, 4 * j + 2 * h + 1);
                }
                reinterpret_cast<sell_v2d*>(sell_val + (size_t)(t0 + j) * kSellStepDoubles)[h * 64 + lane] = v;
            }
        }
        __syncthreads(); // rp / bp / buf are rewritten for the next slice
    }
}

// The same for the BCSR API (mi_bcsr4_update_values*_dev; the PETSc seam hands over MATSEQBAIJ's column-major blocks: COLMAJOR, transposed on
// the way — src/kernels/baij4_mad.c:73-76): a slice's blocks are contiguous in the caller's array; read once into LDS, written as the handle's
// row-major blocks (bval_out; null when src IS that array) and as sliced values.  Round 4: a copy / transpose pass, then the lane-by-lane fill.
template <int CAP, bool COLMAJOR>
__global__ __launch_bounds__(256) void bcsr4_blocks_refresh_kernel(int nslices, int nbrows, const int* __restrict__ bptr, const double* __restrict__ src,
                                                                   double* __restrict__ bval_out, const int* __restrict__ sptr, double* __restrict__ sell_val)
{
    __shared__ double buf[CAP > 0 ? CAP : 1];
    __shared__ int bp[kSellRows + 1];
    const int tid = threadIdx.x;
    for (int s = blockIdx.x; s < nslices; s += gridDim.x) {
        if (tid <= kSellRows) bp[tid] = bptr[min(kSellRows * s + tid, nbrows)];
        __syncthreads();
        const size_t base = 16 * (size_t)bp[0];
        const int seg = 16 * (bp[kSellRows] - bp[0]);
        const bool staged = CAP > 0 && seg <= CAP; // (workgroup-uniform)
        const bool copy_in_stage = staged && !COLMAJOR && bval_out != nullptr; // row-major in, row-major out: the copy rides in the staging loop
        if (staged) {
            for (int k0 = tid; k0 < seg; k0 += 4 * 256) { // four loads in flight per thread
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = src[base + min(k0 + 256 * u, seg - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (k0 + 256 * u < seg) {
                        buf[k0 + 256 * u] = v[u];
                        if (copy_in_stage) bval_out[base + k0 + 256 * u] = v[u];
                    }
            }
            __syncthreads();
        }
        auto el = [&](int blk, int q, int c) -> double { // element (row q, column c) of the slice's block `blk`, whatever the caller's layout
            const int o = 16 * blk + (COLMAJOR ? 4 * c + q : 4 * q + c);
            return staged ? buf[o] : src[base + o];
        };
        if (bval_out && !copy_in_stage)
            for (int k = tid; k < seg; k += 256) bval_out[base + k] = el(k >> 4, (k >> 2) & 3, k & 3);
        if (sell_val) {
            const int t0 = sptr[s], L = sptr[s + 1] - t0;
            for (int k = tid; k < L * 128; k += 256) {
                const int j = k >> 7, h = (k >> 6) & 1, lane = k & 63, r = lane >> 2, q = lane & 3;
                sell_v2d v = {0.0, 0.0};
                if (j < bp[r + 1] - bp[r]) {
                    const int blk = bp[r] - bp[0] + j;
                    v.x = el(blk, q, 2 * h);
                    v.y = el(blk, q, 2 * h + 1);
                }
                reinterpret_cast<sell_v2d*>(sell_val + (size_t)(t0 + j) * kSellStepDoubles)[h * 64 + lane] = v;
            }
        }
        __syncthreads();
    }
}

// D steps of values and x in flight per lane, the block columns of D more.  One wave = one contiguous range of slices.
// Workgroups of 256 threads = 4 independent waves; workgroup b is taken as logical workgroup (b % 8) * (G / 8) + b / 8, so that
// the workgroups that share an XCD (b, b + 8, ...) stream neighbouring slices and share their x lines in that XCD's L2.
// The loop never loads a row pointer or a slice table: where a slice begins is a flag in the column stream it is reading anyway
// (kSellFirstCol on every entry of a slice's first step), read back through readfirstlane so that the branch is a scalar one.
// Each stage is refilled AFTER its old contents have been multiplied, into the same registers: hipcc then needs no copies at the
// back edge and keeps counted waits around the loop (with the refill in front of the fmas it drained every load once per trip).
// YM: how a finished slice's 64 sums reach y.  0: one 512-byte store at once; 1: the same, non-temporal; 3: write-through (sc1);
//     2: PARKED in LDS (kSellPark slices per wave) and stored when the park is full or the wave's range ends — for matrices whose
//     waves own no more slices than the park holds, every store of the launch is issued behind the wave's last load.
// ABL (tools/sell_bench.hip only; results invalid): 1 no x gather (x entries taken from a register), 2 no y stores, 4 no column stream
constexpr int kSellPark = 32;
// NW: waves per workgroup.  The waves of ONE workgroup advance alike; of two workgroups on a CU the one dispatched first takes the
// memory pipeline and ends 40 us before the other (tools/sell_bench.hip, trace), which then runs alone and under-subscribed: one
// workgroup of 8 waves per CU, not two of 4.
template <int D, bool NT, int ABL = 0, int YM = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW) void spmv_bcsr4_sell(SellView S, const double* __restrict__ x, double* __restrict__ y, int nwg)
{
    __shared__ double s_park[YM == 2 ? NW * kSellPark * 64 : 1];
    __shared__ int s_pid[YM == 2 ? NW * kSellPark : 1];
    double* park = s_park + (YM == 2 ? ((int)threadIdx.x >> 6) * kSellPark * 64 + ((int)threadIdx.x & 63) : 0);
    int* pid = s_pid + (YM == 2 ? ((int)threadIdx.x >> 6) * kSellPark : 0);
    int parked = 0; // (wave-uniform)
    const int per = nwg >> 3;
    const int lwg = per > 0 && (nwg & 7) == 0 ? ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(lwg * NW + ((int)threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane >> 2;
    const sell_v2d* vbase = reinterpret_cast<const sell_v2d*>(S.val) + lane;
    const unsigned* cbase = S.col + r;

    auto flush = [&]() {
        for (int j = 0; j < parked; j++) {
            const int bj = kSellRows * pid[j] + r;
            if (bj < S.nbrows) y[4 * (size_t)(S.browmap ? S.browmap[bj] : bj) + (lane & 3)] = park[j * 64];
        }
        parked = 0;
    };
    // the sums of slice `sl` are complete
    auto emit = [&](int sl, double v) {
        const int bi = kSellRows * sl + r;
        double* dst = y + 4 * (size_t)((S.browmap && bi < S.nbrows) ? S.browmap[bi] : bi) + (lane & 3);
        if ((ABL & 2) && v != 123.456) return;
        if (YM == 2) {
            park[parked * 64] = v;
            if (lane == 0) pid[parked] = sl;
            parked++;
            if (parked == kSellPark) flush();
        } else if (bi < S.nbrows) {
            if (YM == 1) __builtin_nontemporal_store(v, dst);
            else if (YM == 3) __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = v;
        }
    };

    const int s_begin = __builtin_amdgcn_readfirstlane(S.wrng[wave]);
    const int s_end = __builtin_amdgcn_readfirstlane(S.wrng[wave + 1]);
    {
        if (s_begin < s_end) {
            const int t0 = __builtin_amdgcn_readfirstlane(S.sptr[s_begin]);
            const int t_end = __builtin_amdgcn_readfirstlane(S.sptr[s_end]);
            int s = s_begin - 1; // the range's first step carries the flag too: the boundary code runs there and counts s up to s_begin
            sell_v2d a01[D], a23[D], x01[D], x23[D];
            unsigned cn[D];  // column entries of steps t + D + d (one round ahead of the values and x)
            unsigned fl[D];  // flags of the step whose values stage d holds: bit 1 padding place, bit 0 first step of a slice
#pragma unroll
            for (int d = 0; d < D; d++) {
                const sell_v2d* p = vbase + (size_t)(t0 + d) * (kSellStepDoubles / 2);
                a01[d] = sell_ld<NT>(p);
                a23[d] = sell_ld<NT>(p + 64);
                cn[d] = cbase[(size_t)(t0 + d) * kSellRows];
            }
#pragma unroll
            for (int d = 0; d < D; d++) {
                fl[d] = cn[d] >> 30;
                const sell_v2d* xb = reinterpret_cast<const sell_v2d*>(x + 4 * (size_t)(cn[d] & kSellColMask));
                if (ABL & 1) {
                    x01[d] = x23[d] = sell_v2d{1.0 + lane, 0.5};
                } else {
                    x01[d] = xb[0];
                    x23[d] = xb[1];
                }
            }
#pragma unroll
            for (int d = 0; d < D; d++) cn[d] = cbase[(size_t)(t0 + D + d) * kSellRows];

            double acc = 0.0;
            for (int t = t0; t < t_end; t += D) {
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const int i = t + d; // the step consumed now
                    if (i < t_end) { // (wave-uniform)
                        const unsigned f = fl[d];
                        if (__builtin_amdgcn_readfirstlane(f) & 1u) { // a slice begins: the one before it is complete
                            if (i != t0) emit(s, acc);
                            acc = 0.0;
                            s++;
                        }
                        double n = fma(a01[d].x, x01[d].x, acc);
                        n = fma(a01[d].y, x01[d].y, n);
                        n = fma(a23[d].x, x23[d].x, n);
                        n = fma(a23[d].y, x23[d].y, n);
                        acc = (f & 2u) ? acc : n; // padding places are not multiplied
                    }
                    // refill stage d with step i + D (its column arrived a round ago), then ask for the column of step i + 2 D
                    const sell_v2d* p = vbase + (size_t)(i + D) * (kSellStepDoubles / 2);
                    a01[d] = sell_ld<NT>(p);
                    a23[d] = sell_ld<NT>(p + 64);
                    const unsigned c = cn[d];
                    fl[d] = c >> 30;
                    const sell_v2d* xb = reinterpret_cast<const sell_v2d*>(x + 4 * (size_t)(c & kSellColMask));
                    if (!(ABL & 1)) {
                        x01[d] = xb[0];
                        x23[d] = xb[1];
                    }
                    // (the old column entry is dead from here: pinning its last uses in front of the reload lets the new entry land in the
                    // same register — left to itself hipcc computed the flags at the bottom of the loop, kept both entries alive, copied at
                    // the back edge and put an s_waitcnt vmcnt(0) in front of the copies: every load drained once per trip)
                    asm volatile("" ::"v"(fl[d]), "v"(xb));
                    if (!(ABL & 4)) cn[d] = cbase[(size_t)(i + 2 * D) * kSellRows];
                }
            }
            emit(s, acc);
        }
    }
    unsigned long long t_loads_done = 0;
    if (ABL & 8) t_loads_done = __builtin_amdgcn_s_memrealtime();
    if (YM == 2 && !(ABL & 16)) flush(); // what is still parked
    if (ABL & 8) { // TRACE (tools/sell_bench.hip): when this wave's loop ended and when its stores were out, 100 MHz ticks, behind y
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end_all = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* tr = reinterpret_cast<unsigned long long*>(y + 4 * (size_t)S.nbrows) + 2 * (size_t)wave;
            tr[0] = t_loads_done;
            tr[1] = t_end_all;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The multi-vector product on the same stream: Y[:, j] = A X[:, j], j < S (S = 4 or 8), the matrix read once
// (MatMatMult_SeqBAIJ_4_AVX2, src/kernels/spmm_avx2.c:7-109; dense column-major X, Y).  Same waves, slices and flags as
// spmv_bcsr4_sell; lane (r, q) keeps S sums (row 4 bi + q of every column) and LOADS the x blocks of the columns q, q + 4 only —
// the quad shares them through DPP (quad_bcast), so a step issues 2 + 2 S / 4 sixteen-byte loads per lane, as few as the
// single-vector product at four columns.  Each (row, column) sum is one lane's sequential chain: ARITH_CHAIN = the bits of
// SpMV_BCSR_FMA per column, ARITH_BLOCKACC = spmm_avx2.c's per-block partial sums (spmm_block_update, spmv_kernels.hpp).
// A finished slice's 64 x S sums are parked in LDS (PARK slices per wave) and stored column by column — 512 contiguous bytes
// each — when the park is full or the wave's range ends.
template <int S, int ARITH, int D, bool NT>
__global__ __launch_bounds__(256) void spmm_bcsr4_sell(SellView Sv, const double* __restrict__ X, long long ldx, double* __restrict__ Y, long long ldy, int nwg)
{
    static_assert(S == 4 || S == 8, "the sliced multi-vector form is built for four and eight columns");
    constexpr int G = S / 4;
    constexpr int PARK = S == 4 ? 16 : 8; // 4 waves x PARK x S x 512 B = 128 KB of LDS: one workgroup per CU, one wave per SIMD
    __shared__ double s_park[4 * PARK * S * 64];
    __shared__ int s_pid[4 * PARK];
    double* park = s_park + ((int)threadIdx.x >> 6) * PARK * S * 64 + ((int)threadIdx.x & 63);
    int* pid = s_pid + ((int)threadIdx.x >> 6) * PARK;
    int parked = 0;
    const int per = nwg >> 3;
    const int lwg = per > 0 && (nwg & 7) == 0 ? ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(lwg * 4 + ((int)threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane >> 2, q = lane & 3;
    const int s_begin = __builtin_amdgcn_readfirstlane(Sv.wrng[wave]);
    const int s_end = __builtin_amdgcn_readfirstlane(Sv.wrng[wave + 1]);
    if (s_begin >= s_end) return;
    const int t0 = __builtin_amdgcn_readfirstlane(Sv.sptr[s_begin]);
    const int t_end = __builtin_amdgcn_readfirstlane(Sv.sptr[s_end]);
    int s = s_begin - 1;
    const sell_v2d* vbase = reinterpret_cast<const sell_v2d*>(Sv.val) + lane;
    const unsigned* cbase = Sv.col + r;
    const double* Xq = X + (size_t)q * ldx; // this lane's first column; its others are 4 * ldx apart

    auto flush = [&]() {
        for (int j = 0; j < parked; j++) {
            const int bj = kSellRows * pid[j] + r;
            if (bj < Sv.nbrows) {
                const size_t orow = 4 * (size_t)(Sv.browmap ? Sv.browmap[bj] : bj) + q;
#pragma unroll
                for (int c = 0; c < S; c++) Y[(size_t)c * ldy + orow] = park[(j * S + c) * 64];
            }
        }
        parked = 0;
    };
    double acc[S];
#pragma unroll
    for (int c = 0; c < S; c++) acc[c] = 0.0;
    auto emit = [&](int sl) {
#pragma unroll
        for (int c = 0; c < S; c++) park[(parked * S + c) * 64] = acc[c];
        if (lane == 0) pid[parked] = sl;
        parked++;
        if (parked == PARK) flush();
    };

    sell_v2d a01[D], a23[D], x01[D][G], x23[D][G];
    unsigned cn[D], fl[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const sell_v2d* p = vbase + (size_t)(t0 + d) * (kSellStepDoubles / 2);
        a01[d] = sell_ld<NT>(p);
        a23[d] = sell_ld<NT>(p + 64);
        cn[d] = cbase[(size_t)(t0 + d) * kSellRows];
    }
#pragma unroll
    for (int d = 0; d < D; d++) {
        fl[d] = cn[d] >> 30;
#pragma unroll
        for (int u = 0; u < G; u++) {
            const sell_v2d* xb = reinterpret_cast<const sell_v2d*>(Xq + (size_t)(4 * u) * ldx + 4 * (size_t)(cn[d] & kSellColMask));
            x01[d][u] = xb[0];
            x23[d][u] = xb[1];
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) cn[d] = cbase[(size_t)(t0 + D + d) * kSellRows];

    for (int t = t0; t < t_end; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int i = t + d;
            if (i < t_end) { // (wave-uniform)
                const unsigned f = fl[d];
                if (__builtin_amdgcn_readfirstlane(f) & 1u) {
                    if (i != t0) emit(s);
#pragma unroll
                    for (int c = 0; c < S; c++) acc[c] = 0.0;
                    s++;
                }
                const bool pad = (f & 2u) != 0; // uniform within the quad (its four lanes share the block row): DPP sources are live either way
                const double2 c01 = make_double2(a01[d].x, a01[d].y), c23 = make_double2(a23[d].x, a23[d].y);
#pragma unroll
                for (int u = 0; u < G; u++) { // columns 4u + K come from lane K of the quad
                    const double n0 = spmm_block_update<ARITH>(acc[4 * u + 0], c01, c23, quad_bcast<0>(x01[d][u].x), quad_bcast<0>(x01[d][u].y), quad_bcast<0>(x23[d][u].x), quad_bcast<0>(x23[d][u].y));
                    const double n1 = spmm_block_update<ARITH>(acc[4 * u + 1], c01, c23, quad_bcast<1>(x01[d][u].x), quad_bcast<1>(x01[d][u].y), quad_bcast<1>(x23[d][u].x), quad_bcast<1>(x23[d][u].y));
                    const double n2 = spmm_block_update<ARITH>(acc[4 * u + 2], c01, c23, quad_bcast<2>(x01[d][u].x), quad_bcast<2>(x01[d][u].y), quad_bcast<2>(x23[d][u].x), quad_bcast<2>(x23[d][u].y));
                    const double n3 = spmm_block_update<ARITH>(acc[4 * u + 3], c01, c23, quad_bcast<3>(x01[d][u].x), quad_bcast<3>(x01[d][u].y), quad_bcast<3>(x23[d][u].x), quad_bcast<3>(x23[d][u].y));
                    acc[4 * u + 0] = pad ? acc[4 * u + 0] : n0; // padding places are not multiplied
                    acc[4 * u + 1] = pad ? acc[4 * u + 1] : n1;
                    acc[4 * u + 2] = pad ? acc[4 * u + 2] : n2;
                    acc[4 * u + 3] = pad ? acc[4 * u + 3] : n3;
                }
            }
            const sell_v2d* p = vbase + (size_t)(i + D) * (kSellStepDoubles / 2);
            a01[d] = sell_ld<NT>(p);
            a23[d] = sell_ld<NT>(p + 64);
            const unsigned c = cn[d];
            fl[d] = c >> 30;
            const double* xrow = Xq + 4 * (size_t)(c & kSellColMask);
#pragma unroll
            for (int u = 0; u < G; u++) {
                const sell_v2d* xb = reinterpret_cast<const sell_v2d*>(xrow + (size_t)(4 * u) * ldx);
                x01[d][u] = xb[0];
                x23[d][u] = xb[1];
            }
            asm volatile("" ::"v"(fl[d]), "v"(xrow)); // (the old column entry dies here: see spmv_bcsr4_sell)
            cn[d] = cbase[(size_t)(i + 2 * D) * kSellRows];
        }
    }
    emit(s);
    flush();
}

} // namespace mi355
