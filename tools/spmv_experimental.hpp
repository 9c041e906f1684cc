// tools/spmv_experimental.hpp — kernel variants under evaluation (development only).
// A variant graduates into navierstokes_amd/csrc/spmv_kernels.hpp once it is
// bit-exact and measurably faster on the GPU box; nothing in the product library
// includes this file.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "spmv_kernels.hpp"

static int g_plan_rr = 0;        // kbench plan builder: round-robin run length (0: private contiguous runs)
static int g_last_plan_nblk = 0; // slots of the last plan built
static int g_want_slots = 0;                          // make_plan also builds the 16-bit column stream
static const unsigned short* g_last_slots = nullptr;  // ... and leaves it here

static unsigned long long* g_prof_ptr = nullptr;
static unsigned long long* g_prof_ptr2 = nullptr;
static int g_prof2_wgs = 0;

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
    std::vector<float> ms;
    bool ok = false;
};

namespace mi355 {

// E1: stream kernel with 16-byte coef / 8-byte indcol loads in phase 1.
template <int NNZB>
__global__ __launch_bounds__(kWG) void spmv_csr_stream_v2(CsrView A, const double* __restrict__ x,
                                                          double* __restrict__ y)
{
    constexpr int PER2 = NNZB / (2 * kWG);
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    const int b = xcd_remap(blockIdx.x, A.nblk);
    if (b >= A.nblk) return;
    const int tid = threadIdx.x;
    const int2 d0 = A.blk[b], d1 = A.blk[b + 1];
    const int r0 = d0.x, p0 = d0.y, r1 = d1.x, p1 = d1.y;
    const int nn = p1 - p0;
    if (nn > NNZB) { // long row: same serial path as the product kernel
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
        return;
    }
    int ra = 0, re = 0;
    const int myrow = r0 + tid;
    if (myrow < r1) {
        ra = A.ptrow[myrow] - p0;
        re = A.ptrow[myrow + 1] - p0;
    }
    const int head = p0 & 1;           // one leading element if the range starts odd
    const int q0 = p0 + head;          // even: 16-B aligned coef, 8-B aligned indcol
    const int npair = (p1 - q0) >> 1;
    const int tail = (p1 - q0) & 1;
    double2 c2[PER2];
    int2 j2[PER2];
#pragma unroll
    for (int i = 0; i < PER2; i++) {
        const int t = tid + i * kWG;
        if (t < npair) {
            c2[i] = *reinterpret_cast<const double2*>(A.coef + q0 + 2 * t);
            j2[i] = *reinterpret_cast<const int2*>(A.indcol + q0 + 2 * t);
        }
    }
    // stragglers: head element (thread 0) and tail element (thread 1)
    double cs = 0.0;
    int js = 0, ks = -1;
    if (tid == 0 && head) { ks = 0; cs = A.coef[p0]; js = A.indcol[p0]; }
    if (tid == 1 && tail) { ks = nn - 1; cs = A.coef[p1 - 1]; js = A.indcol[p1 - 1]; }
#pragma unroll
    for (int i = 0; i < PER2; i++) {
        const int t = tid + i * kWG;
        if (t < npair) {
            const int k = head + 2 * t;
            const double x0 = x[j2[i].x], x1 = x[j2[i].y];
            s_c[sk(k)] = c2[i].x;
            s_x[sk(k)] = x0;
            s_c[sk(k + 1)] = c2[i].y;
            s_x[sk(k + 1)] = x1;
        }
    }
    if (ks >= 0) {
        s_c[sk(ks)] = cs;
        s_x[sk(ks)] = x[js];
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
}

// E2: no XCD remap (plain blockIdx order) — isolates what the remap is worth.
template <int NNZB>
__global__ __launch_bounds__(kWG) void spmv_csr_stream_noremap(CsrView A, const double* __restrict__ x,
                                                               double* __restrict__ y)
{
    constexpr int PER = NNZB / kWG;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    const int b = blockIdx.x;
    if (b >= A.nblk) return;
    const int tid = threadIdx.x;
    const int2 d0 = A.blk[b], d1 = A.blk[b + 1];
    const int r0 = d0.x, p0 = d0.y, r1 = d1.x, p1 = d1.y;
    const int nn = p1 - p0;
    if (nn > NNZB) return; // experiment only: S15/SVAR/SFE have no long rows
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
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int k = tid + i * kWG;
        if (k < nn) {
            s_c[sk(k)] = c[i];
            s_x[sk(k)] = x[j[i]];
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
        y[r] = s;
    }
}

// E3: stream kernel that skips the gather (x := 1.0 for every column) — NOT a
// valid SpMV (reported WRONG); prices what the gather costs on top of the stream.
template <int NNZB>
__global__ __launch_bounds__(kWG) void spmv_csr_stream_nogather(CsrView A, const double* __restrict__ x,
                                                                double* __restrict__ y)
{
    constexpr int PER = NNZB / kWG;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    const int b = xcd_remap(blockIdx.x, A.nblk);
    if (b >= A.nblk) return;
    const int tid = threadIdx.x;
    const int2 d0 = A.blk[b], d1 = A.blk[b + 1];
    const int r0 = d0.x, p0 = d0.y, r1 = d1.x, p1 = d1.y;
    const int nn = p1 - p0;
    if (nn > NNZB) return;
    int ra = 0, re = 0;
    const int myrow = r0 + tid;
    if (myrow < r1) {
        ra = A.ptrow[myrow] - p0;
        re = A.ptrow[myrow + 1] - p0;
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int k = tid + i * kWG;
        if (k < nn) {
            s_c[sk(k)] = A.coef[p0 + k];
            s_x[sk(k)] = (double)A.indcol[p0 + k];
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
        y[r] = s;
    }
}


// E4: persistent workgroups with a SLIDING x WINDOW in LDS ("ring").
// Each workgroup owns a contiguous run of row blocks (XCD-aware: consecutive
// runs go to the same XCD).  The x entries the current block can reference,
// [cmin_b, cmax_b], live in an LDS ring indexed by column; moving to the next
// block only loads the few columns that entered the window (banded / FE-ordered
// matrices: ~rows-per-block new columns), so x is read from L2 about once per
// workgroup instead of once per nonzero, and the per-nonzero gather becomes a
// ds_read_b64.  A block whose columns do not fit the ring's current position
// (span too wide, or reaching back behind the window) takes the global gather.
// The matrix stream for block b+1 is issued before block b is reduced.
template <int T, int NNZB, int RING>
__global__ __launch_bounds__(T) void spmv_csr_ring(CsrView A, const double* __restrict__ x,
                                                   double* __restrict__ y, int bpw)
{
    constexpr int PER = NNZB / T;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int b_end = min(A.nblk, b_begin + bpw);
    if (b_begin >= b_end) return;

    // ring state (wave-uniform): holds x[c] for c in [wlo, whi) at s_ring[c - base (mod RING)]
    int wlo = 0, whi = 0, base = 0;
    bool ring_live = false;

    double c[PER], cn[PER];
    int j[PER], jn[PER];
    int ra = 0, re = 0, ran = 0, ren = 0;
    bool have = false;

    for (int b = b_begin; b < b_end; b++) {
        const int2 d0 = A.blk[b], d1 = A.blk[b + 1];
        const int r0 = d0.x, p0 = d0.y, r1 = d1.x, p1 = d1.y;
        const int nn = p1 - p0;
        const int myrow = r0 + tid;
        if (nn > NNZB) { // long row: serial exact path, global gather
            double s = 0.0;
            for (int bs = p0; bs < p1; bs += NNZB) {
                const int m = min(NNZB, p1 - bs);
                __syncthreads();
                for (int k = tid; k < m; k += T) {
                    s_c[sk(k)] = A.coef[bs + k];
                    s_x[sk(k)] = x[A.indcol[bs + k]];
                }
                __syncthreads();
                if (tid == 0)
                    for (int k = 0; k < m; k++) s = fma(s_c[sk(k)], s_x[sk(k)], s);
            }
            if (tid == 0) y[A.rowmap ? A.rowmap[r0] : r0] = s;
            have = false;
            continue;
        }
        if (!have) {
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = tid + i * T;
                if (k < nn) {
                    c[i] = A.coef[p0 + k];
                    j[i] = A.indcol[p0 + k];
                }
            }
            if (myrow < r1) {
                ra = A.ptrow[myrow] - p0;
                re = A.ptrow[myrow + 1] - p0;
            }
        }
        // ---- slide the window to cover [cmin, cmax]
        const int2 sp = A.blk_span[b];
        const int cmin = sp.x, cmax = sp.y;
        bool use_ring = (nn > 0) && (cmax - cmin + 1 <= RING);
        if (use_ring) {
            int lo = ring_live ? wlo : cmin;
            int hi = ring_live ? whi : cmin;
            if (cmin < lo) use_ring = false;        // reaches back behind the window
            else {
                if (cmin > hi) { lo = cmin; hi = cmin; } // jumped ahead: restart the window
                const int nhi = max(hi, cmax + 1);
                const int nlo = max(lo, nhi - RING);
                if (cmin < nlo) use_ring = false;   // cannot hold cmin..cmax at once
                else {
                    if (!ring_live || lo != wlo || hi != whi) { // restarted
                        base = (lo / RING) * RING;
                    }
                    while (nlo - base >= RING) base += RING;
                    for (int cc = hi + tid; cc < nhi; cc += T) {
                        int pos = cc - base;
                        if (pos >= RING) pos -= RING;
                        s_ring[pos] = x[cc];
                    }
                    wlo = nlo;
                    whi = nhi;
                    ring_live = true;
                }
            }
        }
        __syncthreads(); // ring ready; previous block's reduction is done with s_c/s_x
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = tid + i * T;
            if (k < nn) {
                double xv;
                if (use_ring) {
                    int pos = j[i] - base;
                    if (pos >= RING) pos -= RING;
                    xv = s_ring[pos];
                } else {
                    xv = x[j[i]];
                }
                s_c[sk(k)] = c[i];
                s_x[sk(k)] = xv;
            }
        }
        // ---- issue the matrix stream of the next block before reducing this one
        bool have_next = false;
        if (b + 1 < b_end) {
            const int2 e1 = A.blk[b + 2];
            const int np0 = p1, nr0 = r1, nr1 = e1.x, nnn = e1.y - p1;
            if (nnn <= NNZB) {
                have_next = true;
#pragma unroll
                for (int i = 0; i < PER; i++) {
                    const int k = tid + i * T;
                    if (k < nnn) {
                        cn[i] = A.coef[np0 + k];
                        jn[i] = A.indcol[np0 + k];
                    }
                }
                const int nrow = nr0 + tid;
                if (nrow < nr1) {
                    ran = A.ptrow[nrow] - np0;
                    ren = A.ptrow[nrow + 1] - np0;
                }
            }
        }
        __syncthreads();
        for (int r = myrow; r < r1; r += T) {
            if (r != myrow) {
                ra = A.ptrow[r] - p0;
                re = A.ptrow[r + 1] - p0;
            }
            double s = 0.0;
            for (int k = ra; k < re; k++) s = fma(s_c[sk(k)], s_x[sk(k)], s);
            y[A.rowmap ? A.rowmap[r] : r] = s;
        }
        if (have_next) {
#pragma unroll
            for (int i = 0; i < PER; i++) {
                c[i] = cn[i];
                j[i] = jn[i];
            }
            ra = ran;
            re = ren;
        }
        have = have_next;
    }
}


// E5: stream kernel, RED = 0 simple loop | 1 batched reduce (U=8) | 2 batched (U=16) | 3 no reduce (invalid, timing only)
template <int NNZB, int RED>
__global__ __launch_bounds__(kWG) void spmv_csr_stream3(CsrView A, const double* __restrict__ x,
                                                        double* __restrict__ y)
{
    constexpr int PER = NNZB / kWG;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    const int b = xcd_remap(blockIdx.x, A.nblk);
    if (b >= A.nblk) return;
    const int tid = threadIdx.x;
    const int2 d0 = A.blk[b], d1 = A.blk[b + 1];
    const int r0 = d0.x, p0 = d0.y, r1 = d1.x, p1 = d1.y;
    const int nn = p1 - p0;
    if (nn > NNZB) return; // experiment only
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
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int k = tid + i * kWG;
        if (k < nn) {
            s_c[sk(k)] = c[i];
            s_x[sk(k)] = x[j[i]];
        }
    }
    __syncthreads();
    if (RED == 3) {
        if (myrow < r1) y[myrow] = s_c[sk(ra)] + s_x[sk(ra)];
        return;
    }
    for (int r = myrow; r < r1; r += kWG) {
        if (r != myrow) {
            ra = A.ptrow[r] - p0;
            re = A.ptrow[r + 1] - p0;
        }
        double s = 0.0;
        if (RED == 0) for (int k = ra; k < re; k++) s = fma(s_c[sk(k)], s_x[sk(k)], s);
        else if (RED == 1) s = row_chain<8>(s_c, s_x, ra, re);
        else s = row_chain<16>(s_c, s_x, ra, re);
        y[r] = s;
    }
}

// E6: pure "stream + gather" upper bound: no LDS, no rows; every thread folds its
// own nonzeros (invalid SpMV, timing only).
template <int NNZB>
__global__ __launch_bounds__(kWG) void spmv_csr_stream_nolds(CsrView A, const double* __restrict__ x,
                                                             double* __restrict__ y)
{
    constexpr int PER = NNZB / kWG;
    const int b = xcd_remap(blockIdx.x, A.nblk);
    if (b >= A.nblk) return;
    const int tid = threadIdx.x;
    const int2 d0 = A.blk[b], d1 = A.blk[b + 1];
    const int p0 = d0.y, p1 = d1.y;
    const int nn = p1 - p0;
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
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int k = tid + i * kWG;
        if (k < nn) s = fma(c[i], x[j[i]], s);
    }
    if (d0.x + tid < d1.x) y[d0.x + tid] = s;
}


// ---------------------------------------------------------------------------
// E7: ring2 — E4 with the latencies taken off the per-block critical path:
//   * block metadata {r0, p0, cmin, cmax} of the workgroup's run is staged in LDS
//     once (no dependent scalar global loads per block);
//   * the matrix stream runs D blocks ahead in registers (static stage arrays,
//     block loop unrolled by D);
//   * the x columns entering the window for block b+1 are requested while block b
//     is reduced and written to the ring at the top of b+1;
//   * the row chains fetch their LDS operands 8 at a time.
// ---------------------------------------------------------------------------
struct RingState {
    int wlo, whi, base;
    bool live;
};
struct RingPlan {
    bool use;
    int lo, hi;      // columns [lo, hi) must be loaded into the ring
    RingState next;
};

template <int RING>
__device__ __forceinline__ RingPlan ring_plan(const RingState& st, int cmin, int cmax, int nn)
{
    RingPlan p;
    p.use = false;
    p.lo = p.hi = 0;
    p.next = st;
    if (nn <= 0 || cmax - cmin + 1 > RING) return p;
    int lo = st.live ? st.wlo : cmin;
    int hi = st.live ? st.whi : cmin;
    bool restart = !st.live;
    if (cmin < lo) return p;             // reaches back behind the window
    if (cmin > hi) { lo = cmin; hi = cmin; restart = true; }
    const int nhi = max(hi, cmax + 1);
    const int nlo = max(lo, nhi - RING);
    if (cmin < nlo) return p;            // cannot hold [cmin, cmax] at once
    int base = restart ? (lo / RING) * RING : st.base;
    while (nlo - base >= RING) base += RING;
    p.use = true;
    p.lo = hi;
    p.hi = nhi;
    p.next.wlo = nlo;
    p.next.whi = nhi;
    p.next.base = base;
    p.next.live = true;
    return p;
}

template <int RING>
__device__ __forceinline__ int ring_pos(int c, int base)
{
    int pos = c - base;
    return pos >= RING ? pos - RING : pos;
}

template <int T, int NNZB, int RING, int D>
__global__ __launch_bounds__(T) void spmv_csr_ring2(CsrView A, const int4* __restrict__ meta,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    int bpw)
{
    constexpr int PER = NNZB / T;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    constexpr int MAXB = 96;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ int4 s_meta[MAXB + 2];
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int b_end = min(A.nblk, b_begin + bpw);
    if (b_begin >= b_end) return;

    RingState st;
    st.wlo = st.whi = st.base = 0;
    st.live = false;

    double c[D][PER];
    int j[D][PER];
    int ra[D], re[D];

    for (int cb = b_begin; cb < b_end; cb += MAXB) {
        const int nb = min(MAXB, b_end - cb);
        __syncthreads();
        for (int i = tid; i <= nb; i += T) s_meta[i] = meta[cb + i];
        __syncthreads();

        auto issue = [&](int lb, int s) { // matrix stream of local block lb into stage s
            const int4 m0 = s_meta[lb], m1 = s_meta[lb + 1];
            const int p0 = m0.y, nn = m1.y - m0.y;
            if (nn > NNZB) return;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = tid + i * T;
                if (k < nn) {
                    c[s][i] = A.coef[p0 + k];
                    j[s][i] = A.indcol[p0 + k];
                }
            }
            const int row = m0.x + tid;
            if (row < m1.x) {
                ra[s] = A.ptrow[row] - p0;
                re[s] = A.ptrow[row + 1] - p0;
            }
        };

#pragma unroll
        for (int s = 0; s < D; s++)
            if (s < nb) issue(s, s);

        double xr = 0.0;       // prefetched ring entry for the NEXT block
        bool xr_valid = false; // (uniform)

        for (int g = 0; g < nb; g += D) {
#pragma unroll
            for (int s = 0; s < D; s++) {
                const int lb = g + s;
                if (lb < nb) {
                    const int4 m0 = s_meta[lb], m1 = s_meta[lb + 1];
                    const int r0 = m0.x, p0 = m0.y, r1 = m1.x, p1 = m1.y;
                    const int nn = p1 - p0;
                    const int myrow = r0 + tid;
                    if (nn > NNZB) {
                        // long row: serial exact path with the global gather
                        double sacc = 0.0;
                        for (int bs = p0; bs < p1; bs += NNZB) {
                            const int m = min(NNZB, p1 - bs);
                            __syncthreads();
                            for (int k = tid; k < m; k += T) {
                                s_c[sk(k)] = A.coef[bs + k];
                                s_x[sk(k)] = x[A.indcol[bs + k]];
                            }
                            __syncthreads();
                            if (tid == 0)
                                for (int k = 0; k < m; k++) sacc = fma(s_c[sk(k)], s_x[sk(k)], sacc);
                        }
                        if (tid == 0) y[A.rowmap ? A.rowmap[r0] : r0] = sacc;
                        xr_valid = false;
                        if (lb + D < nb) issue(lb + D, s);
                        continue;
                    }
                    // ---- A: bring the window over [cmin, cmax]
                    const RingPlan pl = ring_plan<RING>(st, m0.z, m0.w, nn);
                    if (pl.use) {
                        if (xr_valid) {
                            const int cc = pl.lo + tid;
                            if (cc < pl.hi) s_ring[ring_pos<RING>(cc, pl.next.base)] = xr;
                        } else {
                            for (int cc = pl.lo + tid; cc < pl.hi; cc += T)
                                s_ring[ring_pos<RING>(cc, pl.next.base)] = x[cc];
                        }
                        st = pl.next;
                    }
                    __syncthreads(); // B: ring visible; staging free again
                    // ---- C: gather + stage
#pragma unroll
                    for (int i = 0; i < PER; i++) {
                        const int k = tid + i * T;
                        if (k < nn) {
                            const double xv = pl.use ? s_ring[ring_pos<RING>(j[s][i], st.base)] : x[j[s][i]];
                            s_c[sk(k)] = c[s][i];
                            s_x[sk(k)] = xv;
                        }
                    }
                    const int mra = ra[s], mre = re[s];
                    // ---- D: refill this stage with block lb + D; request the next block's new columns
                    if (lb + D < nb) issue(lb + D, s);
                    xr_valid = false;
                    if (lb + 1 < nb) {
                        const int4 n0 = s_meta[lb + 1], n1 = s_meta[lb + 2];
                        const int nnn = n1.y - n0.y;
                        if (nnn <= NNZB) {
                            const RingPlan pn = ring_plan<RING>(st, n0.z, n0.w, nnn);
                            if (pn.use && pn.hi - pn.lo <= T) {
                                xr_valid = true;
                                const int cc = pn.lo + tid;
                                if (cc < pn.hi) xr = x[cc];
                            }
                        }
                    }
                    __syncthreads(); // E: staging complete
                    // ---- F: row chains
                    int a = mra, e = mre;
                    for (int r = myrow; r < r1; r += T) {
                        if (r != myrow) {
                            a = A.ptrow[r] - p0;
                            e = A.ptrow[r + 1] - p0;
                        }
                        y[A.rowmap ? A.rowmap[r] : r] = row_chain<8>(s_c, s_x, a, e);
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------
// E8: ring3 — the ring kernel written the way hipcc needs it to pipeline:
//   * every load of the steady state is UNCONDITIONAL (addresses clamped), column
//     ids are unsigned, raw ptrow values are kept (no arithmetic on a load result
//     near its issue), so the waitcnt pass can count: a stage's data is awaited
//     with vmcnt((D-1) * loads_per_stage), not vmcnt(0);
//   * the matrix stream AND the x columns entering the window run D blocks ahead
//     (same distance, because vector memory returns in order: consuming a young
//     load drains every older one);
//   * one workgroup = one run of <= MAXB row blocks whose metadata sits in LDS.
// ---------------------------------------------------------------------------
template <int T, int NNZB, int RING, int D, int MAXB>
__global__ __launch_bounds__(T) void spmv_csr_ring3(CsrView A, const int4* __restrict__ meta,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    int bpw)
{
    constexpr int PER = NNZB / T;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ int4 s_meta[MAXB + 2 * D + 2];
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int nb = min(A.nblk, b_begin + bpw) - b_begin; // <= MAXB by construction of the launch
    if (nb <= 0) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int nlast = A.n - 1, clast = A.ncols - 1;

    for (int i = tid; i < nb + 2 * D + 2; i += T) {
        int4 m = meta[min(b_begin + i, A.nblk)];
        if (i > nb) m = meta[min(b_begin + nb, A.nblk)]; // sentinels: empty blocks at the run's end
        if (i >= nb) { m.z = 0; m.w = -1; }
        s_meta[i] = m;
    }
    __syncthreads();

    RingState st_cur, st_pf;
    st_cur.wlo = st_cur.whi = st_cur.base = 0;
    st_cur.live = false;
    st_pf = st_cur;

    double c[D][PER];
    unsigned j[D][PER];
    int2 pr[D];
    double xr[D];

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_meta[lb], m1 = s_meta[lb + 1];
        const int p0 = m0.y, nn = m1.y - m0.y;
        const int last = max(min(nn, NNZB) - 1, 0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * T, last);
            c[s][i] = A.coef[p0 + k];
            j[s][i] = ucol[p0 + k];
        }
        const int row = min(min(m0.x + tid, max(m1.x - 1, m0.x)), nlast);
        pr[s] = make_int2(A.ptrow[row], A.ptrow[row + 1]);
        // x columns that will enter the window when this block becomes current
        int cc = 0;
        if (nn <= NNZB) {
            const RingPlan pn = ring_plan<RING>(st_pf, m0.z, m0.w, nn);
            if (pn.use) {
                cc = pn.lo + tid;
                st_pf = pn.next;
            }
        }
        xr[s] = x[min(cc, clast)];
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s; // lb >= nb: an empty sentinel block (keeps the load count per iteration fixed)
            {
                const int4 m0 = s_meta[lb], m1 = s_meta[lb + 1];
                const int r0 = m0.x, p0 = m0.y, r1 = m1.x, p1 = m1.y;
                const int nn = p1 - p0;
                const int myrow = r0 + tid;
                if (nn > NNZB) {
                    // long row: serial exact path with the global gather
                    double sacc = 0.0;
                    for (int bs = p0; bs < p1; bs += NNZB) {
                        const int m = min(NNZB, p1 - bs);
                        __syncthreads();
                        for (int k = tid; k < m; k += T) {
                            s_c[sk(k)] = A.coef[bs + k];
                            s_x[sk(k)] = x[A.indcol[bs + k]];
                        }
                        __syncthreads();
                        if (tid == 0)
                            for (int k = 0; k < m; k++) sacc = fma(s_c[sk(k)], s_x[sk(k)], sacc);
                    }
                    if (tid == 0) y[r0] = sacc;
                    issue(lb + D, s);
                    continue;
                }
                // ---- A: bring the window over [cmin, cmax] (same plan the prefetch made)
                const RingPlan pl = ring_plan<RING>(st_cur, m0.z, m0.w, nn);
                if (pl.use) {
                    if (pl.hi - pl.lo <= T) {
                        const int cc = pl.lo + tid;
                        if (cc < pl.hi) s_ring[ring_pos<RING>(cc, pl.next.base)] = xr[s];
                    } else {
                        for (int cc = pl.lo + tid; cc < pl.hi; cc += T)
                            s_ring[ring_pos<RING>(cc, pl.next.base)] = x[cc];
                    }
                    st_cur = pl.next;
                }
                __syncthreads(); // B: ring visible; staging free again
                // ---- C: gather + stage
                const int last = max(nn - 1, 0);
                // each branch finishes its own x values into LDS: a global-gather result
                // that stayed live across the join would make every later wait a full drain
                if (pl.use) {
                    double xv[PER];
#pragma unroll
                    for (int i = 0; i < PER; i++) xv[i] = s_ring[ring_pos<RING>((int)j[s][i], st_cur.base)];
#pragma unroll
                    for (int i = 0; i < PER; i++) s_x[sk(min(tid + i * T, last))] = xv[i];
                } else {
                    double xv[PER];
#pragma unroll
                    for (int i = 0; i < PER; i++) xv[i] = x[j[s][i]];
#pragma unroll
                    for (int i = 0; i < PER; i++) s_x[sk(min(tid + i * T, last))] = xv[i];
                }
#pragma unroll
                for (int i = 0; i < PER; i++) s_c[sk(min(tid + i * T, last))] = c[s][i];
                __builtin_amdgcn_sched_barrier(0);
                const int mra = pr[s].x - p0, mre = pr[s].y - p0;
                // ---- D: refill this stage with block lb + D
                issue(lb + D, s);
                __syncthreads(); // E: staging complete
                // ---- F: row chains
                if (nn > 0) {
                    if (myrow < r1) y[myrow] = row_chain<8>(s_c, s_x, mra, mre);
                    for (int r = myrow + T; r < r1; r += T) {
                        const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
                        y[r] = row_chain<8>(s_c, s_x, a, e);
                    }
                } else {
                    for (int r = myrow; r < r1; r += T) y[r] = 0.0;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------
// E9: ring4 — ring3 with every fallback moved OUT of the pipelined loop.
// hipcc structurises if/else into two predicated regions laid out one after the
// other, and its waitcnt pass is path-insensitive: a global gather in the "else"
// of the hot loop makes every later wait a full drain (vmcnt(0)), even when the
// branch is never taken.  So thread 0 first replays the ring plan over the run's
// metadata; a run with a block the ring cannot serve (span wider than the ring,
// a column behind the window, a row longer than a block) is handled by the plain
// per-block code, and the pipelined loop contains ring blocks only.
// ---------------------------------------------------------------------------
template <int T, int NNZB>
__device__ __forceinline__ void simple_block(const CsrView& A, const double* __restrict__ x,
                                             double* __restrict__ y, int r0, int p0, int r1, int p1,
                                             double* s_c, double* s_x)
{
    const int tid = threadIdx.x;
    const int nn = p1 - p0;
    __syncthreads();
    if (nn > NNZB) {
        double sacc = 0.0;
        for (int bs = p0; bs < p1; bs += NNZB) {
            const int m = min(NNZB, p1 - bs);
            __syncthreads();
            for (int k = tid; k < m; k += T) {
                s_c[sk(k)] = A.coef[bs + k];
                s_x[sk(k)] = x[A.indcol[bs + k]];
            }
            __syncthreads();
            if (tid == 0)
                for (int k = 0; k < m; k++) sacc = fma(s_c[sk(k)], s_x[sk(k)], sacc);
        }
        if (tid == 0) y[A.rowmap ? A.rowmap[r0] : r0] = sacc;
        return;
    }
    for (int k = tid; k < nn; k += T) {
        s_c[sk(k)] = A.coef[p0 + k];
        s_x[sk(k)] = x[A.indcol[p0 + k]];
    }
    __syncthreads();
    for (int r = r0 + tid; r < r1; r += T) {
        const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
        y[A.rowmap ? A.rowmap[r] : r] = row_chain<8>(s_c, s_x, a, e);
    }
}

template <int T, int NNZB, int RING, int D, int MAXB>
__global__ __launch_bounds__(T) void spmv_csr_ring4(CsrView A, const int4* __restrict__ meta,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    int bpw)
{
    constexpr int PER = NNZB / T;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ int4 s_meta[MAXB + 2 * D + 2];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int nb = min(A.nblk, b_begin + bpw) - b_begin; // <= MAXB by construction of the launch
    if (nb <= 0) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int nlast = A.n - 1, clast = A.ncols - 1;

    for (int i = tid; i < nb + 2 * D + 2; i += T) {
        int4 m = meta[min(b_begin + min(i, nb), A.nblk)];
        if (i >= nb) { m.z = 0; m.w = -1; } // sentinels: empty blocks behind the run
        s_meta[i] = m;
    }
    __syncthreads();
    if (tid == 0) { // replay the window plan; the run is "ring-able" iff every non-empty block is
        RingState st;
        st.wlo = st.whi = st.base = 0;
        st.live = false;
        int ok = 1;
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_meta[lb], m1 = s_meta[lb + 1];
            const int nn = m1.y - m0.y;
            if (nn == 0) continue;
            if (nn > NNZB) { ok = 0; break; }
            const RingPlan pl = ring_plan<RING>(st, m0.z, m0.w, nn);
            if (!pl.use) { ok = 0; break; }
            st = pl.next;
        }
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_meta[lb], m1 = s_meta[lb + 1];
            simple_block<T, NNZB>(A, x, y, m0.x, m0.y, m1.x, m1.y, s_c, s_x);
        }
        return;
    }

    RingState st_cur, st_pf;
    st_cur.wlo = st_cur.whi = st_cur.base = 0;
    st_cur.live = false;
    st_pf = st_cur;

    double c[D][PER];
    unsigned j[D][PER];
    int2 pr[D];
    double xr[D];

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_meta[lb], m1 = s_meta[lb + 1];
        const int p0 = m0.y, nn = m1.y - m0.y;
        const int last = max(nn - 1, 0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * T, last);
            c[s][i] = A.coef[p0 + k];
            j[s][i] = ucol[p0 + k];
        }
        const int row = min(min(m0.x + tid, max(m1.x - 1, m0.x)), nlast);
        pr[s] = make_int2(A.ptrow[row], A.ptrow[row + 1]);
        // x column that will enter the window when this block becomes current
        int cc = 0;
        const RingPlan pn = ring_plan<RING>(st_pf, m0.z, m0.w, nn);
        if (pn.use) {
            cc = pn.lo + tid;
            st_pf = pn.next;
        }
        xr[s] = x[min(cc, clast)];
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s; // lb >= nb: an empty sentinel block (keeps the load count per iteration fixed)
            const int4 m0 = s_meta[lb], m1 = s_meta[lb + 1];
            const int r0 = m0.x, p0 = m0.y, r1 = m1.x, p1 = m1.y;
            const int nn = p1 - p0;
            const int myrow = r0 + tid;
            // ---- A: bring the window over [cmin, cmax] (same plan the prefetch made)
            const RingPlan pl = ring_plan<RING>(st_cur, m0.z, m0.w, nn);
            if (pl.use) {
                if (pl.hi - pl.lo <= T) {
                    const int cc = pl.lo + tid;
                    if (cc < pl.hi) s_ring[ring_pos<RING>(cc, pl.next.base)] = xr[s];
                } else {
                    for (int cc = pl.lo + tid; cc < pl.hi; cc += T)
                        s_ring[ring_pos<RING>(cc, pl.next.base)] = x[cc];
                }
                st_cur = pl.next;
            }
            __syncthreads(); // B: ring visible; staging free again
            // ---- C: gather from the ring + stage
            const int last = max(nn - 1, 0);
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const unsigned pos = (unsigned)ring_pos<RING>((int)j[s][i], st_cur.base);
                xv[i] = s_ring[min(pos, (unsigned)(RING - 1))]; // clamp: sentinel blocks gather nothing meaningful
            }
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = sk(min(tid + i * T, last));
                s_c[k] = c[s][i];
                s_x[k] = xv[i];
            }
            const int2 prs = pr[s];
            // ---- D: refill this stage with block lb + D
            issue(lb + D, s);
            __syncthreads(); // E: staging complete
            // ---- F: row chains
            if (myrow < r1) y[myrow] = row_chain<8>(s_c, s_x, prs.x - p0, prs.y - p0);
            for (int r = myrow + T; r < r1; r += T) {
                const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
                y[r] = row_chain<8>(s_c, s_x, a, e);
            }
        }
    }
}


// ---------------------------------------------------------------------------
// E10: ring5 — ring4 with the window plan PRECOMPUTED per block (host side, once
// per matrix and launch shape) instead of replayed by every wave for every block:
//   plan[2b]   = {first row, first nnz, rows, nnz of the block}
//   plan[2b+1] = {first new column, number of new columns, ring base, flags}
// flags bit0: block is served from the ring; run_ok[run] says whether the whole
// run is (otherwise the run takes the plain per-block path).  The ring write for
// block b+1 moves into the reduce phase of block b (it only overwrites columns
// behind b+1's window, which b's gather — finished before the barrier — no longer
// reads), so a block costs: barrier, gather+stage+refill, barrier, reduce+ringwrite.
// ---------------------------------------------------------------------------
template <int T, int NNZB, int RING, int D, int MAXB>
__global__ __launch_bounds__(T) void spmv_csr_ring5(CsrView A, const int4* __restrict__ plan,
                                                    const int* __restrict__ run_ok,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    int bpw)
{
    constexpr int PER = NNZB / T;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ int4 s_plan[2 * (MAXB + 2 * D + 2)];
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int nb = min(A.nblk, b_begin + bpw) - b_begin; // <= MAXB by construction of the launch
    if (nb <= 0) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int nlast = A.n - 1, clast = A.ncols - 1;

    for (int i = tid; i < 2 * (nb + 2 * D + 2); i += T) {
        const int lb = i >> 1;
        int4 m;
        if (lb < nb) m = plan[2 * (b_begin + lb) + (i & 1)];
        else m = (i & 1) ? make_int4(0, 0, 0, 0) : make_int4(A.n, A.ptrow ? (int)0x7fffffff : 0, 0, 0);
        s_plan[i] = m;
    }
    __syncthreads();
    if (tid == 0) { // sentinel blocks start at the end of the run's last block (valid, padded addresses)
        const int4 l0 = s_plan[2 * (nb - 1)];
        for (int lb = nb; lb < nb + 2 * D + 2; lb++) s_plan[2 * lb] = make_int4(l0.x + l0.z, l0.y + l0.w, 0, 0);
    }
    __syncthreads();
    if (!run_ok[gw]) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_plan[2 * lb];
            simple_block<T, NNZB>(A, x, y, m0.x, m0.y, m0.x + m0.z, m0.y + m0.w, s_c, s_x);
        }
        return;
    }

    double c[D][PER];
    unsigned j[D][PER];
    int2 pr[D];
    double xr[D];

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
        const int p0 = m0.y;
        const int last = max(m0.w - 1, 0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * T, last);
            c[s][i] = A.coef[p0 + k];
            j[s][i] = ucol[p0 + k];
        }
        const int row = min(m0.x + min(tid, max(m0.z - 1, 0)), nlast);
        pr[s] = make_int2(A.ptrow[row], A.ptrow[row + 1]);
        xr[s] = x[min(m1.x + tid, clast)]; // the column this thread will put into the ring for block lb
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    { // ring content for block 0 (a whole window: more than T columns)
        const int4 q = s_plan[1];
        for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
    }

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s; // lb >= nb: an empty sentinel block (keeps the load count per iteration fixed)
            const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
            const int r0 = m0.x, p0 = m0.y, nrows = m0.z, nn = m0.w;
            const int base = m1.z;
            __syncthreads(); // ring holds block lb's window; staging is free again
            // ---- gather from the ring + stage
            const int last = max(nn - 1, 0);
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const unsigned pos = (unsigned)ring_pos<RING>((int)j[s][i], base);
                xv[i] = s_ring[min(pos, (unsigned)(RING - 1))]; // clamp: sentinel blocks gather nothing meaningful
            }
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = sk(min(tid + i * T, last));
                s_c[k] = c[s][i];
                s_x[k] = xv[i];
            }
            const int2 prs = pr[s];
            // ---- refill this stage with block lb + D
            issue(lb + D, s);
            __syncthreads(); // staging complete; nobody gathers block lb from the ring any more
            // ---- ring entries for block lb + 1 (prefetched D blocks ago into stage (s+1)%D)
            {
                const int4 q = s_plan[2 * (lb + 1) + 1];
                const double xn = xr[(s + 1) % D];
                if (q.y <= T) {
                    if (tid < q.y) s_ring[ring_pos<RING>(q.x + tid, q.z)] = xn;
                } else { // window restart inside a run (a jump in the column range): synchronous refill
                    for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
                }
            }
            // ---- row chains
            const int myrow = r0 + tid;
            if (tid < nrows) y[myrow] = row_chain<8>(s_c, s_x, prs.x - p0, prs.y - p0);
            for (int r = myrow + T; r < r0 + nrows; r += T) {
                const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
                y[r] = row_chain<8>(s_c, s_x, a, e);
            }
        }
    }
}


// ablation copy of ring5 (timing only; results invalid when ABL != 0)
template <int T, int NNZB, int RING, int D, int MAXB, int ABL>
__global__ __launch_bounds__(T) void spmv_csr_ring5a(CsrView A, const int4* __restrict__ plan,
                                                    const int* __restrict__ run_ok,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    int bpw, const unsigned short* __restrict__ slots = nullptr)
{
    constexpr int PER = NNZB / T;
    typedef unsigned short SlotVec __attribute__((ext_vector_type(PER)));
    constexpr bool C16 = (ABL & 2048) != 0;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ int4 s_plan[2 * (MAXB + 2 * D + 2)];
    constexpr int YB = (ABL & 512) ? 6144 : 1; // rows of y parked in LDS between flushes
    __shared__ double s_yb[YB];
    int yb_cnt = 0, yb_row0 = 0;
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int nb = min(A.nblk, b_begin + bpw) - b_begin; // <= MAXB by construction of the launch
    if (nb <= 0) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int nlast = A.n - 1, clast = A.ncols - 1;

    for (int i = tid; i < 2 * (nb + 2 * D + 2); i += T) {
        const int lb = i >> 1;
        int4 m;
        if (lb < nb) m = plan[2 * (b_begin + lb) + (i & 1)];
        else m = (i & 1) ? make_int4(0, 0, 0, 0) : make_int4(A.n, A.ptrow ? (int)0x7fffffff : 0, 0, 0);
        s_plan[i] = m;
    }
    __syncthreads();
    if (tid == 0) { // sentinel blocks start at the end of the run's last block (valid, padded addresses)
        const int4 l0 = s_plan[2 * (nb - 1)];
        for (int lb = nb; lb < nb + 2 * D + 2; lb++) s_plan[2 * lb] = make_int4(l0.x + l0.z, l0.y + l0.w, 0, 0);
    }
    __syncthreads();
    if (!run_ok[gw]) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_plan[2 * lb];
            simple_block<T, NNZB>(A, x, y, m0.x, m0.y, m0.x + m0.z, m0.y + m0.w, s_c, s_x);
        }
        return;
    }

    double c[D][PER];
    unsigned j[D][C16 ? 1 : PER];
    SlotVec sl[D];
    int2 pr[D];
    double xr[D];
    double ysum = 0.0;

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
        const int p0 = m0.y;
        const int last = max(m0.w - 1, 0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * T, last);
            if (ABL & 1024) { // non-temporal stream loads: do not let the matrix displace x / y / ptrow from L2 / Infinity Cache
                c[s][i] = __builtin_nontemporal_load(&A.coef[p0 + k]);
                if (!C16) j[s][i] = __builtin_nontemporal_load(&ucol[p0 + k]);
            } else {
                c[s][i] = A.coef[p0 + k];
                if (!C16) j[s][i] = ucol[p0 + k];
            }
        }
        if (C16) {
            const SlotVec* sp = &reinterpret_cast<const SlotVec*>(slots)[(size_t)min(b_begin + lb, A.nblk - 1) * T + tid];
            if (ABL & 4096) sl[s] = __builtin_nontemporal_load(sp); else sl[s] = *sp;
        }
        const int row = min(m0.x + min(tid, max(m0.z - 1, 0)), nlast);
        if (ABL & 32) pr[s] = make_int2(p0, p0 + 15);
        else if (ABL & 16384) pr[s] = make_int2(__builtin_nontemporal_load(&A.ptrow[row]), __builtin_nontemporal_load(&A.ptrow[row + 1]));
        else pr[s] = make_int2(A.ptrow[row], A.ptrow[row + 1]);
        if (ABL & 16) xr[s] = 0.0; else xr[s] = x[min(m1.x + tid, clast)];
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    { // ring content for block 0 (a whole window: more than T columns)
        const int4 q = s_plan[1];
        for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
    }

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s; // lb >= nb: an empty sentinel block (keeps the load count per iteration fixed)
            const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
            const int r0 = m0.x, p0 = m0.y, nrows = m0.z, nn = m0.w;
            const int base = m1.z;
            if (!(ABL & 8)) __syncthreads(); // ring holds block lb's window; staging is free again
            // ---- gather from the ring + stage
            const int last = max(nn - 1, 0);
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const unsigned pos = C16 ? (unsigned)sl[s][i] : (unsigned)ring_pos<RING>((int)j[s][i], base);
                if (ABL & 2) xv[i] = (double)pos;
                else xv[i] = s_ring[min(pos, (unsigned)(RING - 1))];
            }
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = sk(min(tid + i * T, last));
                if (ABL & 4) { asm volatile("" ::"v"(c[s][i]), "v"(xv[i]), "v"(k)); }
                else { s_c[k] = c[s][i]; s_x[k] = xv[i]; }
            }
            const int2 prs = pr[s];
            // ---- refill this stage with block lb + D
            issue(lb + D, s);
            if (!(ABL & 8)) __syncthreads(); // staging complete; nobody gathers block lb from the ring any more
            // ---- ring entries for block lb + 1 (prefetched D blocks ago into stage (s+1)%D)
            {
                const int4 q = s_plan[2 * (lb + 1) + 1];
                const double xn = xr[(s + 1) % D];
                if (q.y <= T) {
                    if (tid < q.y) s_ring[ring_pos<RING>(q.x + tid, q.z)] = xn;
                } else { // window restart inside a run (a jump in the column range): synchronous refill
                    for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
                }
            }
            // ---- row chains
            const int myrow = r0 + tid;
            if (ABL & 64) { if (tid < nrows) ysum += (ABL & 1) ? s_c[sk(min(max(prs.x - p0, 0), NNZB - 1))] + (double)(prs.y - p0) : row_chain<8>(s_c, s_x, prs.x - p0, prs.y - p0); }
            else if (ABL & 1) { if (tid < nrows) y[myrow] = s_c[sk(min(max(prs.x - p0, 0), NNZB - 1))] + (double)(prs.y - p0); }
            else if (ABL & 512) { // park the block's y values in LDS; flush every few blocks with all threads
                if (yb_cnt == 0) yb_row0 = r0;
                if (tid < nrows) s_yb[yb_cnt + tid] = row_chain<8>(s_c, s_x, prs.x - p0, prs.y - p0);
                yb_cnt += nrows;
                const int nxt = s_plan[2 * (lb + 1)].z;
                if (yb_cnt + nxt > YB || lb + 1 >= nb) {
                    __syncthreads();
                    for (int r = tid; r < yb_cnt; r += T) y[yb_row0 + r] = s_yb[r];
                    yb_cnt = 0;
                }
            }
            else if (ABL & 128) { // unconditional store: rowless threads write a scratch slot behind y
                const double v = row_chain<8>(s_c, s_x, tid < nrows ? prs.x - p0 : 0, tid < nrows ? prs.y - p0 : 0);
                double* dst = tid < nrows ? &y[myrow] : &y[(size_t)A.n + (size_t)blockIdx.x * T + tid];
                *dst = v;
            }
            else if (ABL & 8192) { if (tid < nrows) __builtin_nontemporal_store(row_chain<8>(s_c, s_x, prs.x - p0, prs.y - p0), &y[myrow]); }
            else if (tid < nrows) y[myrow] = row_chain<8>(s_c, s_x, prs.x - p0, prs.y - p0);
            for (int r = myrow + T; r < r0 + nrows; r += T) {
                const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
                y[r] = row_chain<8>(s_c, s_x, a, e);
            }
        }
    }
    if (ABL & 64) y[b_begin * 16 + tid] = ysum;
}



// diagnostic copy of ring5 with s_memtime stamps (shares of the block loop per phase)
template <int T, int NNZB, int RING, int D, int MAXB, bool C16NT = false>
__global__ __launch_bounds__(T) void spmv_csr_ring5t(CsrView A, const int4* __restrict__ plan, unsigned long long* __restrict__ prof,
                                                    const int* __restrict__ run_ok,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    int bpw, const unsigned short* __restrict__ slots = nullptr)
{
    const long long t_start = clock64();
    constexpr int PER = NNZB / T;
    typedef unsigned short SlotVec __attribute__((ext_vector_type(PER)));
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ int4 s_plan[2 * (MAXB + 2 * D + 2)];
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int nb = min(A.nblk, b_begin + bpw) - b_begin; // <= MAXB by construction of the launch
    if (nb <= 0) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int nlast = A.n - 1, clast = A.ncols - 1;

    for (int i = tid; i < 2 * (nb + 2 * D + 2); i += T) {
        const int lb = i >> 1;
        int4 m;
        if (lb < nb) m = plan[2 * (b_begin + lb) + (i & 1)];
        else m = (i & 1) ? make_int4(0, 0, 0, 0) : make_int4(A.n, A.ptrow ? (int)0x7fffffff : 0, 0, 0);
        s_plan[i] = m;
    }
    __syncthreads();
    if (tid == 0) { // sentinel blocks start at the end of the run's last block (valid, padded addresses)
        const int4 l0 = s_plan[2 * (nb - 1)];
        for (int lb = nb; lb < nb + 2 * D + 2; lb++) s_plan[2 * lb] = make_int4(l0.x + l0.z, l0.y + l0.w, 0, 0);
    }
    __syncthreads();
    if (!run_ok[gw]) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_plan[2 * lb];
            simple_block<T, NNZB>(A, x, y, m0.x, m0.y, m0.x + m0.z, m0.y + m0.w, s_c, s_x);
        }
        return;
    }

    double c[D][PER];
    unsigned j[D][C16NT ? 1 : PER];
    SlotVec sl[D];
    int2 pr[D];
    double xr[D];
    long long acc[6] = {0, 0, 0, 0, 0, 0};
    long long t0 = clock64(), t1;
#define STAMP(i) t1 = clock64(); acc[i] += t1 - t0; t0 = t1;

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
        const int p0 = m0.y;
        const int last = max(m0.w - 1, 0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * T, last);
            if (C16NT) c[s][i] = __builtin_nontemporal_load(&A.coef[p0 + k]);
            else {
                c[s][i] = A.coef[p0 + k];
                j[s][i] = ucol[p0 + k];
            }
        }
        if (C16NT) sl[s] = reinterpret_cast<const SlotVec*>(slots)[(size_t)min(b_begin + lb, A.nblk - 1) * T + tid];
        const int row = min(m0.x + min(tid, max(m0.z - 1, 0)), nlast);
        pr[s] = make_int2(A.ptrow[row], A.ptrow[row + 1]);
        xr[s] = x[min(m1.x + tid, clast)]; // the column this thread will put into the ring for block lb
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    { // ring content for block 0 (a whole window: more than T columns)
        const int4 q = s_plan[1];
        for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
    }

    t0 = clock64();
    acc[5] = t0 - t_start; // launch-to-loop: plan load, barriers, prologue issue, first window
    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s; // lb >= nb: an empty sentinel block (keeps the load count per iteration fixed)
            const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
            const int r0 = m0.x, p0 = m0.y, nrows = m0.z, nn = m0.w;
            const int base = m1.z;
            STAMP(4)
            __syncthreads(); // ring holds block lb's window; staging is free again
            STAMP(0)
            // ---- gather from the ring + stage
            const int last = max(nn - 1, 0);
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const unsigned pos = C16NT ? (unsigned)sl[s][i] : (unsigned)ring_pos<RING>((int)j[s][i], base);
                xv[i] = s_ring[min(pos, (unsigned)(RING - 1))]; // clamp: sentinel blocks gather nothing meaningful
            }
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = sk(min(tid + i * T, last));
                s_c[k] = c[s][i];
                s_x[k] = xv[i];
            }
            const int2 prs = pr[s];
            STAMP(1)
            // ---- refill this stage with block lb + D
            issue(lb + D, s);
            STAMP(2)
            __syncthreads(); // staging complete; nobody gathers block lb from the ring any more
            STAMP(3)
            // ---- ring entries for block lb + 1 (prefetched D blocks ago into stage (s+1)%D)
            {
                const int4 q = s_plan[2 * (lb + 1) + 1];
                const double xn = xr[(s + 1) % D];
                if (q.y <= T) {
                    if (tid < q.y) s_ring[ring_pos<RING>(q.x + tid, q.z)] = xn;
                } else { // window restart inside a run (a jump in the column range): synchronous refill
                    for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
                }
            }
            // ---- row chains
            const int myrow = r0 + tid;
            if (tid < nrows) y[myrow] = row_chain<8>(s_c, s_x, prs.x - p0, prs.y - p0);
            for (int r = myrow + T; r < r0 + nrows; r += T) {
                const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
                y[r] = row_chain<8>(s_c, s_x, a, e);
            }
        }
    }
    STAMP(4)
    if ((tid & 63) == 0) {
        const int w = (tid >> 6) == 0 ? 0 : ((tid >> 6) == (T / 64 - 1) ? 1 : 2);
        if (w < 2)
            for (int i = 0; i < 6; i++) atomicAdd(&prof[w * 8 + i], (unsigned long long)acc[i]);
    }
#undef STAMP
}



// ---------------------------------------------------------------------------
// E11: ring6 — ring5 with trimmed index arithmetic:
//   * staging slots sk(tid + i*T) (as LDS byte offsets) precomputed once per thread;
//     per block only the clamp to the block's last nonzero remains (cmp + select);
//   * sentinel blocks point at the run's LAST block, whose columns are inside the
//     final window, so the gather needs no range clamp;
//   * U (operands fetched per batch in the row chain) is a template parameter.
// ---------------------------------------------------------------------------
template <int T, int NNZB, int RING, int D, int MAXB, int U>
__global__ __launch_bounds__(T) void spmv_csr_ring6(CsrView A, const int4* __restrict__ plan,
                                                    const int* __restrict__ run_ok,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    int bpw)
{
    constexpr int PER = NNZB / T;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ int4 s_plan[2 * (MAXB + 2 * D + 2)];
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int nb = min(A.nblk, b_begin + bpw) - b_begin;
    if (nb <= 0) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int nlast = A.n - 1, clast = A.ncols - 1;

    for (int i = tid; i < 2 * nb; i += T) s_plan[i] = plan[2 * b_begin + i];
    __syncthreads();
    {
        const int4 l0 = s_plan[2 * (nb - 1)], l1 = s_plan[2 * (nb - 1) + 1];
        // sentinels re-read the first nonzero of the run's last block: a column of the final window
        const int4 sent0 = make_int4(l0.x + l0.z, l0.y, 0, 0);
        const int4 sent1 = make_int4(0, 0, l1.z, 0);
        for (int i = tid; i < 2 * D + 2; i += T) {
            s_plan[2 * (nb + i)] = sent0;
            s_plan[2 * (nb + i) + 1] = sent1;
        }
    }
    __syncthreads();
    if (!run_ok[gw]) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_plan[2 * lb];
            simple_block<T, NNZB>(A, x, y, m0.x, m0.y, m0.x + m0.z, m0.y + m0.w, s_c, s_x);
        }
        return;
    }

    int kk[PER], ksk[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        kk[i] = tid + i * T;
        ksk[i] = sk(kk[i]);
    }
    double c[D][PER];
    unsigned j[D][PER];
    int2 pr[D];
    double xr[D];

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
        const int p0 = m0.y;
        const int last = max(m0.w - 1, 0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(kk[i], last);
            c[s][i] = A.coef[p0 + k];
            j[s][i] = ucol[p0 + k];
        }
        const int row = min(m0.x + min(tid, max(m0.z - 1, 0)), nlast);
        pr[s] = make_int2(A.ptrow[row], A.ptrow[row + 1]);
        xr[s] = x[min(m1.x + tid, clast)];
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    {
        const int4 q = s_plan[1];
        for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
    }

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s;
            const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
            const int r0 = m0.x, p0 = m0.y, nrows = m0.z, nn = m0.w;
            const int base = m1.z;
            __syncthreads();
            const int last = max(nn - 1, 0);
            const int lsk = sk(last);
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) xv[i] = s_ring[ring_pos<RING>((int)j[s][i], base)];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = kk[i] <= last ? ksk[i] : lsk;
                s_c[k] = c[s][i];
                s_x[k] = xv[i];
            }
            const int2 prs = pr[s];
            issue(lb + D, s);
            __syncthreads();
            {
                const int4 q = s_plan[2 * (lb + 1) + 1];
                const double xn = xr[(s + 1) % D];
                if (q.y <= T) {
                    if (tid < q.y) s_ring[ring_pos<RING>(q.x + tid, q.z)] = xn;
                } else {
                    for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
                }
            }
            const int myrow = r0 + tid;
            if (tid < nrows) y[myrow] = row_chain<U>(s_c, s_x, prs.x - p0, prs.y - p0);
            for (int r = myrow + T; r < r0 + nrows; r += T) {
                const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
                y[r] = row_chain<U>(s_c, s_x, a, e);
            }
        }
    }
}


// ---------------------------------------------------------------------------
// E12: ring7 — ring5 + a WRITER WAVE.  Measured: the y stores, although only 4 % of
// the bytes, cost 10-20 % when they sit in the same in-order vector-memory queue as
// the prefetched matrix stream (ablation: identical kernel without the stores runs
// at the pure streaming rate).  So the T worker threads never store: row results go
// to an LDS array, and one extra wave (threads T..T+63), which issues no loads,
// writes block b-1's results to y while the workers gather block b.
// ---------------------------------------------------------------------------
template <int T, int NNZB, int RING, int D, int MAXB, int MAXROWS>
__global__ __launch_bounds__(T + 64) void spmv_csr_ring7(CsrView A, const int4* __restrict__ plan,
                                                         const int* __restrict__ run_ok,
                                                         const double* __restrict__ x, double* __restrict__ y,
                                                         int bpw)
{
    constexpr int PER = NNZB / T;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    constexpr int TW = T + 64;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ double s_yv[MAXROWS];
    __shared__ int4 s_plan[2 * (MAXB + 2 * D + 2)];
    const int tid = threadIdx.x;
    const int gw = (blockIdx.x & (kNXCD - 1)) * (gridDim.x / kNXCD) + (blockIdx.x >> 3);
    const int b_begin = gw * bpw;
    const int nb = min(A.nblk, b_begin + bpw) - b_begin;
    if (nb <= 0) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int nlast = A.n - 1, clast = A.ncols - 1;

    for (int i = tid; i < 2 * nb; i += TW) s_plan[i] = plan[2 * b_begin + i];
    __syncthreads();
    {
        const int4 l0 = s_plan[2 * (nb - 1)];
        const int4 sent = make_int4(l0.x + l0.z, l0.y + l0.w, 0, 0);
        for (int i = tid; i < 2 * D + 2; i += TW) {
            s_plan[2 * (nb + i)] = sent;
            s_plan[2 * (nb + i) + 1] = make_int4(0, 0, 0, 0);
        }
    }
    __syncthreads();
    if (!run_ok[gw]) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_plan[2 * lb];
            simple_block<TW, NNZB>(A, x, y, m0.x, m0.y, m0.x + m0.z, m0.y + m0.w, s_c, s_x);
        }
        return;
    }
    const int nbp = ((nb + D - 1) / D) * D; // blocks incl. sentinels, same for workers and writer

    if (tid >= T) {
        // ------------------------------------------------ writer wave
        const int lane = tid - T;
        for (int lb = 0; lb < nbp; lb++) {
            __syncthreads(); // (1) of block lb: s_yv holds block lb-1
            if (lb > 0) {
                const int4 m0 = s_plan[2 * (lb - 1)];
                for (int r = lane; r < m0.z; r += 64) y[m0.x + r] = s_yv[r];
            }
            __syncthreads(); // (2) of block lb
        }
        __syncthreads(); // final: s_yv holds the last block
        {
            const int4 m0 = s_plan[2 * (nbp - 1)];
            for (int r = lane; r < m0.z; r += 64) y[m0.x + r] = s_yv[r];
        }
        return;
    }

    // ---------------------------------------------------- workers
    double c[D][PER];
    unsigned j[D][PER];
    int2 pr[D];
    double xr[D];

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
        const int p0 = m0.y;
        const int last = max(m0.w - 1, 0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * T, last);
            c[s][i] = A.coef[p0 + k];
            j[s][i] = ucol[p0 + k];
        }
        const int row = min(m0.x + min(tid, max(m0.z - 1, 0)), nlast);
        pr[s] = make_int2(A.ptrow[row], A.ptrow[row + 1]);
        xr[s] = x[min(m1.x + tid, clast)];
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    {
        const int4 q = s_plan[1];
        for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
    }

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s;
            const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
            const int p0 = m0.y, nrows = m0.z, nn = m0.w;
            const int base = m1.z;
            __syncthreads(); // (1)
            const int last = max(nn - 1, 0);
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const unsigned pos = (unsigned)ring_pos<RING>((int)j[s][i], base);
                xv[i] = s_ring[min(pos, (unsigned)(RING - 1))];
            }
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = sk(min(tid + i * T, last));
                s_c[k] = c[s][i];
                s_x[k] = xv[i];
            }
            const int2 prs = pr[s];
            issue(lb + D, s);
            __syncthreads(); // (2) staging complete; the writer is done with s_yv of block lb-1
            {
                const int4 q = s_plan[2 * (lb + 1) + 1];
                const double xn = xr[(s + 1) % D];
                if (q.y <= T) {
                    if (tid < q.y) s_ring[ring_pos<RING>(q.x + tid, q.z)] = xn;
                } else {
                    for (int cc = q.x + tid; cc < q.x + q.y; cc += T) s_ring[ring_pos<RING>(cc, q.z)] = x[cc];
                }
            }
            if (tid < nrows) s_yv[tid] = row_chain<8>(s_c, s_x, prs.x - p0, prs.y - p0);
            for (int r = tid + T; r < nrows; r += T) {
                const int a = A.ptrow[m0.x + r] - p0, e = A.ptrow[m0.x + r + 1] - p0;
                s_yv[r] = row_chain<8>(s_c, s_x, a, e);
            }
        }
    }
    __syncthreads(); // final: hand the last block's results to the writer
}

} // namespace mi355

inline void add_experimental_variants(std::vector<Variant>& vars, int n, const int* d_ptrow, const int* d_indcol,
                                      const double* d_coef, const double* d_x, double* d_y, mi355::CsrView V1k,
                                      mi355::CsrView V2k, mi355::CsrView V4k, const int4* M1k = nullptr,
                                      const int4* M2k = nullptr, const int4* M4k = nullptr,
                                      std::function<void(int, int, int, int, const int4**, const int**, int*, int*)> make_plan = nullptr, int* stagger = nullptr, int* row_align = nullptr)
{
    using namespace mi355;
    (void)n; (void)d_ptrow; (void)d_indcol; (void)d_coef;
    auto grid8 = [](int nblk) { return dim3(kNXCD * ((nblk + kNXCD - 1) / kNXCD)); };
    vars.push_back({"E1 stream_v2<2048> (16B loads)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_v2<2048>), grid8(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    vars.push_back({"E1 stream_v2<1024> (16B loads)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_v2<1024>), grid8(V1k.nblk), dim3(kWG), 0, s, V1k, d_x, d_y); }});
    vars.push_back({"E1 stream_v2<4096> (16B loads)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_v2<4096>), grid8(V4k.nblk), dim3(kWG), 0, s, V4k, d_x, d_y); }});
    vars.push_back({"E2 stream<2048> no XCD remap", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_noremap<2048>), dim3(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    vars.push_back({"E3 stream<2048> NO GATHER (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_nogather<2048>), grid8(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    {
        auto ring_launch = [=](auto kern, CsrView V, int threads, int wgs) {
            const int bpw = (V.nblk + wgs - 1) / wgs;
            return [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, s, V, d_x, d_y, bpw); };
        };
        vars.push_back({"E4 ring<512,2048,5120> 512 WGs", ring_launch(spmv_csr_ring<512, 2048, 5120>, V2k, 512, 512)});
        vars.push_back({"E4 ring<512,2048,5120> 1024 WGs", ring_launch(spmv_csr_ring<512, 2048, 5120>, V2k, 512, 1024)});
        vars.push_back({"E4 ring<256,2048,5120> 512 WGs", ring_launch(spmv_csr_ring<256, 2048, 5120>, V2k, 256, 512)});
        vars.push_back({"E4 ring<256,1024,4608> 768 WGs", ring_launch(spmv_csr_ring<256, 1024, 4608>, V1k, 256, 768)});
        vars.push_back({"E4 ring<1024,4096,8192> 256 WGs", ring_launch(spmv_csr_ring<1024, 4096, 8192>, V4k, 1024, 256)});
        vars.push_back({"E4 ring<512,2048,5120> 2048 WGs", ring_launch(spmv_csr_ring<512, 2048, 5120>, V2k, 512, 2048)});
    }
    vars.push_back({"E5 stream3<2048> batched-8 reduce", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream3<2048, 1>), grid8(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    vars.push_back({"E5 stream3<2048> batched-16 reduce", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream3<2048, 2>), grid8(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    vars.push_back({"E5 stream3<4096> batched-8 reduce", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream3<4096, 1>), grid8(V4k.nblk), dim3(kWG), 0, s, V4k, d_x, d_y); }});
    vars.push_back({"E5 stream3<4096> batched-16 reduce", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream3<4096, 2>), grid8(V4k.nblk), dim3(kWG), 0, s, V4k, d_x, d_y); }});
    vars.push_back({"E5 stream3<1024> batched-16 reduce", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream3<1024, 2>), grid8(V1k.nblk), dim3(kWG), 0, s, V1k, d_x, d_y); }});
    vars.push_back({"E5 stream3<2048> NO REDUCE (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream3<2048, 3>), grid8(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    vars.push_back({"E6 stream+gather, no LDS (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_nolds<2048>), grid8(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    vars.push_back({"E6 stream+gather<4096>, no LDS (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_nolds<4096>), grid8(V4k.nblk), dim3(kWG), 0, s, V4k, d_x, d_y); }});
    if (M2k) {
        auto ring2_launch = [=](auto kern, CsrView V, const int4* M, int threads, int wgs) {
            const int bpw = (V.nblk + wgs - 1) / wgs;
            return [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, s, V, M, d_x, d_y, bpw); };
        };
        vars.push_back({"E7 ring2<512,2048,5120,D2> 512 WGs", ring2_launch(spmv_csr_ring2<512, 2048, 5120, 2>, V2k, M2k, 512, 512)});
        vars.push_back({"E7 ring2<512,2048,5120,D3> 512 WGs", ring2_launch(spmv_csr_ring2<512, 2048, 5120, 3>, V2k, M2k, 512, 512)});
        vars.push_back({"E7 ring2<512,2048,5120,D4> 512 WGs", ring2_launch(spmv_csr_ring2<512, 2048, 5120, 4>, V2k, M2k, 512, 512)});
        vars.push_back({"E7 ring2<512,2048,5120,D3> 1024 WGs", ring2_launch(spmv_csr_ring2<512, 2048, 5120, 3>, V2k, M2k, 512, 1024)});
        vars.push_back({"E7 ring2<256,1024,4608,D4> 768 WGs", ring2_launch(spmv_csr_ring2<256, 1024, 4608, 4>, V1k, M1k, 256, 768)});
        vars.push_back({"E7 ring2<256,2048,5120,D2> 512 WGs", ring2_launch(spmv_csr_ring2<256, 2048, 5120, 2>, V2k, M2k, 256, 512)});
        vars.push_back({"E7 ring2<1024,4096,5120,D2> 256 WGs", ring2_launch(spmv_csr_ring2<1024, 4096, 5120, 2>, V4k, M4k, 1024, 256)});
        vars.push_back({"E7 ring2<1024,4096,5120,D3> 256 WGs", ring2_launch(spmv_csr_ring2<1024, 4096, 5120, 3>, V4k, M4k, 1024, 256)});
    }
    if (M2k) {
        auto ring3_launch = [=](auto kern, CsrView V, const int4* M, int threads, int maxb, int min_wgs) {
            int wgs = std::max(min_wgs, (V.nblk + maxb - 1) / maxb);
            wgs = ((wgs + 7) / 8) * 8;
            const int bpw = (V.nblk + wgs - 1) / wgs;
            return [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, s, V, M, d_x, d_y, bpw); };
        };
        vars.push_back({"E8 ring3<512,2048,5120,D2> 512 WGs", ring3_launch(spmv_csr_ring3<512, 2048, 5120, 2, 160>, V2k, M2k, 512, 160, 512)});
        vars.push_back({"E8 ring3<512,2048,5120,D3> 512 WGs", ring3_launch(spmv_csr_ring3<512, 2048, 5120, 3, 160>, V2k, M2k, 512, 160, 512)});
        vars.push_back({"E8 ring3<512,2048,5120,D4> 512 WGs", ring3_launch(spmv_csr_ring3<512, 2048, 5120, 4, 160>, V2k, M2k, 512, 160, 512)});
        vars.push_back({"E8 ring3<512,2048,5120,D3> 1024 WGs", ring3_launch(spmv_csr_ring3<512, 2048, 5120, 3, 160>, V2k, M2k, 512, 160, 1024)});
        vars.push_back({"E8 ring3<256,1024,4608,D4> 768 WGs", ring3_launch(spmv_csr_ring3<256, 1024, 4608, 4, 160>, V1k, M1k, 256, 160, 768)});
        vars.push_back({"E8 ring3<1024,4096,5120,D2> 256 WGs", ring3_launch(spmv_csr_ring3<1024, 4096, 5120, 2, 160>, V4k, M4k, 1024, 160, 256)});
        vars.push_back({"E8 ring3<1024,4096,5120,D3> 256 WGs", ring3_launch(spmv_csr_ring3<1024, 4096, 5120, 3, 160>, V4k, M4k, 1024, 160, 256)});
    }
    if (M2k) {
        auto ring4_launch = [=](auto kern, CsrView V, const int4* M, int threads, int maxb, int min_wgs) {
            int wgs = std::max(min_wgs, (V.nblk + maxb - 1) / maxb);
            wgs = ((wgs + 7) / 8) * 8;
            const int bpw = (V.nblk + wgs - 1) / wgs;
            return [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, s, V, M, d_x, d_y, bpw); };
        };
        vars.push_back({"E9 ring4<512,2048,5120,D2> 512 WGs", ring4_launch(spmv_csr_ring4<512, 2048, 5120, 2, 160>, V2k, M2k, 512, 160, 512)});
        vars.push_back({"E9 ring4<512,2048,5120,D3> 512 WGs", ring4_launch(spmv_csr_ring4<512, 2048, 5120, 3, 160>, V2k, M2k, 512, 160, 512)});
        vars.push_back({"E9 ring4<512,2048,5120,D4> 512 WGs", ring4_launch(spmv_csr_ring4<512, 2048, 5120, 4, 160>, V2k, M2k, 512, 160, 512)});
        vars.push_back({"E9 ring4<512,2048,5120,D3> 1024 WGs", ring4_launch(spmv_csr_ring4<512, 2048, 5120, 3, 160>, V2k, M2k, 512, 160, 1024)});
        vars.push_back({"E9 ring4<256,1024,4608,D4> 768 WGs", ring4_launch(spmv_csr_ring4<256, 1024, 4608, 4, 160>, V1k, M1k, 256, 160, 768)});
        vars.push_back({"E9 ring4<1024,4096,5120,D2> 256 WGs", ring4_launch(spmv_csr_ring4<1024, 4096, 5120, 2, 160>, V4k, M4k, 1024, 160, 256)});
        vars.push_back({"E9 ring4<256,2048,5120,D2> 512 WGs", ring4_launch(spmv_csr_ring4<256, 2048, 5120, 2, 160>, V2k, M2k, 256, 160, 512)});
        vars.push_back({"E9 ring4<512,2048,5120,D2> 2048 WGs", ring4_launch(spmv_csr_ring4<512, 2048, 5120, 2, 160>, V2k, M2k, 512, 160, 2048)});
        vars.push_back({"E9 ring4<512,2048,5120,D2> 4096 WGs", ring4_launch(spmv_csr_ring4<512, 2048, 5120, 2, 160>, V2k, M2k, 512, 160, 4096)});
        vars.push_back({"E9 ring4<512,2048,5120,D2> 8192 WGs", ring4_launch(spmv_csr_ring4<512, 2048, 5120, 2, 160>, V2k, M2k, 512, 160, 8192)});
    }
    if (make_plan) {
        // make_plan(nnzb_table(1024|2048|4096), ring, maxb, min_wgs) -> plan, run_ok, wgs, bpw
        auto ring5_launch = [=](auto kern, CsrView V, int tab, int ring, int threads, int maxb, int min_wgs) {
            const int4* P; const int* OK; int wgs, bpw;
            make_plan(tab, ring, maxb, min_wgs, &P, &OK, &wgs, &bpw);
            return [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, s, V, P, OK, d_x, d_y, bpw); };
        };
        auto ring5_launch_rr = [=](auto kern, CsrView V, int tab, int ring, int threads, int wgsreq, int R) {
            const int4* P; const int* OK; int wgs, bpw;
            g_plan_rr = R;
            make_plan(tab, ring, 1 << 20, wgsreq, &P, &OK, &wgs, &bpw);
            g_plan_rr = 0;
            V.nblk = g_last_plan_nblk;
            return [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, s, V, P, OK, d_x, d_y, bpw); };
        };
        for (int R : {4, 8, 12, 24}) {
            vars.push_back({"E13 ring5<512,4096,5120,D2> 256 WGs rr R=" + std::to_string(R), ring5_launch_rr(spmv_csr_ring5<512, 4096, 5120, 2, 160>, V4k, 4096, 5120, 512, 256, R)});
            vars.push_back({"E13 ring5<512,2048,5120,D2> 512 WGs rr R=" + std::to_string(R), ring5_launch_rr(spmv_csr_ring5<512, 2048, 5120, 2, 160>, V2k, 2048, 5120, 512, 512, R)});
        }
        vars.push_back({"E10 ring5<512,2048,5120,D2> 512 WGs", ring5_launch(spmv_csr_ring5<512, 2048, 5120, 2, 160>, V2k, 2048, 5120, 512, 160, 512)});
        vars.push_back({"E10 ring5<512,2048,5120,D3> 512 WGs", ring5_launch(spmv_csr_ring5<512, 2048, 5120, 3, 160>, V2k, 2048, 5120, 512, 160, 512)});
        vars.push_back({"E10 ring5<256,2048,5120,D2> 512 WGs", ring5_launch(spmv_csr_ring5<256, 2048, 5120, 2, 160>, V2k, 2048, 5120, 256, 160, 512)});
        vars.push_back({"E10 ring5<256,2048,5120,D3> 512 WGs", ring5_launch(spmv_csr_ring5<256, 2048, 5120, 3, 160>, V2k, 2048, 5120, 256, 160, 512)});
        vars.push_back({"E10 ring5<256,1024,4352,D4> 768 WGs", ring5_launch(spmv_csr_ring5<256, 1024, 4352, 4, 160>, V1k, 1024, 4352, 256, 160, 768)});
        vars.push_back({"E10 ring5<1024,4096,5120,D2> 256 WGs", ring5_launch(spmv_csr_ring5<1024, 4096, 5120, 2, 160>, V4k, 4096, 5120, 1024, 160, 256)});
        vars.push_back({"E10 ring5<512,4096,5120,D2> 256 WGs", ring5_launch(spmv_csr_ring5<512, 4096, 5120, 2, 160>, V4k, 4096, 5120, 512, 160, 256)});
        {
            const int4* P; const int* OK; int wgs, bpw;
            make_plan(4096, 5120, 160, 256, &P, &OK, &wgs, &bpw);
            unsigned long long* prof = nullptr;
            (void)hipMalloc(&prof, 16 * sizeof(unsigned long long));
            (void)hipMemset(prof, 0, 16 * sizeof(unsigned long long));
            g_prof_ptr = prof;
            CsrView V = V4k;
            {
                const int4* Pd; const int* OKd; int wgsd, bpwd;
                g_want_slots = 256;
                make_plan(2048, 5120, 160, 512, &Pd, &OKd, &wgsd, &bpwd);
                g_want_slots = 0;
                const unsigned short* SLd = g_last_slots;
                unsigned long long* prof2 = nullptr;
                (void)hipMalloc(&prof2, 16 * sizeof(unsigned long long));
                (void)hipMemset(prof2, 0, 16 * sizeof(unsigned long long));
                g_prof_ptr2 = prof2;
                g_prof2_wgs = wgsd;
                CsrView Vd = V2k;
                vars.push_back({"DIAG2 ring5t C16 NT <256,2048,5120,D2>", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5t<256, 2048, 5120, 2, 160, true>), dim3(wgsd), dim3(256), 0, s, Vd, Pd, prof2, OKd, d_x, d_y, bpwd, SLd); }});
            }
            vars.push_back({"DIAG ring5t stamps <512,4096,5120,D2>", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5t<512, 4096, 5120, 2, 160>), dim3(wgs), dim3(512), 0, s, V, P, prof, OK, d_x, d_y, bpw); }});
        }
    }
    if (make_plan) {
        auto ring6_launch = [=](auto kern, CsrView V, int tab, int ring, int threads, int maxb, int min_wgs) {
            const int4* P; const int* OK; int wgs, bpw;
            make_plan(tab, ring, maxb, min_wgs, &P, &OK, &wgs, &bpw);
            return [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, s, V, P, OK, d_x, d_y, bpw); };
        };
        {
            const int4* P; const int* OK; int wgs, bpw;
            make_plan(4096, 5120, 160, 256, &P, &OK, &wgs, &bpw);
            CsrView V = V4k;
            {
                const int4* Pc; const int* OKc; int wgsc, bpwc;
                g_want_slots = 1;
                make_plan(4096, 5120, 160, 256, &Pc, &OKc, &wgsc, &bpwc);
                g_want_slots = 0;
                const unsigned short* SL = g_last_slots;
#define C16V(name, abl) vars.push_back({name, [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 2048 | (abl)>), dim3(wgsc), dim3(512), 0, s, V, Pc, OKc, d_x, d_y, bpwc, SL); }});
                C16V("C16 ring5a full", 0)
                C16V("C16 no reduce (invalid)", 1)
                C16V("C16 no gather (invalid)", 2)
                C16V("C16 no stage writes (invalid)", 4)
                C16V("C16 no reduce+gather+stage (invalid)", 7)
                C16V("C16 full but NO y stores (invalid)", 64)
                C16V("C16 skeleton no y stores (invalid)", 7 + 64)
                C16V("C16 skeleton no barriers (invalid)", 15)
                C16V("C16 skeleton no barr/xr/ptrow/stores (invalid)", 127)
                C16V("C16 full, NT loads", 1024)
                {
                    auto shape = [&](const char* name, auto kern, CsrView VV, int tab, int ring, int T, int maxb, int wgsreq) {
                        const int4* Pq; const int* OKq; int wgsq, bpwq;
                        g_want_slots = T;
                        make_plan(tab, ring, maxb, wgsreq, &Pq, &OKq, &wgsq, &bpwq);
                        g_want_slots = 0;
                        const unsigned short* SLq = g_last_slots;
                        vars.push_back({name, [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgsq), dim3(T), 0, s, VV, Pq, OKq, d_x, d_y, bpwq, SLq); }});
                    };
                    shape("C16S NT <512,2048,5120,D2> 512 WGs", spmv_csr_ring5a<512, 2048, 5120, 2, 160, 2048 + 1024>, V2k, 2048, 5120, 512, 160, 512);
                    shape("C16S NT <512,2048,5120,D3> 512 WGs", spmv_csr_ring5a<512, 2048, 5120, 3, 160, 2048 + 1024>, V2k, 2048, 5120, 512, 160, 512);
                    shape("C16S NT <256,2048,5120,D2> 512 WGs", spmv_csr_ring5a<256, 2048, 5120, 2, 160, 2048 + 1024>, V2k, 2048, 5120, 256, 160, 512);
                    shape("C16S NT <256,2048,5120,D3> 512 WGs", spmv_csr_ring5a<256, 2048, 5120, 3, 160, 2048 + 1024>, V2k, 2048, 5120, 256, 160, 512);
                    shape("C16S NT <256,2048,5120,D2> 1024 WGs", spmv_csr_ring5a<256, 2048, 5120, 2, 160, 2048 + 1024>, V2k, 2048, 5120, 256, 160, 1024);
                    shape("C16S NT <128,2048,5120,D2> 512 WGs", spmv_csr_ring5a<128, 2048, 5120, 2, 160, 2048 + 1024>, V2k, 2048, 5120, 128, 160, 512);
                    shape("C16S NT <128,1024,4224,D2> 768 WGs", spmv_csr_ring5a<128, 1024, 4224, 2, 96, 2048 + 1024>, V1k, 1024, 4224, 128, 96, 768);
                    shape("C16S NT <256,4096,5120,D2> 256 WGs", spmv_csr_ring5a<256, 4096, 5120, 2, 160, 2048 + 1024>, V, 4096, 5120, 256, 160, 256);
                    shape("C16S NT <256,2048,4224,D2> 512 WGs", spmv_csr_ring5a<256, 2048, 4224, 2, 160, 2048 + 1024>, V2k, 2048, 4224, 256, 160, 512);
                    shape("C16S NT <256,2048,5120,D2> no y stores", spmv_csr_ring5a<256, 2048, 5120, 2, 160, 2048 + 1024 + 64>, V2k, 2048, 5120, 256, 160, 512);
                    shape("C16S NT <256,2048,5120,D2> skeleton", spmv_csr_ring5a<256, 2048, 5120, 2, 160, 2048 + 1024 + 7>, V2k, 2048, 5120, 256, 160, 512);
                    shape("C16S NT <256,2048,5120,D2> no reduce", spmv_csr_ring5a<256, 2048, 5120, 2, 160, 2048 + 1024 + 1>, V2k, 2048, 5120, 256, 160, 512);
                    shape("C16S NT <256,1024,4224,D2> 768 WGs", spmv_csr_ring5a<256, 1024, 4224, 2, 96, 2048 + 1024>, V1k, 1024, 4224, 256, 96, 768);
                    shape("C16S NT <256,1024,4224,D3> 768 WGs", spmv_csr_ring5a<256, 1024, 4224, 3, 96, 2048 + 1024>, V1k, 1024, 4224, 256, 96, 768);
                    shape("C16S NT <512,1024,4224,D2> 768 WGs", spmv_csr_ring5a<512, 1024, 4224, 2, 96, 2048 + 1024>, V1k, 1024, 4224, 512, 96, 768);
                    shape("C16S NT <512,4096,5120,D2> 256 WGs", spmv_csr_ring5a<512, 4096, 5120, 2, 160, 2048 + 1024>, V, 4096, 5120, 512, 160, 256);
                    shape("C16S T  <512,2048,5120,D2> 512 WGs", spmv_csr_ring5a<512, 2048, 5120, 2, 160, 2048>, V2k, 2048, 5120, 512, 160, 512);
                    shape("C16S NT <512,2048,5120,D2> no y stores", spmv_csr_ring5a<512, 2048, 5120, 2, 160, 2048 + 1024 + 64>, V2k, 2048, 5120, 512, 160, 512);
                    shape("C16S NT <512,2048,5120,D2> skeleton", spmv_csr_ring5a<512, 2048, 5120, 2, 160, 2048 + 1024 + 7>, V2k, 2048, 5120, 512, 160, 512);
                    shape("C16S NT <512,2048,5120,D2> no reduce", spmv_csr_ring5a<512, 2048, 5120, 2, 160, 2048 + 1024 + 1>, V2k, 2048, 5120, 512, 160, 512);
                }
                C16V("C16 full, NT coef + NT y store", 1024 + 8192)
                C16V("C16 full, NT coef + NT ptrow", 1024 + 16384)
                C16V("C16 full, NT coef + NT ptrow + NT y store", 1024 + 16384 + 8192)
                C16V("C16 full, NT coef + NT slots", 1024 + 4096)
                C16V("C16 full, NT slots only", 4096)
                C16V("C16 NT no y stores (invalid)", 1024 + 64)
                C16V("C16 NT skeleton no y stores (invalid)", 1024 + 7 + 64)
                C16V("C16 NT skeleton (invalid)", 1024 + 7)
                C16V("C16 NT no reduce (invalid)", 1024 + 1)
#undef C16V
            }
            vars.push_back({"ABL ring5a full (ABL=0)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 0>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL ring5a no reduce (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 1>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL ring5a no gather (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 2>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL ring5a no stage writes (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 4>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL ring5a no reduce+gather+stage (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 7>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL skeleton D3 (ABL=7)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 3, 160, 7>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL skeleton D4 (ABL=7)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 4, 160, 7>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL skeleton D2 no barriers (ABL=15)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 15>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL skeleton D2 no barriers, no xr (31)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 31>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL skeleton D2 no barr, no xr, no ptrow (63)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 63>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL skeleton D4 no barr, no xr, no ptrow (63)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 4, 160, 63>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            for (int st : {7, 29, 37}) {
                *stagger = st;
                const int4* P2; const int* OK2; int wgs2, bpw2;
                make_plan(4096, 5120, 160, 256, &P2, &OK2, &wgs2, &bpw2);
                *stagger = 0;
                vars.push_back({"STAG" + std::to_string(st) + " ring5 full <512,4096,5120,D2>", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 0>), dim3(wgs2), dim3(512), 0, s, V, P2, OK2, d_x, d_y, bpw2); }});
                vars.push_back({"STAG" + std::to_string(st) + " skeleton D2 (ABL=7)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 7>), dim3(wgs2), dim3(512), 0, s, V, P2, OK2, d_x, d_y, bpw2); }});
            }
            for (int al : {16, 32, 64}) {
                *row_align = al;
                const int4* P3; const int* OK3; int wgs3, bpw3;
                make_plan(4096, 5120, 160, 256, &P3, &OK3, &wgs3, &bpw3);
                *row_align = 1;
                CsrView V3 = V;
                V3.nblk = wgs3 * bpw3; // upper bound; runs clamp through the plan's terminator below
                V3.nblk = g_last_plan_nblk;
                vars.push_back({"ALIGN" + std::to_string(al) + " ring5 full <512,4096,5120,D2>", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 0>), dim3(wgs3), dim3(512), 0, s, V3, P3, OK3, d_x, d_y, bpw3); }});
                vars.push_back({"ALIGN" + std::to_string(al) + " skeleton (7)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 7>), dim3(wgs3), dim3(512), 0, s, V3, P3, OK3, d_x, d_y, bpw3); }});
            }
            vars.push_back({"NT ring5 full, nt loads of coef/indcol (1024)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 1024>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"NT skeleton (7) + nt loads", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 1031>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"NT skeleton no y stores (71) + nt loads", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 1095>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"FIX y parked in LDS, flush per ~22 blocks (512)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 512>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"FIX unconditional y store D2 (128)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 128>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"FIX unconditional y store D3 (128)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 3, 160, 128>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL full but NO y stores (64)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 64>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL skeleton no y stores (7+64)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 71>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL skeleton no barr/xr/ptrow/stores (127)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 127>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
            vars.push_back({"ABL ring5a no reduce+stage (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_ring5a<512, 4096, 5120, 2, 160, 5>), dim3(wgs), dim3(512), 0, s, V, P, OK, d_x, d_y, bpw); }});
        }
        vars.push_back({"E11 ring6<512,4096,5120,D2,U8> 256 WGs", ring6_launch(spmv_csr_ring6<512, 4096, 5120, 2, 160, 8>, V4k, 4096, 5120, 512, 160, 256)});
        vars.push_back({"E11 ring6<512,4096,5120,D2,U16> 256 WGs", ring6_launch(spmv_csr_ring6<512, 4096, 5120, 2, 160, 16>, V4k, 4096, 5120, 512, 160, 256)});
        vars.push_back({"E11 ring6<512,4096,5120,D3,U16> 256 WGs", ring6_launch(spmv_csr_ring6<512, 4096, 5120, 3, 160, 16>, V4k, 4096, 5120, 512, 160, 256)});
        vars.push_back({"E11 ring6<512,2048,5120,D2,U16> 512 WGs", ring6_launch(spmv_csr_ring6<512, 2048, 5120, 2, 160, 16>, V2k, 2048, 5120, 512, 160, 512)});
        vars.push_back({"E11 ring6<256,2048,5120,D2,U16> 512 WGs", ring6_launch(spmv_csr_ring6<256, 2048, 5120, 2, 160, 16>, V2k, 2048, 5120, 256, 160, 512)});
        vars.push_back({"E11 ring6<256,4096,5120,D2,U16> 256 WGs", ring6_launch(spmv_csr_ring6<256, 4096, 5120, 2, 160, 16>, V4k, 4096, 5120, 256, 160, 256)});
        vars.push_back({"E11 ring6<1024,4096,5120,D2,U16> 256 WGs", ring6_launch(spmv_csr_ring6<1024, 4096, 5120, 2, 160, 16>, V4k, 4096, 5120, 1024, 160, 256)});
    }
    if (make_plan) {
        auto ring7_launch = [=](auto kern, CsrView V, int tab, int ring, int threads, int maxb, int min_wgs) {
            const int4* P; const int* OK; int wgs, bpw;
            make_plan(tab, ring, maxb, min_wgs, &P, &OK, &wgs, &bpw);
            return [=](hipStream_t s) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads + 64), 0, s, V, P, OK, d_x, d_y, bpw); };
        };
        vars.push_back({"E12 ring7<512,4096,5120,D2>+writer 256 WGs", ring7_launch(spmv_csr_ring7<512, 4096, 5120, 2, 160, 1024>, V4k, 4096, 5120, 512, 160, 256)});
        vars.push_back({"E12 ring7<512,4096,5120,D3>+writer 256 WGs", ring7_launch(spmv_csr_ring7<512, 4096, 5120, 3, 160, 1024>, V4k, 4096, 5120, 512, 160, 256)});
        vars.push_back({"E12 ring7<512,2048,5120,D2>+writer 512 WGs", ring7_launch(spmv_csr_ring7<512, 2048, 5120, 2, 160, 1024>, V2k, 2048, 5120, 512, 160, 512)});
        vars.push_back({"E12 ring7<256,2048,5120,D2>+writer 512 WGs", ring7_launch(spmv_csr_ring7<256, 2048, 5120, 2, 160, 1024>, V2k, 2048, 5120, 256, 160, 512)});
        vars.push_back({"E12 ring7<256,2048,5120,D3>+writer 512 WGs", ring7_launch(spmv_csr_ring7<256, 2048, 5120, 3, 160, 1024>, V2k, 2048, 5120, 256, 160, 512)});
    }
}
