// tools/spmv_experimental.hpp — kernel variants under evaluation (development only).
// A variant graduates into navierstokes_amd/csrc/spmv_kernels.hpp once it is
// bit-exact and measurably faster on the GPU box; nothing in the product library
// includes this file.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <string>
#include <vector>

#include "spmv_kernels.hpp"

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

} // namespace mi355

inline void add_experimental_variants(std::vector<Variant>& vars, int n, const int* d_ptrow, const int* d_indcol,
                                      const double* d_coef, const double* d_x, double* d_y, mi355::CsrView V1k,
                                      mi355::CsrView V2k, mi355::CsrView V4k)
{
    using namespace mi355;
    (void)n; (void)d_ptrow; (void)d_indcol; (void)d_coef;
    auto grid8 = [](int nblk) { return dim3(kNXCD * ((nblk + kNXCD - 1) / kNXCD)); };
    vars.push_back({"E1 stream_v2<2048> (16B loads)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_v2<2048>), grid8(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    vars.push_back({"E1 stream_v2<1024> (16B loads)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_v2<1024>), grid8(V1k.nblk), dim3(kWG), 0, s, V1k, d_x, d_y); }});
    vars.push_back({"E1 stream_v2<4096> (16B loads)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_v2<4096>), grid8(V4k.nblk), dim3(kWG), 0, s, V4k, d_x, d_y); }});
    vars.push_back({"E2 stream<2048> no XCD remap", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_noremap<2048>), dim3(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
    vars.push_back({"E3 stream<2048> NO GATHER (invalid)", [=](hipStream_t s) { hipLaunchKernelGGL((spmv_csr_stream_nogather<2048>), grid8(V2k.nblk), dim3(kWG), 0, s, V2k, d_x, d_y); }});
}
