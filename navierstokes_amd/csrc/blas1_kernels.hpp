// blas1_kernels.hpp — the vector primitives that sit between SpMVs in the
// reference's pipelines: dot + AXPY ("orthogonalize", mpk/SpMVmulti.cpp:146-151,
// mpk/2SpMV.cpp:3-11), norm2 / rel_error (mpk/utils.cpp:131-143) and the halo
// pack gather.  All are HBM-bound streams: 16-byte loads per lane, fixed grids,
// and a fixed two-stage reduction tree (lane chain -> wave shuffle -> LDS across
// the 4 waves -> one partial per workgroup -> one finishing workgroup), so a
// result depends only on (n, data), never on scheduling: deterministic run to
// run, though not the CPU's left-to-right order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

constexpr int kRedWG = 256;
// NT: stream the vectors past the caches (non-temporal loads / stores).  Between two SpMVs of a
// Krylov step the dot + update touch 160 MB of vectors at C4 size; left temporal they push the ring
// kernel's 16-bit column stream and x out of the Infinity Cache and each SpMV of the pipeline costs
// 167 us instead of 152 (tools/bench_pipeline.py: 368 -> 350 us per SpMV-orthogonalize-SpMV pass with
// NT).  Small vectors live in the caches anyway and stay temporal (kBlas1NtMin).
constexpr int kBlas1NtMin = 2000000;
typedef double d2v_ __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 ld2_stream(const double* p)
{
    if (NT) {
        const d2v_ v = __builtin_nontemporal_load(reinterpret_cast<const d2v_*>(p));
        return make_double2(v.x, v.y);
    }
    return *reinterpret_cast<const double2*>(p);
}
template <bool NT>
__device__ __forceinline__ double ld1_stream(const double* p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
constexpr int kMaxPartials = 1024; // = 4 per thread of the finishing workgroup

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v; // lane 0 holds the sum
}

// Sum of one value per thread over the workgroup, fixed order; valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* s_part /* [4] */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_part[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) t = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
    __syncthreads();
    return t;
}

// MODE 0: sum a_i b_i     MODE 1: sum (a_i-b_i)^2 and sum a_i^2 (rel_error)
// Workgroup g owns the contiguous segment [g*seg, (g+1)*seg); seg is a multiple
// of 2*kRedWG so every lane's double2 is 16-byte aligned when the bases are.
template <int MODE, bool NT>
__global__ __launch_bounds__(kRedWG) void reduce_stage1(int n, int seg, const double* __restrict__ a,
                                                        const double* __restrict__ b,
                                                        double* __restrict__ partial,
                                                        double* __restrict__ partial2)
{
    __shared__ double s_part[4];
    const long long lo = (long long)blockIdx.x * seg;
    const long long hi = (lo + seg < n) ? lo + seg : n;
    double s = 0.0, s2 = 0.0;
    const bool aligned = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    if (aligned) {
        for (long long i = lo + 2 * threadIdx.x; i + 1 < hi; i += 2 * kRedWG) {
            const double2 av = ld2_stream<NT>(a + i);
            const double2 bv = ld2_stream<NT>(b + i);
            if (MODE == 0) {
                s = fma(av.x, bv.x, s);
                s = fma(av.y, bv.y, s);
            } else {
                const double d0 = av.x - bv.x, d1 = av.y - bv.y;
                s = fma(d0, d0, s);
                s = fma(d1, d1, s);
                s2 = fma(av.x, av.x, s2);
                s2 = fma(av.y, av.y, s2);
            }
        }
        // odd tail element of the segment (only the last segment can have one)
        if (((hi - lo) & 1) && threadIdx.x == 0) {
            const double av = a[hi - 1], bv = b[hi - 1];
            if (MODE == 0) s = fma(av, bv, s);
            else { const double d = av - bv; s = fma(d, d, s); s2 = fma(av, av, s2); }
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += kRedWG) {
            const double av = a[i], bv = b[i];
            if (MODE == 0) s = fma(av, bv, s);
            else { const double d = av - bv; s = fma(d, d, s); s2 = fma(av, av, s2); }
        }
    }
    const double t = block_sum(s, s_part);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
    if (MODE == 1) {
        const double t2 = block_sum(s2, s_part);
        if (threadIdx.x == 0) partial2[blockIdx.x] = t2;
    }
}

// The finishing tree over the per-workgroup partials, fixed order; valid in thread 0 (kRedWG threads).
__device__ __forceinline__ double finish_sum(int np, const double* __restrict__ partial, double* s_part)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += kRedWG) s += partial[i];
    return block_sum(s, s_part);
}

// FIN 0: out = sum   FIN 1: out = sqrt(sum)   FIN 2: out = sqrt(sum)/sqrt(sum2)
template <int FIN>
__global__ __launch_bounds__(kRedWG) void reduce_stage2(int np, const double* __restrict__ partial,
                                                        const double* __restrict__ partial2,
                                                        double* __restrict__ out)
{
    __shared__ double s_part[4];
    double s2 = 0.0;
    if (FIN == 2)
        for (int i = threadIdx.x; i < np; i += kRedWG) s2 += partial2[i];
    const double t = finish_sum(np, partial, s_part);
    double t2 = 0.0;
    if (FIN == 2) t2 = block_sum(s2, s_part);
    if (threadIdx.x == 0) {
        if (FIN == 0) out[0] = t;
        else if (FIN == 1) out[0] = sqrt(t);
        else out[0] = sqrt(t) / sqrt(t2);
    }
}

// y += a x
__global__ __launch_bounds__(256) void axpy_kernel(int n, double a, const double* __restrict__ x,
                                                   double* __restrict__ y)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool aligned = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    if (aligned) {
        const long long n2 = n >> 1;
        for (long long i = t; i < n2; i += stride) {
            const double2 xv = reinterpret_cast<const double2*>(x)[i];
            double2 yv = reinterpret_cast<double2*>(y)[i];
            yv.x = fma(a, xv.x, yv.x);
            yv.y = fma(a, xv.y, yv.y);
            reinterpret_cast<double2*>(y)[i] = yv;
        }
        if ((n & 1) && t == 0) y[n - 1] = fma(a, x[n - 1], y[n - 1]);
    } else {
        for (long long i = t; i < n; i += stride) y[i] = fma(a, x[i], y[i]);
    }
}

// out = x1 - (alpha * beta) * b with beta read from device memory (no host round trip between the
// dot and the update): the AXPY half of orthogonalize.  Evaluated as the reference's object code
// evaluates it (g++ -O3 on an FMA target contracts x1[i] - (alpha*beta)*b[i], mpk/SpMVmulti.cpp:149 and
// mpk/2SpMV.cpp:10, into ONE vfnmadd per element behind a rounded alpha*beta): out = fma(-(alpha*beta), b, x1)
// — pinned by tests/golden/blas1_*.npz.  out may alias x1 (the in-place form of mpk/2SpMV.cpp:3-11).
template <bool NT>
__global__ __launch_bounds__(kRedWG) void ortho_update_kernel(int n, double alpha, int np, const double* __restrict__ partial,
                                                              double* __restrict__ beta_out, const double* __restrict__ b,
                                                              const double* x1, double* out)
{
    // every workgroup finishes the dot itself — the same fixed tree over the same <= 1024 partials (8 KB, in L2),
    // hence the same bits everywhere — instead of waiting for a one-workgroup kernel to publish beta
    __shared__ double s_part[4];
    __shared__ double s_beta;
    const double t = finish_sum(np, partial, s_part);
    if (threadIdx.x == 0) {
        s_beta = t;
        if (blockIdx.x == 0) beta_out[0] = t;
    }
    __syncthreads();
    const double nab = -__dmul_rn(alpha, s_beta);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = fma(nab, ld1_stream<NT>(b + i), ld1_stream<NT>(x1 + i));
        if (NT) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}

// The same update for a vector spread over several ranks (mi_dist_orthogonalize*, capi_dist.hip): beta = the ranks' partial dots added
// in RANK ORDER — what the host did with them until round 4, between two stream synchronisations — by every workgroup itself (the
// same <= 64 numbers in the same order: the same bits everywhere, on every rank), then out = fma(-(alpha beta), b, x1) on this rank's
// slice.  `parts` may live in pinned host memory (the event / push exchanges) or be the result of an ncclAllGather (the RCCL exchange).
template <bool NT>
__global__ __launch_bounds__(kRedWG) void ortho_update_ranks_kernel(int n, double alpha, int nparts, const double* parts, double* __restrict__ beta_out,
                                                                    const double* __restrict__ b, const double* x1, double* out)
{
    __shared__ double s_beta;
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int r = 0; r < nparts; r++) t += __hip_atomic_load(parts + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (never from a stale cache line)
        s_beta = t;
        if (blockIdx.x == 0 && beta_out) beta_out[0] = t;
    }
    __syncthreads();
    const double nab = -__dmul_rn(alpha, s_beta);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = fma(nab, ld1_stream<NT>(b + i), ld1_stream<NT>(x1 + i));
        if (NT) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}

// One step of orthonormalize_against_basis (mpk/2SpMV.cpp:13-28), fused with the NEXT step's dot:
//   d = sum of partial_in (fixed tree) = y . v;  y <- fma(-d, v, y);  partial_out[g] = (y_new . v_next) over segment g
// so that a sweep against m vectors is m + 1 launches and reads y once per vector instead of twice
// (32 instead of 40 bytes per element and vector).  Workgroup g owns the same segment as in
// reduce_stage1 (grid = np workgroups).  Every workgroup finishes d itself from the same partials in the
// same order, hence the same bits everywhere.  v_next == nullptr: last vector, no further dot.
// Like the reference each projection uses the y updated by the previous ones (the loop is sequential:
// modified Gram-Schmidt in effect), which is why the dots cannot be batched.
template <bool NT>
__global__ __launch_bounds__(kRedWG) void mgs_step_kernel(int n, int seg, int np, const double* __restrict__ partial_in,
                                                          double* __restrict__ dot_out, const double* __restrict__ v,
                                                          const double* __restrict__ v_next, double* __restrict__ y,
                                                          double* __restrict__ partial_out)
{
    __shared__ double s_part[4];
    __shared__ double s_d;
    const double t = finish_sum(np, partial_in, s_part);
    if (threadIdx.x == 0) {
        s_d = t;
        if (blockIdx.x == 0) dot_out[0] = t;
    }
    __syncthreads();
    const double nd = -s_d;
    const long long lo = (long long)blockIdx.x * seg;
    const long long hi = (lo + seg < n) ? lo + seg : n;
    double s = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += kRedWG) {
        const double yn = fma(nd, ld1_stream<NT>(v + i), ld1_stream<NT>(y + i));
        if (NT) __builtin_nontemporal_store(yn, y + i);
        else y[i] = yn;
        if (v_next) s = fma(yn, ld1_stream<NT>(v_next + i), s);
    }
    if (v_next) { // uniform branch
        const double p = block_sum(s, s_part);
        if (threadIdx.x == 0) partial_out[blockIdx.x] = p;
    }
}

// dst[4j .. 4j+3] = src[idx[4j] .. idx[4j]+3]: the gather of a NODE renumbering (four dofs stay together and idx[4j] is a
// multiple of 4), 32 bytes per thread as two 16-byte accesses instead of four 8-byte ones
__global__ __launch_bounds__(256) void gather_nodes_kernel(int nnodes, const int* __restrict__ idx, const double* __restrict__ src,
                                                           double* __restrict__ dst)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < nnodes; j += stride) {
        const double2* s = reinterpret_cast<const double2*>(src + idx[4 * j]);
        const double2 a = s[0], b = s[1];
        double2* d = reinterpret_cast<double2*>(dst + 4 * j);
        d[0] = a;
        d[1] = b;
    }
}

// dst[i] = src[idx[i]]
__global__ __launch_bounds__(256) void gather_kernel(int m, const int* __restrict__ idx,
                                                     const double* __restrict__ src,
                                                     double* __restrict__ dst)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        dst[i] = src[idx[i]];
}

} // namespace mi355
