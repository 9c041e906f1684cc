// capi_blas1.hip: dot, AXPY, orthogonalize, Gram-Schmidt sweep, norms — part of libmi355spmv.so (see capi_internal.hpp for the layout of the library).
// Built for gfx950 only; no CPU fallback anywhere: every compute entry point needs a HIP device.
#include "capi_internal.hpp"
#include "blas1_kernels.hpp"

// Reduction workspace: partials of the two-stage reductions, one per (device, stream) so that
// reductions enqueued on different streams (or by different rank threads of one process) never share
// partials.  4 * kMaxPartials doubles: [0, 2K) the two partial arrays of a reduction (or the
// ping-pong pair of the Gram-Schmidt sweep), the rest spare.  32 KB per stream that ever reduced; a
// destroyed stream's slot is simply reused if the runtime hands the same handle out again.
static std::map<std::pair<int, hipStream_t>, double*> g_ws;

int get_ws(hipStream_t s, double** out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mu);
    double*& w = g_ws[std::make_pair(dev, s)];
    if (!w) HIP_TRY(hipMalloc(&w, sizeof(double) * (4 * kMaxPartials + 8)));
    *out = w;
    return MI_OK;
}

// ---------------------------------------------------------------- BLAS-1
// large vectors are streamed past the caches (blas1_kernels.hpp); MI355_BLAS1_NT=0|1 forces the choice
static bool blas1_nt(int n)
{
    static const int forced = getenv("MI355_BLAS1_NT") ? atoi(getenv("MI355_BLAS1_NT")) : -1;
    return forced >= 0 ? forced != 0 : n >= kBlas1NtMin;
}

static int red_geometry(int n, int* np, int* seg)
{
    // segments of a multiple of 2*kRedWG elements, at most kMaxPartials of them
    long long s = ((long long)n + kMaxPartials - 1) / kMaxPartials;
    const int q = 2 * kRedWG;
    s = ((s + q - 1) / q) * q;
    if (s < q) s = q;
    *seg = (int)s;
    *np = (int)(((long long)n + s - 1) / s);
    if (*np < 1) *np = 1;
    return MI_OK;
}

template <int MODE, int FIN>
static int reduce_dev(int n, const double* a, const double* b, double* d_out, hipStream_t s)
{
    CHECK_ARG(n >= 0, "negative n");
    CHECK_ARG(d_out && (n == 0 || (a && b)), "null vector");
    double* ws = nullptr;
    int rc = get_ws(s, &ws);
    if (rc) return rc;
    int np, seg;
    red_geometry(n, &np, &seg);
    if (blas1_nt(n)) hipLaunchKernelGGL((reduce_stage1<MODE, true>), dim3(np), dim3(kRedWG), 0, s, n, seg, a, b, ws, ws + kMaxPartials);
    else hipLaunchKernelGGL((reduce_stage1<MODE, false>), dim3(np), dim3(kRedWG), 0, s, n, seg, a, b, ws, ws + kMaxPartials);
    hipLaunchKernelGGL((reduce_stage2<FIN>), dim3(1), dim3(kRedWG), 0, s, np, ws, ws + kMaxPartials, d_out);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_dot_dev(int n, const double* d_x, const double* d_y, double* d_out, mi_stream_t s)
{
    return reduce_dev<0, 0>(n, d_x, d_y, d_out, (hipStream_t)s);
}

extern "C" int mi_norm2_dev(int n, const double* d_x, double* d_out, mi_stream_t s)
{
    return reduce_dev<0, 1>(n, d_x, d_x, d_out, (hipStream_t)s);
}

extern "C" int mi_rel_error_dev(int n, const double* d_ref, const double* d_test, double* d_out, mi_stream_t s)
{
    return reduce_dev<1, 2>(n, d_ref, d_test, d_out, (hipStream_t)s);
}

extern "C" int mi_axpy_dev(int n, double a, const double* d_x, double* d_y, mi_stream_t s)
{
    CHECK_ARG(n >= 0, "negative n");
    CHECK_ARG(n == 0 || (d_x && d_y), "null vector");
    if (n == 0) return MI_OK;
    int grid = (n / 2 + 255) / 256;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, n, a, d_x, d_y);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_orthogonalize_dev(int n, const double* d_b, const double* d_x1, double* d_x3, double alpha,
                                    double* d_beta_out, mi_stream_t s_)
{
    CHECK_ARG(d_beta_out, "null beta");
    CHECK_ARG(n >= 0, "negative n");
    hipStream_t s = (hipStream_t)s_;
    if (n == 0) return reduce_dev<0, 0>(n, d_b, d_x1, d_beta_out, s); // beta = 0
    CHECK_ARG(d_b && d_x1 && d_x3, "null vector");
    // two kernels: per-workgroup partials of b.x1, then the update, whose workgroups each finish the dot themselves
    double* ws = nullptr;
    int rc = get_ws(s, &ws);
    if (rc) return rc;
    int np, seg;
    red_geometry(n, &np, &seg);
    int grid = (n + kRedWG - 1) / kRedWG;
    if (grid > 2048) grid = 2048;
    if (blas1_nt(n)) {
        hipLaunchKernelGGL((reduce_stage1<0, true>), dim3(np), dim3(kRedWG), 0, s, n, seg, d_b, d_x1, ws, ws + kMaxPartials);
        hipLaunchKernelGGL(ortho_update_kernel<true>, dim3(grid), dim3(kRedWG), 0, s, n, alpha, np, ws, d_beta_out, d_b, d_x1, d_x3);
    } else {
        hipLaunchKernelGGL((reduce_stage1<0, false>), dim3(np), dim3(kRedWG), 0, s, n, seg, d_b, d_x1, ws, ws + kMaxPartials);
        hipLaunchKernelGGL(ortho_update_kernel<false>, dim3(grid), dim3(kRedWG), 0, s, n, alpha, np, ws, d_beta_out, d_b, d_x1, d_x3);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// the update of a distributed orthogonalize on one rank's slice: beta = sum of the ranks' partial dots in rank order (capi_dist.hip)
int ortho_update_from_parts(int n, int nparts, const double* parts, double alpha, const double* d_b, const double* d_x1, double* d_x3, double* d_beta_out,
                            hipStream_t s)
{
    CHECK_ARG(n >= 0 && nparts >= 1 && nparts <= 64 && parts, "bad argument");
    CHECK_ARG(n == 0 || (d_b && d_x1 && d_x3), "null vector");
    int grid = (n + kRedWG - 1) / kRedWG;
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    if (blas1_nt(n)) hipLaunchKernelGGL(ortho_update_ranks_kernel<true>, dim3(grid), dim3(kRedWG), 0, s, n, alpha, nparts, parts, d_beta_out, d_b, d_x1, d_x3);
    else hipLaunchKernelGGL(ortho_update_ranks_kernel<false>, dim3(grid), dim3(kRedWG), 0, s, n, alpha, nparts, parts, d_beta_out, d_b, d_x1, d_x3);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// ---- product + dot in one pass (the f-4 pipeline SpMV -> dot + AXPY -> SpMV, mpk/SpMVmulti.cpp:563-569) -------------------
// y = A x with the partial sums of b . y accumulated in the product's epilogue where the launch can carry it (the LEAN ring
// kernel: ring_dot_eligible), else product and dot as separate launches — the same API either way.  *np = partials written to ws.
static int spmv_with_dot_partials(mi_csr_t A, const double* d_x, double* d_y, const double* d_b, hipStream_t s, double* ws, int* np)
{
    int rc;
    if (ring_dot_eligible(A)) {
        RingDot dt{d_b, ws};
        if ((rc = launch_spmv(A, d_x, d_y, s, true, nullptr, &dt))) return rc;
        *np = A->ring.wgs;
        return MI_OK;
    }
    if ((rc = launch_spmv(A, d_x, d_y, s))) return rc;
    int seg;
    red_geometry(A->n, np, &seg);
    if (blas1_nt(A->n)) hipLaunchKernelGGL((reduce_stage1<0, true>), dim3(*np), dim3(kRedWG), 0, s, A->n, seg, d_b, d_y, ws, ws + kMaxPartials);
    else hipLaunchKernelGGL((reduce_stage1<0, false>), dim3(*np), dim3(kRedWG), 0, s, A->n, seg, d_b, d_y, ws, ws + kMaxPartials);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_csr_dot_epilogue_info(mi_csr_t A, int* in_epilogue)
{
    CHECK_ARG(A && in_epilogue, "null argument");
    *in_epilogue = ring_dot_eligible(A) ? 1 : 0;
    return MI_OK;
}

extern "C" int mi_spmv_dot_dev(mi_csr_t A, const double* d_x, double* d_y, const double* d_b, double* d_beta_out, mi_stream_t s_)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(d_beta_out, "null beta");
    CHECK_ARG(!A->mapped, "needs an unmapped matrix (b is indexed like y)");
    hipStream_t s = (hipStream_t)s_;
    if (A->n == 0) return reduce_dev<0, 0>(0, d_b, d_y, d_beta_out, s);
    CHECK_ARG(d_x && d_y && d_b, "null vector");
    double* ws = nullptr;
    int rc = get_ws(s, &ws), np = 0;
    if (rc) return rc;
    if ((rc = spmv_with_dot_partials(A, d_x, d_y, d_b, s, ws, &np))) return rc;
    hipLaunchKernelGGL((reduce_stage2<0>), dim3(1), dim3(kRedWG), 0, s, np, ws, ws + kMaxPartials, d_beta_out);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_spmv_orthogonalize_dev(mi_csr_t A, const double* d_x, double* d_x1, const double* d_b, double* d_x3, double alpha,
                                         double* d_beta_out, mi_stream_t s_)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(d_beta_out, "null beta");
    CHECK_ARG(!A->mapped, "needs an unmapped matrix (b is indexed like y)");
    hipStream_t s = (hipStream_t)s_;
    const int n = A->n;
    if (n == 0) return reduce_dev<0, 0>(0, d_b, d_x1, d_beta_out, s);
    CHECK_ARG(d_x && d_x1 && d_b && d_x3, "null vector");
    double* ws = nullptr;
    int rc = get_ws(s, &ws), np = 0;
    if (rc) return rc;
    if ((rc = spmv_with_dot_partials(A, d_x, d_x1, d_b, s, ws, &np))) return rc;
    int grid = (n + kRedWG - 1) / kRedWG;
    if (grid > 2048) grid = 2048;
    if (blas1_nt(n)) hipLaunchKernelGGL(ortho_update_kernel<true>, dim3(grid), dim3(kRedWG), 0, s, n, alpha, np, ws, d_beta_out, d_b, d_x1, d_x3);
    else hipLaunchKernelGGL(ortho_update_kernel<false>, dim3(grid), dim3(kRedWG), 0, s, n, alpha, np, ws, d_beta_out, d_b, d_x1, d_x3);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_orthonormalize_against_basis_dev(int n, int m, const double* const* d_basis, double* d_y, double* d_dots,
                                                    mi_stream_t s_)
{
    CHECK_ARG(n >= 0 && m >= 0, "negative size");
    if (m == 0) return MI_OK;
    CHECK_ARG(d_basis && d_dots, "null basis / dots");
    hipStream_t s = (hipStream_t)s_;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_dots, 0, sizeof(double) * (size_t)m, s));
        return MI_OK;
    }
    CHECK_ARG(d_y, "null y");
    for (int j = 0; j < m; j++) CHECK_ARG(d_basis[j], "null basis vector");
    double* ws = nullptr;
    int rc = get_ws(s, &ws);
    if (rc) return rc;
    int np, seg;
    red_geometry(n, &np, &seg);
    double* part[2] = {ws, ws + kMaxPartials};
    const bool nt = blas1_nt(n);
    // dot of the first vector, then one launch per vector: finish dot j, update y, partials of dot j+1
    if (nt) hipLaunchKernelGGL((reduce_stage1<0, true>), dim3(np), dim3(kRedWG), 0, s, n, seg, d_y, d_basis[0], part[0], part[1]);
    else hipLaunchKernelGGL((reduce_stage1<0, false>), dim3(np), dim3(kRedWG), 0, s, n, seg, d_y, d_basis[0], part[0], part[1]);
    for (int j = 0; j < m; j++) {
        const double* vn = j + 1 < m ? d_basis[j + 1] : nullptr;
        if (nt) hipLaunchKernelGGL(mgs_step_kernel<true>, dim3(np), dim3(kRedWG), 0, s, n, seg, np, part[j & 1], d_dots + j, d_basis[j], vn, d_y, part[(j + 1) & 1]);
        else hipLaunchKernelGGL(mgs_step_kernel<false>, dim3(np), dim3(kRedWG), 0, s, n, seg, np, part[j & 1], d_dots + j, d_basis[j], vn, d_y, part[(j + 1) & 1]);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_gather_dev(int m, const int* d_idx, const double* d_src, double* d_dst, mi_stream_t s)
{
    CHECK_ARG(m >= 0, "negative m");
    if (m == 0) return MI_OK;
    CHECK_ARG(d_idx && d_src && d_dst, "null pointer");
    int grid = (m + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, m, d_idx, d_src, d_dst);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_dot(int n, const double* x, const double* y, double* out)
{
    CHECK_ARG(n >= 0 && out && (n == 0 || (x && y)), "bad argument");
    int rc = need_device();
    if (rc) return rc;
    Scratch S;
    double *dx, *dy, *dout;
    if ((rc = S.up(x, n, &dx)) || (rc = S.up(y, n, &dy)) || (rc = S.up(nullptr, 1, &dout))) return rc;
    if ((rc = mi_dot_dev(n, dx, dy, dout, nullptr))) return rc;
    HIP_TRY(hipMemcpy(out, dout, sizeof(double), hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" int mi_norm2(int n, const double* x, double* out)
{
    CHECK_ARG(n >= 0 && out && (n == 0 || x), "bad argument");
    int rc = need_device();
    if (rc) return rc;
    Scratch S;
    double *dx, *dout;
    if ((rc = S.up(x, n, &dx)) || (rc = S.up(nullptr, 1, &dout))) return rc;
    if ((rc = mi_norm2_dev(n, dx, dout, nullptr))) return rc;
    HIP_TRY(hipMemcpy(out, dout, sizeof(double), hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" int mi_rel_error(int n, const double* ref, const double* test, double* out)
{
    CHECK_ARG(n >= 0 && out && (n == 0 || (ref && test)), "bad argument");
    int rc = need_device();
    if (rc) return rc;
    Scratch S;
    double *da, *db, *dout;
    if ((rc = S.up(ref, n, &da)) || (rc = S.up(test, n, &db)) || (rc = S.up(nullptr, 1, &dout))) return rc;
    if ((rc = mi_rel_error_dev(n, da, db, dout, nullptr))) return rc;
    HIP_TRY(hipMemcpy(out, dout, sizeof(double), hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" int mi_axpy(int n, double a, const double* x, double* y)
{
    CHECK_ARG(n >= 0 && (n == 0 || (x && y)), "bad argument");
    int rc = need_device();
    if (rc) return rc;
    Scratch S;
    double *dx, *dy;
    if ((rc = S.up(x, n, &dx)) || (rc = S.up(y, n, &dy))) return rc;
    if ((rc = mi_axpy_dev(n, a, dx, dy, nullptr))) return rc;
    if (n) HIP_TRY(hipMemcpy(y, dy, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" int mi_orthogonalize(int n, const double* b, const double* x1, double* x3, double alpha, double* beta_out)
{
    CHECK_ARG(n >= 0 && (n == 0 || (b && x1 && x3)), "bad argument");
    int rc = need_device();
    if (rc) return rc;
    Scratch S;
    double *db, *dx1, *dx3, *dbeta;
    if ((rc = S.up(b, n, &db)) || (rc = S.up(x1, n, &dx1)) || (rc = S.up(nullptr, n, &dx3)) || (rc = S.up(nullptr, 1, &dbeta)))
        return rc;
    if ((rc = mi_orthogonalize_dev(n, db, dx1, dx3, alpha, dbeta, nullptr))) return rc;
    if (n) HIP_TRY(hipMemcpy(x3, dx3, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    double beta = 0.0;
    HIP_TRY(hipMemcpy(&beta, dbeta, sizeof(double), hipMemcpyDeviceToHost));
    if (beta_out) *beta_out = beta;
    return MI_OK;
}

extern "C" int mi_orthonormalize_against_basis(int n, int m, const double* const* basis, double* y, double* dots_out)
{
    CHECK_ARG(n >= 0 && m >= 0 && (m == 0 || basis) && (n == 0 || y), "bad argument");
    int rc = need_device();
    if (rc) return rc;
    if (m == 0) return MI_OK;
    Scratch S;
    std::vector<const double*> dv((size_t)m);
    double *dy = nullptr, *dd = nullptr;
    for (int j = 0; j < m; j++) {
        CHECK_ARG(n == 0 || basis[j], "null basis vector");
        double* p = nullptr;
        if ((rc = S.up(basis[j], n, &p))) return rc;
        dv[j] = p;
    }
    if ((rc = S.up(y, n, &dy)) || (rc = S.up(nullptr, m, &dd))) return rc;
    if ((rc = mi_orthonormalize_against_basis_dev(n, m, dv.data(), dy, dd, nullptr))) return rc;
    if (n) HIP_TRY(hipMemcpy(y, dy, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    if (dots_out) HIP_TRY(hipMemcpy(dots_out, dd, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost));
    else HIP_TRY(hipDeviceSynchronize());
    return MI_OK;
}

// ---------------------------------------------------------------- permutations of a relabelled handle (reorder.hpp)
// x into a reordered handle's numbering (whole nodes at a time when nodes were moved and x allows 16-byte accesses)
int gather_perm(mi_csr_t A, const double* d_x, double* d_xp, hipStream_t s)
{
    if (A->reorder_block == 4 && (((uintptr_t)d_x | (uintptr_t)d_xp) & 15) == 0) {
        const int nn = A->n / 4;
        int grid = (nn + 255) / 256;
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(gather_nodes_kernel, dim3(grid), dim3(256), 0, s, nn, A->d_iperm, d_x, d_xp);
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    return mi_gather_dev(A->n, A->d_iperm, d_x, d_xp, (mi_stream_t)s);
}


// dst[idx[i]] = src[i]
__global__ __launch_bounds__(256) void scatter_kernel(int m, const int* __restrict__ idx, const double* __restrict__ src,
                                                      double* __restrict__ dst)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) dst[idx[i]] = src[i];
}

// a vector in a reordered handle's numbering back into the caller's: dst[iperm[r']] = src[r']
int scatter_perm(mi_csr_t A, const double* d_src, double* d_dst, hipStream_t s)
{
    int grid = (A->n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(scatter_kernel, dim3(grid), dim3(256), 0, s, A->n, A->d_iperm, d_src, d_dst);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}
