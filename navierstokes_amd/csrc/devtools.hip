// devtools.hip: diagnostics for tools/ — linked into libmi355spmv_dev.so only (include/mi355_devtools.h)
#include "capi_internal.hpp"
#include "mi355_devtools.h"

// diagnostic: which XCD each workgroup of a launch shaped like the ring kernel's lands on
__global__ __launch_bounds__(256) void xcc_probe_kernel(int* out)
{
    __shared__ double hog[9000]; // ~70 KB: two workgroups per CU, like ring configuration 4
    hog[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; // HW_REG_XCC_ID[3:0]
    if (hog[threadIdx.x] != 0.0) out[blockIdx.x] = -1;
}

extern "C" int mi_debug_xcc_map(int wgs, int* host_out)
{
    CHECK_ARG(wgs > 0 && host_out, "bad argument");
    int rc = need_device();
    if (rc) return rc;
    int* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(int) * wgs));
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(wgs), dim3(256), 0, nullptr, d);
    HIP_TRY(hipMemcpy(host_out, d, sizeof(int) * wgs, hipMemcpyDeviceToHost));
    dfree(d);
    return MI_OK;
}
// one 4-byte read every `stride` bytes of an array: brings its address translations (and 1 line per stride) back after
// mi_flush_cache() without bringing the data back — separates "cold caches" from "cold TLB" in a cold-start measurement
__global__ __launch_bounds__(256) void touch_pages_kernel(const char* __restrict__ p, size_t bytes, size_t stride, int* __restrict__ sink)
{
    int acc = 0;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * stride < bytes; i += step)
        acc += *reinterpret_cast<const int*>(p + i * stride);
    if (acc == 0x7fffffff) sink[0] = acc;
}

extern "C" int mi_debug_touch_pages(mi_csr_t A, int stride_bytes, const void* d_extra0, long long bytes0, const void* d_extra1, long long bytes1)
{
    CHECK_ARG(A && stride_bytes >= 64 && stride_bytes % 4 == 0, "bad argument");
    if (A->inner) A = A->inner;
    int* sink = nullptr;
    HIP_TRY(hipMalloc(&sink, 64));
    auto touch = [&](const void* p, size_t bytes) {
        if (!p || bytes < 4) return;
        hipLaunchKernelGGL(touch_pages_kernel, dim3(256), dim3(256), 0, nullptr, (const char*)p, bytes - 3, (size_t)stride_bytes, sink);
    };
    const size_t nnz = (size_t)A->nnz, n = (size_t)A->n;
    touch(A->d_coef, 8 * nnz);
    touch(A->d_indcol, 4 * nnz);
    touch(A->d_ptrow, 4 * (n + 1));
    touch(A->d_rowmap, 4 * n);
    if (A->ring.d_plan) {
        touch(A->ring.d_plan, 32 * (size_t)A->ring.nblk);
        touch(A->ring.d_slots, 2 * (size_t)A->ring.nblk * A->ring.cfg.nnzb);
    }
    if (A->tile.d_desc) {
        touch(A->tile.d_desc, 16 * (size_t)A->tile.nblk);
        touch(A->tile.d_ulist, (size_t)(A->tile.unique_per_nnz * 4.0 * (double)nnz));
        touch(A->tile.d_slots, 2 * nnz);
    }
    if (A->blocked) {
        touch(A->blocked->d_coef, 128 * (size_t)A->blocked->nblocks);
        touch(A->blocked->d_indcol, 4 * (size_t)A->blocked->nblocks);
        touch(A->blocked->d_ptrow, 4 * ((size_t)A->blocked->nbrows + 1));
    }
    touch(d_extra0, (size_t)bytes0);
    touch(d_extra1, (size_t)bytes1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    dfree(sink);
    return MI_OK;
}
// development aid (tools/sim_rank.py): set every flag slot of MY window to `value`, so that one rank's step can be timed on
// one GPU with its pushes looped back into its own window and its waits satisfied in advance
extern "C" int mi_part_push_debug_preset(mi_part_t P, unsigned value)
{
    CHECK_ARG(P && P->win, "no window");
    std::vector<unsigned> f((size_t)P->plan.nranks * kWinFlagStride, value);
    HIP_TRY(hipMemcpy(P->win_flags, f.data(), sizeof(unsigned) * f.size(), hipMemcpyHostToDevice));
    return MI_OK;
}
