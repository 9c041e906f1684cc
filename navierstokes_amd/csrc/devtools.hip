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

// the multi-window ring kernel's timeline: one launch of the TRACE instantiation (depth 4, non-temporal, unmapped) on the handle's
// own tables (the relabelled twin's, if there is one); out[4 * i] = {start, end (100 MHz ticks), XCD, blocks (< 0: plain path)} of
// workgroup i.  x and y are device vectors in the numbering of the matrix that runs.
#include "spmv_mring.hpp"
extern "C" int mi_debug_mring_trace(mi_csr_t A, const double* d_x, double* d_y, int max_wgs, long long* host_out, int* wgs_out)
{
    CHECK_ARG(A && d_x && d_y && host_out && wgs_out, "null argument");
    if (A->inner) A = A->inner;
    const MringTable& M = A->mring;
    if (!M.d_plan) return fail(MI_ERR_STATE, "no multi-window ring plan on this handle");
    CHECK_ARG(M.wgs <= max_wgs, "output array too small");
    CsrView V{};
    V.n = A->n;
    V.ncols = A->ncols;
    V.ptrow = A->d_ptrow;
    V.indcol = A->d_indcol;
    V.coef = A->d_coef;
    V.rowmap = nullptr;
    V.nblk = M.nblk;
    long long* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(long long) * 4 * (size_t)M.wgs));
    HIP_TRY(hipMemset(d, 0, sizeof(long long) * 4 * (size_t)M.wgs));
    hipLaunchKernelGGL((spmv_csr_mring<kMringThreads, kMringNnzb, 4, kMringMaxB, false, true, false, true>), dim3(M.wgs), dim3(kMringThreads), 0, nullptr,
                       V, reinterpret_cast<const int4*>(M.d_plan), reinterpret_cast<const int4*>(M.d_first), M.d_ok, M.d_slots, d_x, d_y,
                       reinterpret_cast<const int2*>(M.d_rng), M.wgs, d);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, d, sizeof(long long) * 4 * (size_t)M.wgs, hipMemcpyDeviceToHost));
    dfree(d);
    *wgs_out = M.wgs;
    return MI_OK;
}

// placement experiments (tools/placement_lottery.py): move ONE device array of the handle's ring path to a fresh allocation (the old one
// is freed only after the new one exists, so the new one is different memory).  which: 0 coefficients, 1 ring slots, 2 row pointers,
// 3 ring plan records; how: 0 hipMalloc, 2 / 3 hipExtMallocWithFlags uncached / fine-grained (1 = contiguous is refused).  *old_ptr / *new_ptr: the
// addresses, for the record.
extern "C" int mi_debug_move_array(mi_csr_t A, int which, int how, unsigned long long* old_ptr, unsigned long long* new_ptr)
{
    // how = 1 (hipDeviceMallocContiguous) stays refused.  Round 3 moved the value array into such a buffer twice and both runs ended in
    // a GPU memory fault; the flag was blamed and refused, the logs were not kept (profiles/r03_contiguous_fault.txt).  What is known
    // from the code of that session: this function then copied exactly sizeof(double) * nnz bytes into an exactly-sized buffer, while the
    // ring kernels read up to kRingPadNnz values past the last nonzero (launch_ring_impl.hpp: "device arrays are padded for the kernel's
    // unclamped loads").  Behind a hipMalloc block the allocator's granule usually maps that tail; behind a physically contiguous
    // exactly-sized buffer it need not.  The missing pad explains the fault at least as well as the flag does; the move now carries
    // the pad (below), the flag has not been tried again.
    CHECK_ARG(A && which >= 0 && which <= 3 && (how == 0 || how == 2 || how == 3), "bad argument (how = 1, hipDeviceMallocContiguous, is refused: see the comment in devtools.hip)");
    if (A->inner) A = A->inner;
    void** slot = nullptr;
    size_t bytes = 0, pad_bytes = 0;
    if (which == 0) { slot = (void**)&A->d_coef; bytes = sizeof(double) * (size_t)A->nnz; pad_bytes = sizeof(double) * (size_t)kRingPadNnz; }
    if (which == 1) { slot = (void**)&A->ring.d_slots; bytes = sizeof(unsigned short) * (size_t)A->ring.nblk * A->ring.cfg.nnzb; }
    if (which == 2) { slot = (void**)&A->d_ptrow; bytes = sizeof(int) * ((size_t)A->n + 1); pad_bytes = sizeof(int) * (size_t)kRingPadRows; }
    if (which == 3) { slot = (void**)&A->ring.d_plan; bytes = sizeof(int) * 8 * (size_t)A->ring.nblk; }
    if (!*slot || bytes == 0) return fail(MI_ERR_STATE, "the handle has no such array");
    void* fresh = nullptr;
    if (how == 0) HIP_TRY(hipMalloc(&fresh, bytes + pad_bytes));
    else HIP_TRY(hipExtMallocWithFlags(&fresh, bytes + pad_bytes, how == 2 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    if (pad_bytes) HIP_TRY(hipMemset((char*)fresh + bytes, 0, pad_bytes)); // the zeroed tail the original allocation has (capi_csr.hip)
    HIP_TRY(hipMemcpy(fresh, *slot, bytes, hipMemcpyDeviceToDevice));
    HIP_TRY(hipDeviceSynchronize());
    if (old_ptr) *old_ptr = (unsigned long long)(uintptr_t)*slot;
    if (new_ptr) *new_ptr = (unsigned long long)(uintptr_t)fresh;
    void* old = *slot;
    *slot = fresh;
    HIP_TRY(hipFree(old));
    return MI_OK;
}

// run-length experiments on ONE placement (tools/mring_skew_ab.py): rebuild the handle's multi-window ring plan with the given skew and
// write it into the device arrays the handle already has (plan records and slots keep their size: the row blocks do not change; the small
// per-run tables are re-allocated).  Needs the handle's column indices on the device.
#include "mring_plan.hpp"
extern "C" int mi_debug_mring_replan(mi_csr_t A, int skew_pct, int* table_len, int* longest_run)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    MringTable& M = A->mring;
    if (!M.d_plan || !A->d_indcol) return fail(MI_ERR_STATE, "no multi-window ring plan (or no column indices) on this handle");
    std::vector<int> ind((size_t)A->nnz);
    HIP_TRY(hipMemcpy(ind.data(), A->d_indcol, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToHost));
    MringPlanHost P;
    build_mring_plan(A->n, A->h_ptrow.data(), ind.data(), P, 0, skew_pct);
    if (P.nblk != M.nblk) return fail(MI_ERR_STATE, "the new plan has other row blocks");
    if (const char* bad = check_mring_plan(P, A->n, A->h_ptrow.data(), ind.data())) return fail(MI_ERR_STATE, std::string("mring plan: ") + bad);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(M.d_plan, P.plan.data(), sizeof(int) * P.plan.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(M.d_slots, P.slots.data(), sizeof(unsigned short) * P.slots.size(), hipMemcpyHostToDevice));
    dfree(M.d_first); dfree(M.d_ok); dfree(M.d_rng);
    M.d_first = M.d_ok = M.d_rng = nullptr;
    HIP_TRY(hipMalloc(&M.d_first, sizeof(int) * P.first.size()));
    HIP_TRY(hipMalloc(&M.d_ok, sizeof(int) * P.run_ok.size()));
    HIP_TRY(hipMalloc(&M.d_rng, sizeof(int) * P.run_rng.size()));
    HIP_TRY(hipMemcpy(M.d_first, P.first.data(), sizeof(int) * P.first.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(M.d_ok, P.run_ok.data(), sizeof(int) * P.run_ok.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(M.d_rng, P.run_rng.data(), sizeof(int) * P.run_rng.size(), hipMemcpyHostToDevice));
    M.wgs = P.wgs;
    M.nruns = P.nruns;
    M.bpw = P.bpw;
    M.bad_runs = P.bad_runs;
    if (table_len) *table_len = P.wgs;
    if (longest_run) {
        *longest_run = 0;
        for (int g = 0; g < P.wgs; g++) *longest_run = std::max(*longest_run, P.run_rng[2 * g + 1] - P.run_rng[2 * g]);
    }
    return MI_OK;
}

// ---- timeline of ONE one-launch push step through the sliced stream (tools/sim_rank.py; MI355_PUSH_LOOPBACK arrangement) ----
extern "C" int mi_debug_part_push_trace(mi_part_t P, double* d_x_ext, double* d_y_local, int max_wgs, long long* host_out, int* wgs_out, int* halo_out)
{
    CHECK_ARG(P && d_x_ext && d_y_local && host_out && wgs_out, "null argument");
    if (!P->push_ready || !P->fused || !P->piece_all || resolve_kernel(P->piece_all) != MI_KERNEL_SSTREAM || !P->piece_all->ss.fusable)
        return fail(MI_ERR_STATE, "the one-launch push step of this handle does not run the sliced stream");
    mi_csr_t A = P->piece_all;
    SstreamTable& T = A->ss;
    CHECK_ARG(T.nwg <= max_wgs, "host buffer too small");
    const PartPlan& pl = P->plan;
    const unsigned step = ++P->push_step;
    RingComm C{};
    C.links = P->d_links;
    C.send_idx = P->d_send_idx;
    C.flags = P->win_flags;
    C.nb = P->d_nb;
    C.halo = P->win_data + (size_t)(step & 1u) * (size_t)(pl.n_halo > 0 ? pl.n_halo : 1);
    C.run_halo = P->d_run_halo;
    C.timeouts = P->d_timeouts;
    C.n_links = P->n_links;
    C.n_nb = P->n_nb;
    C.n_local = pl.n_local;
    C.n_left = pl.n_left;
    C.run_link = P->d_run_link;
    C.npush_runs = P->npush_runs;
    C.push_wgs = (P->npush_runs == 0 && P->n_links > 0) ? kNXCD : 0;
    C.step = step;
    C.spin_max = 1u << kPushSpinLog2Default;
    C.gate_push = P->ghost_readers ? 0 : 1;
    unsigned long long* d_tr = nullptr;
    HIP_TRY(hipMalloc(&d_tr, sizeof(unsigned long long) * 4 * (size_t)T.nwg));
    HIP_TRY(hipMemset(d_tr, 0, sizeof(unsigned long long) * 4 * (size_t)T.nwg));
    SsView S{T.dev.val, T.dev.slot, T.dev.wg, T.dev.win, T.nwg, A->n + T.shift, A->ncols, nullptr, T.shift};
    S.trace = d_tr;
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL((spmv_sstream_fused<8, false, 8>), dim3((unsigned)(T.nwg + C.push_wgs)), dim3(256), 0, nullptr, S, d_x_ext, d_y_local - T.shift, C);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, d_tr, sizeof(unsigned long long) * 4 * (size_t)T.nwg, hipMemcpyDeviceToHost));
    dfree(d_tr);
    *wgs_out = T.nwg;
    if (halo_out)
        for (int g = 0; g < T.nwg; g++) halo_out[g] = T.h_wg_halo[g] + 2 * (T.h_wg[g].link >= 0 ? 1 : 0) + 4 * (T.h_wg[g].r_end - T.h_wg[g].r_begin);
    return MI_OK;
}

// ---- timing experiments on the staged one-launch step (spmv_bcsr4_ext.hpp): leave parts of it out.  Results are WRONG afterwards. ----
// mode bits: 1 no workgroup waits for the exchange, 2 no push, 4 no window copy, 8 no wait for the neighbours' flags
extern "C" int mi_debug_part_ext_mode(mi_part_t P, int mode)
{
    CHECK_ARG(P, "null handle");
    if (!P->fused_ext) return fail(MI_ERR_STATE, "the handle does not run the staged one-launch step");
    HIP_TRY(hipDeviceSynchronize());
    P->ext_debug = mode;
    if ((mode & 1) && P->ext_csr) {
        std::vector<unsigned> o((size_t)P->n_ext_units);
        HIP_TRY(hipMemcpy(o.data(), P->d_ext_order, sizeof(unsigned) * o.size(), hipMemcpyDeviceToHost));
        for (unsigned& e : o) e &= 0x7fffffffu;
        HIP_TRY(hipMemcpy(P->d_ext_order, o.data(), sizeof(unsigned) * o.size(), hipMemcpyHostToDevice));
    } else if (mode & 1) {
        std::vector<int2> u((size_t)P->n_ext_units);
        HIP_TRY(hipMemcpy(u.data(), P->d_ext_units, sizeof(int2) * u.size(), hipMemcpyDeviceToHost));
        for (int2& e : u) e.y &= ~1;
        HIP_TRY(hipMemcpy(P->d_ext_units, u.data(), sizeof(int2) * u.size(), hipMemcpyHostToDevice));
    }
    return MI_OK;
}

// one traced launch of the staged step: host_out[3 g + {0, 1, 2}] = {start, wait over, end} of workgroup g (s_memrealtime ticks, 10 ns);
// modes_out[g]: -2 a pushing workgroup, -1 a copying one, else the unit's mode bits
extern "C" int mi_debug_part_ext_trace(mi_part_t P, double* d_x_ext, double* d_y_local, int max_wgs, long long* host_out, int* wgs_out, int* modes_out)
{
    CHECK_ARG(P && d_x_ext && d_y_local && host_out && wgs_out, "null argument");
    if (!P->fused_ext) return fail(MI_ERR_STATE, "the handle does not run the staged one-launch step");
    const int grid = P->n_ext_units + P->n_push_work + P->ext_wgs;
    CHECK_ARG(grid <= max_wgs, "host buffer too small");
    unsigned long long* d_tr = nullptr;
    HIP_TRY(hipMalloc(&d_tr, sizeof(unsigned long long) * 3 * (size_t)grid));
    HIP_TRY(hipMemset(d_tr, 0, sizeof(unsigned long long) * 3 * (size_t)grid));
    HIP_TRY(hipDeviceSynchronize());
    const unsigned step = ++P->push_step;
    int rc = part_ext_launch(P, d_x_ext, d_y_local, step, 1u << kPushSpinLog2Default, nullptr, d_tr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, d_tr, sizeof(unsigned long long) * 3 * (size_t)grid, hipMemcpyDeviceToHost));
    dfree(d_tr);
    *wgs_out = grid;
    if (modes_out && P->ext_csr) {
        std::vector<unsigned> o((size_t)P->n_ext_units);
        HIP_TRY(hipMemcpy(o.data(), P->d_ext_order, sizeof(unsigned) * o.size(), hipMemcpyDeviceToHost));
        for (int g = 0; g < grid; g++) modes_out[g] = g < P->n_push_work ? -2 : (g < P->n_push_work + P->ext_wgs ? -1 : (int)(o[g - P->n_push_work - P->ext_wgs] >> 31));
    } else if (modes_out) {
        std::vector<int2> u((size_t)P->n_ext_units);
        HIP_TRY(hipMemcpy(u.data(), P->d_ext_units, sizeof(int2) * u.size(), hipMemcpyDeviceToHost));
        for (int g = 0; g < grid; g++) modes_out[g] = g < P->n_push_work ? -2 : (g < P->n_push_work + P->ext_wgs ? -1 : u[g - P->n_push_work - P->ext_wgs].y);
    }
    return MI_OK;
}
