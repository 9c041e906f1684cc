// capi_part.hip: row-range partition, RCCL exchange, peer-push exchange — part of libmi355spmv.so (see capi_internal.hpp for the layout of the library).
// Built for gfx950 only; no CPU fallback anywhere: every compute entry point needs a HIP device.
#include "capi_internal.hpp"
#include "handoff_kernels.hpp"
#include "push_kernels.hpp"
#include "spmv_bcsr4_ext.hpp"
#include "rccl_loader.hpp"

// windows of ranks living in THIS process
static std::map<std::string, void*> g_win_registry;
static void part_comm_release(mi_part_s* P);
static int part_need_timeouts(mi_part_s* P);

// ---------------------------------------------------------------- RCCL, resolved at run time (rccl_loader.hpp)
static Rccl& g_rccl = rccl_state();
static_assert(MI_COMM_ID_BYTES == kCommIdBytes, "id size");

#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        int r_ = (expr);                                                                                \
        if (r_ != 0) return fail(MI_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));    \
    } while (0)

static void part_comm_release(mi_part_s* P)
{
    if (P->comm && g_rccl.ok) g_rccl.CommDestroy(P->comm);
    P->comm = nullptr;
    if (P->comm_stream) (void)hipStreamDestroy(P->comm_stream);
    if (P->ev_pack) (void)hipEventDestroy(P->ev_pack);
    if (P->ev_comm) (void)hipEventDestroy(P->ev_comm);
    P->comm_stream = nullptr;
    P->ev_pack = P->ev_comm = nullptr;
    if (P->d_sendbuf) dfree(P->d_sendbuf);
    dfree(P->d_ag_idx);
    dfree(P->d_ag_src);
    dfree(P->d_ag_send);
    dfree(P->d_ag_recv);
    P->d_ag_idx = P->d_ag_src = nullptr;
    P->d_ag_send = P->d_ag_recv = nullptr;
    P->ag_ready = P->ag_use = false;
    if (P->d_flags) dfree(P->d_flags);
    if (P->h_timeouts) (void)hipHostFree(P->h_timeouts);
    for (void* m : P->ipc_opened) (void)hipIpcCloseMemHandle(m);
    P->ipc_opened.clear();
    if (P->win_registered) {
        std::lock_guard<std::mutex> lock(g_mu);
        g_win_registry.erase(P->win_key);
        P->win_registered = false;
    }
    dfree(P->win);
    dfree(P->d_links);
    dfree(P->d_push_work);
    dfree(P->d_link_chunks);
    dfree(P->d_tickets);
    P->d_push_work = nullptr;
    P->d_link_chunks = nullptr;
    P->d_tickets = nullptr;
    dfree(P->d_nb);
    dfree(P->d_run_link);
    dfree(P->d_stage);
    dfree(P->d_ready);
    dfree(P->d_ext_units);
    dfree(P->d_ext_order);
    P->d_stage = nullptr;
    P->d_ready = nullptr;
    P->d_ext_units = nullptr;
    P->d_ext_order = nullptr;
    P->ext_csr = false;
    mi_csr_destroy(P->piece_all);
    P->piece_all = nullptr;
    P->d_run_link = nullptr;
    P->fused = P->fused_ext = false;
    P->win = nullptr;
    P->d_links = nullptr;
    P->d_nb = nullptr;
    P->push_ready = false;
    P->d_sendbuf = nullptr;
    P->d_flags = nullptr;
    P->h_timeouts = P->d_timeouts = nullptr;
}

// A hand-off wait that gave up means every result since is invalid: sticky, reported by every later call.
static int part_handoff_status(const mi_part_s* P)
{
    if (P->h_timeouts && __atomic_load_n(P->h_timeouts, __ATOMIC_ACQUIRE) != 0)
        return fail(MI_ERR_HIP, "mi_part: a stream hand-off timed out (a peer rank stalled or died); results since then are invalid");
    return MI_OK;
}

extern "C" int mi_part_status(mi_part_t P)
{
    CHECK_ARG(P, "null handle");
    return part_handoff_status(P);
}

extern "C" int mi_comm_available(void)
{
    if (!rccl_load()) return fail(MI_ERR_UNSUPPORTED, "RCCL unavailable: " + g_rccl.why);
    return MI_OK;
}

extern "C" int mi_comm_unique_id(void* id128)
{
    CHECK_ARG(id128, "null id");
    if (!rccl_load()) return fail(MI_ERR_UNSUPPORTED, "RCCL unavailable: " + g_rccl.why);
    NCCL_TRY(g_rccl.GetUniqueId(id128));
    return MI_OK;
}

// the exchange of one step, enqueued on cs: send my packed entries to every peer that
// needs some, receive my ghosts straight into x_ext's halo region (contiguous per owner)
// d_x_direct != nullptr: every send list is a contiguous slice of the owned x (PartPlan::sends_contiguous) and is
// sent from there, no packed copy
static int enqueue_exchange(const PartPlan& pl, void* comm, const double* d_sendbuf, double* d_halo, hipStream_t cs,
                            const double* d_x_direct = nullptr)
{
    NCCL_TRY(g_rccl.GroupStart());
    for (int p = 0; p < pl.nranks; p++) {
        if (pl.send_counts[p]) {
            const double* src = d_x_direct ? d_x_direct + pl.send_lists[p][0] : d_sendbuf + pl.send_offsets[p];
            NCCL_TRY(g_rccl.Send(src, (size_t)pl.send_counts[p], kNcclDouble, p, comm, cs));
        }
        if (pl.recv_counts[p])
            NCCL_TRY(g_rccl.Recv(d_halo + pl.recv_offsets[p], (size_t)pl.recv_counts[p], kNcclDouble, p, comm, cs));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return MI_OK;
}

extern "C" int mi_comm_selftest(int count, double* max_abs_err)
{
    CHECK_ARG(count > 0 && max_abs_err, "bad argument");
    int rc = need_device();
    if (rc) return rc;
    if (!rccl_load()) return fail(MI_ERR_UNSUPPORTED, "RCCL unavailable: " + g_rccl.why);
    IdByValue id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    void* comm = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&comm, 1, id, 0));
    PartPlan pl; // a 1-rank "partition" that sends `count` entries to itself
    pl.nranks = 1;
    pl.rank = 0;
    pl.send_counts = {count};
    pl.send_offsets = {0, count};
    pl.recv_counts = {count};
    pl.recv_offsets = {0, count};
    std::vector<double> h((size_t)count), back((size_t)count);
    for (int i = 0; i < count; i++) h[i] = 0.5 * i - 3.0;
    double *d_src = nullptr, *d_dst = nullptr;
    hipStream_t s0 = nullptr, cs = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipMalloc(&d_src, sizeof(double) * count));
    HIP_TRY(hipMalloc(&d_dst, sizeof(double) * count));
    HIP_TRY(hipStreamCreate(&s0));
    HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(d_src, h.data(), sizeof(double) * count, hipMemcpyHostToDevice, s0));
    HIP_TRY(hipMemsetAsync(d_dst, 0, sizeof(double) * count, s0));
    HIP_TRY(hipEventRecord(e0, s0));
    HIP_TRY(hipStreamWaitEvent(cs, e0, 0));
    rc = enqueue_exchange(pl, comm, d_src, d_dst, cs);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e1, cs));
    HIP_TRY(hipStreamWaitEvent(s0, e1, 0));
    HIP_TRY(hipMemcpyAsync(back.data(), d_dst, sizeof(double) * count, hipMemcpyDeviceToHost, s0));
    HIP_TRY(hipStreamSynchronize(s0));
    double m = 0.0;
    for (int i = 0; i < count; i++) m = std::max(m, std::fabs(back[i] - h[i]));
    *max_abs_err = m;
    g_rccl.CommDestroy(comm);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(cs);
    (void)hipStreamDestroy(s0);
    dfree(d_src);
    dfree(d_dst);
    return MI_OK;
}

// ---------------------------------------------------------------- partition
extern "C" int mi_part_create(int nranks, int rank, const long long* row_starts, const int* ptrow,
                              const int* indcol_global, const double* coef, mi_part_t* out)
{
    CHECK_ARG(out, "out is null");
    *out = nullptr;
    mi_part_t P = new (std::nothrow) mi_part_s();
    if (!P) return fail(MI_ERR_ALLOC, "host allocation failed");
    std::string err = P->plan.build(nranks, rank, row_starts, ptrow, indcol_global, coef);
    if (!err.empty()) {
        delete P;
        return fail(MI_ERR_ARG, "mi_part_create: " + err);
    }
    *out = P;
    return MI_OK;
}

extern "C" int mi_part_destroy(mi_part_t P)
{
    if (!P) return MI_OK;
    int status = MI_OK;
    if (P->comm_stream || P->push_ready) { // let queued steps finish, then report a wait that gave up during them
        if (P->comm_stream) (void)hipStreamSynchronize(P->comm_stream);
        else (void)hipDeviceSynchronize();
        status = part_handoff_status(P);
    }
    mi_csr_destroy(P->piece[0]);
    mi_csr_destroy(P->piece[1]);
    if (P->d_send_idx) dfree(P->d_send_idx);
    part_comm_release(P);
    delete P;
    return status;
}

extern "C" int mi_part_sizes(mi_part_t P, int* n_local, int* n_halo, int* n_interior_rows, int* n_boundary_rows)
{
    CHECK_ARG(P, "null handle");
    if (n_local) *n_local = P->plan.n_local;
    if (n_halo) *n_halo = P->plan.n_halo;
    if (n_interior_rows) *n_interior_rows = (int)P->plan.piece[0].rowmap.size();
    if (n_boundary_rows) *n_boundary_rows = (int)P->plan.piece[1].rowmap.size();
    return MI_OK;
}

extern "C" int mi_part_recv_counts(mi_part_t P, int* counts)
{
    CHECK_ARG(P && counts, "null argument");
    for (int p = 0; p < P->plan.nranks; p++) counts[p] = P->plan.recv_counts[p];
    return MI_OK;
}

extern "C" int mi_part_recv_ids(mi_part_t P, int peer, long long* ids)
{
    CHECK_ARG(P && peer >= 0 && peer < P->plan.nranks, "bad peer");
    const int c = P->plan.recv_counts[peer];
    CHECK_ARG(c == 0 || ids, "null ids");
    for (int i = 0; i < c; i++) ids[i] = P->plan.halo_ids[P->plan.recv_offsets[peer] + i];
    return MI_OK;
}

extern "C" int mi_part_set_send_ids(mi_part_t P, int peer, int count, const long long* ids)
{
    CHECK_ARG(P, "null handle");
    if (P->finalized) return fail(MI_ERR_STATE, "partition already finalized");
    std::string err = P->plan.set_send(peer, count, ids);
    if (!err.empty()) return fail(MI_ERR_ARG, "mi_part_set_send_ids: " + err);
    return MI_OK;
}

extern "C" int mi_part_send_counts(mi_part_t P, int* counts)
{
    CHECK_ARG(P && counts, "null argument");
    for (int p = 0; p < P->plan.nranks; p++) counts[p] = P->plan.send_counts[p];
    return MI_OK;
}

extern "C" int mi_part_local_csr(mi_part_t P, int which, int* nrows, const int** ptrow, const int** indcol_local,
                                 const double** coef, const int** rowmap)
{
    CHECK_ARG(P && (which == 0 || which == 1 || which == 2), "bad argument");
    if (which == 2) P->plan.build_combined(); // all rows, natural order, columns [ghosts in front | owned | ghosts behind]
    const LocalPiece& L = which == 2 ? P->plan.all : P->plan.piece[which];
    if (nrows) *nrows = which == 2 ? P->plan.n_local : (int)L.rowmap.size();
    if (ptrow) *ptrow = L.ptrow.data();
    if (indcol_local) *indcol_local = L.indcol.data();
    if (coef) *coef = L.coef.data();
    if (rowmap) *rowmap = which == 2 ? nullptr : L.rowmap.data();
    return MI_OK;
}

extern "C" int mi_part_combined_info(mi_part_t P, int* n_left)
{
    CHECK_ARG(P && n_left, "null argument");
    P->plan.build_combined();
    *n_left = P->plan.n_left;
    return MI_OK;
}

// 1 when every send list is a run of consecutive local ids (banded partitions): the RCCL step sends slices of x in place, no pack kernel
extern "C" int mi_part_sends_contiguous(mi_part_t P, int* contiguous)
{
    CHECK_ARG(P && contiguous, "null argument");
    *contiguous = P->plan.sends_set && P->plan.sends_contiguous ? 1 : 0;
    return MI_OK;
}

extern "C" int mi_part_send_index(mi_part_t P, int* total, const int** local_idx)
{
    CHECK_ARG(P, "null handle");
    if (total) *total = (int)P->plan.send_idx.size();
    if (local_idx) *local_idx = P->plan.send_idx.data();
    return MI_OK;
}

extern "C" int mi_part_finalize(mi_part_t P)
{
    CHECK_ARG(P, "null handle");
    if (P->finalized) return MI_OK;
    if (!P->plan.sends_set && P->plan.nranks > 1)
        return fail(MI_ERR_STATE, "mi_part_set_send_ids was never called (exchange the recv ids first)");
    int rc = need_device();
    if (rc) return rc;
    const int ncols = P->plan.n_local + P->plan.n_halo;
    for (int w = 0; w < 2; w++) {
        const LocalPiece& L = P->plan.piece[w];
        rc = mi_csr_create_mapped((int)L.rowmap.size(), ncols, L.ptrow.data(), L.indcol.data(), L.coef.data(),
                                  L.rowmap.data(), &P->piece[w]);
        if (rc) return rc;
        if (P->kernel == MI_KERNEL_SSTREAM) (void)mi_csr_set_kernel(P->piece[w], P->kernel); // builds the sliced copy where the piece is eligible
        P->piece[w]->kernel = P->kernel;
    }
    const size_t ns = P->plan.send_idx.size();
    if (ns) {
        HIP_TRY(hipMalloc(&P->d_send_idx, sizeof(int) * ns));
        HIP_TRY(hipMemcpy(P->d_send_idx, P->plan.send_idx.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
    }
    P->finalized = true;
    return MI_OK;
}

extern "C" int mi_part_comm_init(mi_part_t P, const void* id128)
{
    CHECK_ARG(P && id128, "null argument");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    if (P->comm) return MI_OK;
    if (!rccl_load()) return fail(MI_ERR_UNSUPPORTED, "RCCL unavailable: " + g_rccl.why);
    IdByValue id;
    memcpy(&id, id128, sizeof id);
    NCCL_TRY(g_rccl.CommInitRank(&P->comm, P->plan.nranks, id, P->plan.rank));
    HIP_TRY(hipStreamCreateWithFlags(&P->comm_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&P->ev_pack, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&P->ev_comm, hipEventDisableTiming));
    const size_t ns = P->plan.send_idx.size();
    HIP_TRY(hipMalloc(&P->d_sendbuf, sizeof(double) * (ns ? ns : 1)));
    HIP_TRY(hipMalloc(&P->d_flags, 4 * sizeof(unsigned)));
    HIP_TRY(hipMemset(P->d_flags, 0, 4 * sizeof(unsigned)));
    HIP_TRY(hipHostMalloc((void**)&P->h_timeouts, sizeof(unsigned), hipHostMallocMapped));
    *P->h_timeouts = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&P->d_timeouts, P->h_timeouts, 0));
    P->step_no = 0;
    if (const char* e = getenv("MI355_PART_HANDOFF")) P->flag_handoff = strcmp(e, "flags") == 0;
    return MI_OK;
}

extern "C" int mi_part_comm_info(mi_part_t P, int* comm_ranks, int* comm_rank)
{
    CHECK_ARG(P, "null handle");
    if (comm_ranks) *comm_ranks = 0;
    if (comm_rank) *comm_rank = -1;
    if (!P->comm) return MI_OK; // no communicator: zeros
    if (comm_ranks && g_rccl.CommCount) NCCL_TRY(g_rccl.CommCount(P->comm, comm_ranks));
    if (comm_rank && g_rccl.CommUserRank) NCCL_TRY(g_rccl.CommUserRank(P->comm, comm_rank));
    return MI_OK;
}

extern "C" int mi_part_spmv_dev(mi_part_t P, double* d_x_ext, double* d_y_local, mi_stream_t s_)
{
    CHECK_ARG(P, "null handle");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    hipStream_t s = (hipStream_t)s_;
    const PartPlan& pl = P->plan;
    int rc;
    if (pl.nranks == 1) { // no halo: the two pieces back to back on the caller's stream
        if ((rc = mi_spmv_dev(P->piece[0], d_x_ext, d_y_local, s))) return rc;
        return mi_spmv_dev(P->piece[1], d_x_ext, d_y_local, s);
    }
    if (!P->comm) return fail(MI_ERR_STATE, "mi_part_comm_init was not called");
    // Two concurrent chains:
    //   comm stream:      pack -> exchange -> boundary rows (they need the halo, nothing else)
    //   caller's stream:  interior rows (they need only owned x)
    // joined at the end.  The first hand-off orders the comm chain behind everything already queued
    // on s: the producer of x, and the previous step (whose boundary rows read the halo region this
    // exchange overwrites, and which joined s with its own closing hand-off).  The two row sets are
    // disjoint in y.  On an 8-rank piece the boundary kernel (~4 us, mostly launch latency) and the
    // exchange (~7 us) thus hide behind the interior kernel (~23 us).  The hand-offs are HIP events, or flag
    // kernels (handoff_kernels.hpp) with MI355_PART_HANDOFF=flags.
    if ((rc = part_handoff_status(P))) return rc; // a plain load of pinned memory: no copy, no synchronisation
    const unsigned step = ++P->step_no;
    if (P->flag_handoff) {
        hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(64), 0, s, P->d_flags, step);
        hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, P->comm_stream, P->d_flags, step, P->d_timeouts);
    } else {
        HIP_TRY(hipEventRecord(P->ev_pack, s));
        HIP_TRY(hipStreamWaitEvent(P->comm_stream, P->ev_pack, 0));
    }
    if (P->ag_use && P->ag_ready) {
        // wide halos (an FE slab's boundary planes): ONE ncclAllGather of fixed-size slices instead of a grouped send / recv per
        // neighbour — north_star's collective.  My slice = the entries anybody needs from me; every ghost is then picked out of
        // the gathered N x M buffer into the halo part of x (16 bytes of index arithmetic per ghost, no per-peer offsets).
        if ((rc = mi_gather_dev(P->ag_slice, P->d_ag_idx, d_x_ext, P->d_ag_send, P->comm_stream))) return rc;
        NCCL_TRY(g_rccl.AllGather(P->d_ag_send, P->d_ag_recv, (size_t)P->ag_slice, kNcclDouble, P->comm, P->comm_stream));
        if ((rc = mi_gather_dev(pl.n_halo, P->d_ag_src, P->d_ag_recv, d_x_ext + pl.n_local, P->comm_stream))) return rc;
    } else if (pl.sends_contiguous) { // banded partitions: the neighbours' ghosts are slices of x, sent in place
        if ((rc = enqueue_exchange(pl, P->comm, nullptr, d_x_ext + pl.n_local, P->comm_stream, d_x_ext))) return rc;
    } else {
        if ((rc = mi_gather_dev((int)pl.send_idx.size(), P->d_send_idx, d_x_ext, P->d_sendbuf, P->comm_stream))) return rc;
        if ((rc = enqueue_exchange(pl, P->comm, P->d_sendbuf, d_x_ext + pl.n_local, P->comm_stream))) return rc;
    }
    if ((rc = mi_spmv_dev(P->piece[1], d_x_ext, d_y_local, P->comm_stream))) return rc;
    if (P->flag_handoff) hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(64), 0, P->comm_stream, P->d_flags + 1, step);
    else HIP_TRY(hipEventRecord(P->ev_comm, P->comm_stream));
    if ((rc = mi_spmv_dev(P->piece[0], d_x_ext, d_y_local, s))) return rc;
    if (P->flag_handoff) hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, s, P->d_flags + 1, step, P->d_timeouts);
    else HIP_TRY(hipStreamWaitEvent(s, P->ev_comm, 0));
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// ---- the all-gather form of the RCCL exchange ---------------------------------------------------------------------------------
extern "C" int mi_part_send_union(mi_part_t P, int* count, const int** local_idx)
{
    CHECK_ARG(P && count, "null argument");
    if (!P->plan.sends_set) return fail(MI_ERR_STATE, "mi_part_set_send_ids was never called");
    if (P->ag_union.empty() && !P->plan.send_idx.empty()) {
        P->ag_union = P->plan.send_idx;
        std::sort(P->ag_union.begin(), P->ag_union.end());
        P->ag_union.erase(std::unique(P->ag_union.begin(), P->ag_union.end()), P->ag_union.end());
    }
    *count = (int)P->ag_union.size();
    if (local_idx) *local_idx = P->ag_union.data();
    return MI_OK;
}

extern "C" int mi_part_allgather_setup(mi_part_t P, const int* counts, const long long* ids)
{
    CHECK_ARG(P && counts, "null argument");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    if (!rccl_load() || !g_rccl.AllGather) return fail(MI_ERR_UNSUPPORTED, "ncclAllGather is not available in the resolved RCCL");
    const PartPlan& pl = P->plan;
    const int R = pl.nranks;
    int mine = 0;
    int rc = mi_part_send_union(P, &mine, nullptr);
    if (rc) return rc;
    if (counts[pl.rank] != mine) return fail(MI_ERR_ARG, "counts[rank] is not this rank's union (mi_part_send_union)");
    int M = 1;
    std::vector<size_t> off((size_t)R + 1, 0);
    for (int p = 0; p < R; p++) {
        CHECK_ARG(counts[p] >= 0, "negative count");
        M = std::max(M, counts[p]);
        off[p + 1] = off[p] + (size_t)counts[p];
    }
    CHECK_ARG(off[R] == 0 || ids, "null ids");
    std::vector<int> src((size_t)std::max(pl.n_halo, 1), 0);
    for (int p = 0; p < R; p++)
        for (int i = 0; i < pl.recv_counts[p]; i++) {
            const long long g = pl.halo_ids[pl.recv_offsets[p] + i];
            const long long* b = ids + off[p];
            const long long* e = b + counts[p];
            const long long* it = std::lower_bound(b, e, g);
            if (it == e || *it != g) return fail(MI_ERR_STATE, "a ghost of this rank is missing from its owner's union: the lists are not those of this partition");
            src[pl.recv_offsets[p] + i] = p * M + (int)(it - b);
        }
    std::vector<int> idx((size_t)M, 0); // padded with entry 0 (n_local > 0 whenever anything is sent; an empty rank gathers x[0] of a 1-entry buffer)
    for (int i = 0; i < mine; i++) idx[i] = P->ag_union[i];
    dfree(P->d_ag_idx); dfree(P->d_ag_src); dfree(P->d_ag_send); dfree(P->d_ag_recv);
    P->d_ag_idx = P->d_ag_src = nullptr;
    P->d_ag_send = P->d_ag_recv = nullptr;
    HIP_TRY(hipMalloc(&P->d_ag_idx, sizeof(int) * idx.size()));
    HIP_TRY(hipMalloc(&P->d_ag_src, sizeof(int) * src.size()));
    HIP_TRY(hipMalloc(&P->d_ag_send, sizeof(double) * (size_t)M));
    HIP_TRY(hipMalloc(&P->d_ag_recv, sizeof(double) * (size_t)M * R));
    HIP_TRY(hipMemcpy(P->d_ag_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P->d_ag_src, src.data(), sizeof(int) * src.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(P->d_ag_send, 0, sizeof(double) * (size_t)M));
    P->ag_slice = M;
    P->ag_ready = true;
    return MI_OK;
}

extern "C" int mi_part_set_allgather(mi_part_t P, int on)
{
    CHECK_ARG(P, "null handle");
    if (on && !P->ag_ready) return fail(MI_ERR_STATE, "mi_part_allgather_setup was not called");
    P->ag_use = on != 0;
    return MI_OK;
}

extern "C" int mi_part_allgather_info(mi_part_t P, int* ready, int* in_use, int* slice)
{
    CHECK_ARG(P, "null handle");
    if (ready) *ready = P->ag_ready ? 1 : 0;
    if (in_use) *in_use = P->ag_use && P->ag_ready ? 1 : 0;
    if (slice) *slice = P->ag_slice;
    return MI_OK;
}

// ---- peer-push exchange (push_exchange.hpp) ----------------------------------------------------------------------
static int part_need_timeouts(mi_part_s* P)
{
    if (P->h_timeouts) return MI_OK;
    HIP_TRY(hipHostMalloc((void**)&P->h_timeouts, sizeof(unsigned), hipHostMallocMapped));
    *P->h_timeouts = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&P->d_timeouts, P->h_timeouts, 0));
    return MI_OK;
}

// my receive window (allocated once): uncached device memory, nranks flag slots + two parities of n_halo doubles
int part_push_window(mi_part_s* P)
{
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    const PartPlan& pl = P->plan;
    if (!P->win) {
        const size_t bytes = win_data_offset(pl.nranks) + sizeof(double) * 2 * (size_t)(pl.n_halo > 0 ? pl.n_halo : 1);
        // uncached: neither the writer's nor the reader's L2 may keep a line of the window
        // (no fallback to cached memory: the receiver reads the window with plain loads and relies on no cache holding a line
        // of it — a neighbour's writes over xGMI would not update this GPU's L2.  Without uncached memory the push exchange is
        // refused and DistCSR falls back to the RCCL or torch.distributed exchange.)
        hipError_t e = hipExtMallocWithFlags(&P->win, bytes, hipDeviceMallocUncached);
        P->win_uncached = e == hipSuccess;
        if (e != hipSuccess) {
            (void)hipGetLastError();
            P->win = nullptr;
            return fail(MI_ERR_UNSUPPORTED, std::string("peer push needs uncached device memory (hipExtMallocWithFlags): ") + hipGetErrorString(e));
        }
        HIP_TRY(hipMemset(P->win, 0, bytes));
        HIP_TRY(hipDeviceSynchronize());
        P->win_flags = (unsigned*)P->win;
        P->win_data = (double*)((char*)P->win + win_data_offset(pl.nranks));
    }
    return MI_OK;
}

void part_push_layout(const mi_part_s* P, long long* layout)
{
    const PartPlan& pl = P->plan;
    layout[0] = pl.n_halo;
    for (int p = 0; p < pl.nranks; p++) {
        layout[1 + p] = pl.recv_offsets[p];
        layout[1 + pl.nranks + p] = pl.recv_counts[p];
    }
}

extern "C" int mi_part_push_export(mi_part_t P, void* handle64, long long* layout)
{
    CHECK_ARG(P && handle64 && layout, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == MI_IPC_HANDLE_BYTES, "IPC handle size");
    int rc = part_push_window(P);
    if (rc) return rc;
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, P->win));
    memcpy(handle64, &h, sizeof h);
    {
        std::lock_guard<std::mutex> lock(g_mu);
        P->win_key.assign((const char*)&h, sizeof h);
        g_win_registry[P->win_key] = P->win;
        P->win_registered = true;
    }
    part_push_layout(P, layout);
    return MI_OK;
}

extern "C" int mi_part_push_connect(mi_part_t P, const void* handles, const long long* layouts)
{
    CHECK_ARG(P && handles && layouts, "null argument");
    if (!P->win) return fail(MI_ERR_STATE, "mi_part_push_export was not called");
    if (P->push_ready) return MI_OK;
    const PartPlan& pl = P->plan;
    const int R = pl.nranks, me = pl.rank;
    // MI355_PUSH_LOOPBACK=1 (tools/sim_rank.py only): a handle may map windows of its own process — one rank's step timed on
    // one GPU with its pushes looped back and every flag preset, so that nothing ever waits
    const bool loopback = getenv("MI355_PUSH_LOOPBACK") && !strcmp(getenv("MI355_PUSH_LOOPBACK"), "1");
    std::vector<void*> bases((size_t)R, nullptr);
    for (int p = 0; p < R; p++) {
        if (p == me || (pl.send_counts[p] == 0 && pl.recv_counts[p] == 0)) continue; // not a neighbour
        void* base = nullptr;
        const std::string key((const char*)handles + (size_t)p * MI_IPC_HANDLE_BYTES, MI_IPC_HANDLE_BYTES);
        {
            std::lock_guard<std::mutex> lock(g_mu);
            auto it = g_win_registry.find(key);
            if (it != g_win_registry.end()) base = it->second; // a window of this very process
        }
        if (base && !loopback)
            // Ranks as threads of one process ON ONE DEVICE cannot use this exchange: their streams share the process's few hardware
            // queues (GPU_MAX_HW_QUEUES, 4 by default), a queue runs its kernels in order, and a kernel that spins on a peer's flag can
            // sit in the queue in front of the very kernel that would raise it — seen as a hang of four rank threads
            // (gpurun_out/t_dist.log, round 2).  Nothing the library does can order another rank's launches, so it refuses.
            // (mi_dist_*, capi_dist.hip, connects windows of one process that live on DIFFERENT devices — separate queues — through
            // part_push_connect_bases directly.)
            return fail(MI_ERR_UNSUPPORTED, "peer push needs one PROCESS per rank: a peer's window belongs to this process "
                                            "(rank threads share hardware queues and can deadlock in the wait); use the RCCL or torch exchange");
        if (!base) {
            hipIpcMemHandle_t h;
            memcpy(&h, key.data(), sizeof h);
            HIP_TRY(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
            P->ipc_opened.push_back(base);
            // another process's window on THIS device: the ranks share a card (forms that wait in many workgroups step down: part_ext_launch)
            hipPointerAttribute_t at;
            int dev = -1;
            if (hipPointerGetAttributes(&at, base) == hipSuccess && hipGetDevice(&dev) == hipSuccess && at.device == dev) P->peer_on_my_device = true;
            (void)hipGetLastError();
        }
        bases[p] = base;
    }
    return part_push_connect_bases(P, bases.data(), layouts);
}

// bases[p] = the address at which THIS rank's device reaches peer p's receive window (IPC mapping, or — ranks of one process on
// different devices with peer access enabled — the peer's own pointer); null for ranks that are not neighbours
int part_push_connect_bases(mi_part_s* P, void* const* bases, const long long* layouts)
{
    if (!P->win) return fail(MI_ERR_STATE, "no receive window");
    if (P->push_ready) return MI_OK;
    const PartPlan& pl = P->plan;
    const int R = pl.nranks, me = pl.rank, LW = 2 * R + 1;
    int rc = part_need_timeouts(P);
    if (rc) return rc;
    std::vector<PushLink> links;
    std::vector<int> nb;
    for (int p = 0; p < R; p++) {
        if (p == me) continue;
        const long long* Lp = layouts + (size_t)p * LW;
        const long long peer_nhalo = Lp[0], peer_off = Lp[1 + me], peer_cnt = Lp[1 + R + me];
        if (peer_cnt != pl.send_counts[p]) return fail(MI_ERR_STATE, "peer expects a different number of entries than this rank sends");
        if (pl.send_counts[p] == 0 && pl.recv_counts[p] == 0) continue; // not a neighbour
        nb.push_back(p);
        void* base = bases[p];
        if (!base) return fail(MI_ERR_ARG, "no window address for a neighbour");
        double* pdata = (double*)((char*)base + win_data_offset(R));
        PushLink L;
        L.dst[0] = pdata + peer_off;
        L.dst[1] = pdata + (peer_nhalo > 0 ? peer_nhalo : 1) + peer_off;
        L.flag = (unsigned*)base + (size_t)me * kWinFlagStride;
        L.send_off = pl.send_offsets[p];
        L.count = pl.send_counts[p];
        L.first = -1;
        if (L.count > 0) {
            bool contiguous = true;
            for (int i = 1; i < L.count && contiguous; i++) contiguous = pl.send_lists[p][i] == pl.send_lists[p][0] + i;
            if (contiguous) L.first = pl.send_lists[p][0];
        }
        links.push_back(L);
    }
    P->n_links = (int)links.size();
    P->n_nb = (int)nb.size();
    if (P->n_links) {
        std::vector<int2> work;
        std::vector<int> chunks(links.size());
        for (size_t l = 0; l < links.size(); l++) {
            chunks[l] = std::max(1, (links[l].count + kPushChunk - 1) / kPushChunk);
            for (int ch = 0; ch < chunks[l]; ch++) work.push_back(make_int2((int)l, ch));
        }
        P->n_push_work = (int)work.size();
        HIP_TRY(hipMalloc(&P->d_push_work, sizeof(int2) * work.size()));
        HIP_TRY(hipMemcpy(P->d_push_work, work.data(), sizeof(int2) * work.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&P->d_link_chunks, sizeof(int) * chunks.size()));
        HIP_TRY(hipMemcpy(P->d_link_chunks, chunks.data(), sizeof(int) * chunks.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&P->d_tickets, sizeof(unsigned) * links.size()));
        HIP_TRY(hipMemset(P->d_tickets, 0, sizeof(unsigned) * links.size()));
        HIP_TRY(hipMalloc(&P->d_links, sizeof(PushLink) * links.size()));
        HIP_TRY(hipMemcpy(P->d_links, links.data(), sizeof(PushLink) * links.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&P->d_nb, sizeof(int) * nb.size()));
        HIP_TRY(hipMemcpy(P->d_nb, nb.data(), sizeof(int) * nb.size(), hipMemcpyHostToDevice));
    }
    P->push_step = 0;
    P->push_ready = true;
    // The one-launch step needs all local rows as ONE piece that the ring kernel serves (the push duty and the ghost
    // reads live in that kernel).  MI355_PUSH_FUSED=0 keeps the four-launch form.
    const char* fe = getenv("MI355_PUSH_FUSED");
    // Ranks whose rows have the 4x4 node structure (FE): the piece numbered as x_ext is, the exchange in front of the grid, the ghosts
    // staged once per step (spmv_bcsr4_ext.hpp).  MI355_PUSH_FUSED_EXT=0 keeps the four-launch form for them.  (Up to round 5 halos of
    // <= 16 384 ghosts took a form that read every ghost use from the uncached window, spmv_bcsr4_fused: 14.5 against 7.8 us per step for
    // a 41^3-node box at N = 4, 8.5 against 6.8 for a 31^3-node one — removed.)
    const char* fx = getenv("MI355_PUSH_FUSED_EXT");
    if (!(fe && !strcmp(fe, "0")) && !(fx && !strcmp(fx, "0")) && P->kernel == MI_KERNEL_AUTO && pl.n_local > 0 && pl.n_local % 4 == 0 && pl.n_halo % 4 == 0 &&
        P->n_links > 0) {
        P->plan.build_all_ext();
        const LocalPiece& L = P->plan.all_ext;
        if (csr_has_block4_pattern(pl.n_local, L.ptrow.data(), L.indcol.data())) {
            rc = csr_create_impl(pl.n_local, pl.n_local + pl.n_halo, L.ptrow.data(), L.indcol.data(), L.coef.data(), nullptr, &P->piece_all);
            if (rc) return rc;
            mi_csr_t A = P->piece_all;
            // (the blocked copy is enough: whether this rank's create-time measurement put a CSR kernel a hair ahead of the blocked one must
            // not decide — the one-launch step only happens if EVERY rank has it, mi_part_push_unfuse)
            if (A->blocked) {
                const int nbr = pl.n_local / 4, per = kWG / 4, nwg = (nbr + per - 1) / per;
                std::vector<int> wg_halo((size_t)nwg, 0);
                for (int w = 0; w < nwg; w++) {
                    const int r0 = 4 * w * per, r1 = std::min(pl.n_local, 4 * (w + 1) * per);
                    for (int k = L.ptrow[r0]; k < L.ptrow[r1] && !wg_halo[w]; k++) wg_halo[w] = L.indcol[k] >= pl.n_local;
                }
                // a unit that waits starts late: four workgroups of 16 block rows, 16 lanes per row (spmv_bcsr4_ext.hpp); the others 64 rows, 4 lanes
                // (MI355_PUSH_EXT_LANES16=0 none / 2 all of them: A/B)
                const char* l16 = getenv("MI355_PUSH_EXT_LANES16");
                const int l16mode = l16 ? atoi(l16) : 1;
                std::vector<int2> units; // the units that wait LAST: dispatched behind the others, and a launch of their own in the two-launch form
                for (int pass = 0; pass < 2; pass++) {
                    for (int w = 0; w < nwg; w++) {
                        if ((wg_halo[w] != 0) != (pass == 1)) continue;
                        const bool wide = l16mode == 2 || (l16mode == 1 && wg_halo[w]);
                        if (!wide) units.push_back(make_int2(w * per, wg_halo[w] ? 1 : 0));
                        else
                            for (int i = 0; i < 4; i++)
                                if (w * per + 16 * i < nbr) units.push_back(make_int2(w * per + 16 * i, 2 | (wg_halo[w] ? 1 : 0)));
                    }
                    if (pass == 0) P->n_ext_plain = (int)units.size();
                }
                // ranks sharing a card: two launches (spmv_bcsr4_ext.hpp says why); MI355_PUSH_EXT_SPLIT=0|1 forces (sim_rank.py loops its pushes back
                // into its own window and asks for 0)
                P->ext_split = P->peer_on_my_device;
                if (const char* e = getenv("MI355_PUSH_EXT_SPLIT")) P->ext_split = atoi(e) != 0;
                P->n_ext_units = (int)units.size();
                HIP_TRY(hipMalloc(&P->d_ext_units, sizeof(int2) * units.size()));
                HIP_TRY(hipMemcpy(P->d_ext_units, units.data(), sizeof(int2) * units.size(), hipMemcpyHostToDevice));
                HIP_TRY(hipMalloc(&P->d_stage, sizeof(double) * (size_t)std::max(pl.n_halo, 2)));
                HIP_TRY(hipMemset(P->d_stage, 0, sizeof(double) * (size_t)std::max(pl.n_halo, 2)));
                HIP_TRY(hipMalloc(&P->d_ready, sizeof(unsigned) * kExtReadyStride * (1 + kExtReadyLines)));
                HIP_TRY(hipMemset(P->d_ready, 0, sizeof(unsigned) * kExtReadyStride * (1 + kExtReadyLines)));
                // inbound workgroups: four 16-byte loads of the (uncached) window per thread.  Measured at 38 648 ghosts (sim_rank.py 8 1 fe):
                // 4 workgroups 20.5 us per step, 8 17.6, 12 16.6, 19 15.3, 38 15.7, 76 16.9, 152 19.8 — every one of them polls the flags and
                // invalidates its caches once.  (MI355_PUSH_EXT_WGS: A/B)
                int xw = (pl.n_halo + 8 * kWG - 1) / (8 * kWG);
                if (const char* e = getenv("MI355_PUSH_EXT_WGS")) xw = atoi(e);
                P->ext_wgs = std::max(1, std::min(256, xw));
                bcsr4_drop_sliced(A->blocked); // (the step's kernel reads the row-major blocks)
                P->fused = P->fused_ext = true;
                P->ghost_readers = true;
            } else {
                mi_csr_destroy(P->piece_all);
                P->piece_all = nullptr;
            }
        }
    }
    // (MI355_PUSH_FUSED_KERNEL=csr_ext: none of the two forms below — the staged scalar step further down instead: tests, A/B)
    const char* fk0 = getenv("MI355_PUSH_FUSED_KERNEL");
    if (!P->fused && !(fe && !strcmp(fe, "0")) && !(fk0 && !strcmp(fk0, "csr_ext")) && pl.n_local > 0) {
        P->plan.build_combined();
        const LocalPiece& L = P->plan.all;
        rc = csr_create_impl(pl.n_local, pl.n_local + pl.n_halo, L.ptrow.data(), L.indcol.data(), L.coef.data(), nullptr, &P->piece_all,
                             P->plan.n_left, P->plan.n_left + pl.n_local); // ghosts: columns outside [n_left, n_left + n_local)
        if (rc) return rc;
        mi_csr_t A = P->piece_all;
        if (P->kernel != MI_KERNEL_AUTO && P->kernel != MI_KERNEL_RING) A->kernel = P->kernel;
        // Which kernel carries the one-launch step: the sliced stream (spmv_sstream_fused, round 5) or the ring kernel's FUSED form.  One of
        // the two is taken even if a kernel WITHOUT a fused form measured a hair faster on this rank: the one-launch step saves three
        // launches, and it only happens if EVERY rank has it (mi_part_push_unfuse) — a rank whose create-time measurement tipped the other
        // way by noise would cost all of them the fused step.  MI355_PUSH_FUSED_KERNEL=ring|sstream forces (A/B).
        const char* fk = getenv("MI355_PUSH_FUSED_KERNEL");
        const bool ss_ok = A->ss.dev.val && A->ss.fusable && !A->ss.h_wg_halo.empty() && !A->blocked && !(fk && !strcmp(fk, "ring"));
        const bool ring_ok = A->ring.d_plan && A->ring.d_run_halo && A->ring.ok_fraction >= 0.90 && !A->blocked && !(fk && !strcmp(fk, "sstream"));
        // (both are one-launch forms and may be mixed across ranks, so between the two this rank's own create-time measurement decides:
        // at N = 2 — 37 M nonzeros per rank, beyond the Infinity Cache — the ring form measured 66 us against 70 on one box, at N = 8 the
        // sliced stream 20.3 against 21.9)
        if (P->kernel == MI_KERNEL_AUTO && ss_ok && !(ring_ok && A->auto_kernel == MI_KERNEL_RING)) A->kernel = MI_KERNEL_SSTREAM;
        else if ((P->kernel == MI_KERNEL_AUTO || P->kernel == MI_KERNEL_RING) && ring_ok) A->kernel = MI_KERNEL_RING;
        const bool by_ss = resolve_kernel(A) == MI_KERNEL_SSTREAM && ss_ok;
        P->fused = by_ss || (resolve_kernel(A) == MI_KERNEL_RING && A->ring.d_run_halo);
        if (P->fused) { // push duty goes to the ghost-touching runs / workgroups (short by construction): link l to the (l mod k)-th of them
            const std::vector<int>& marks = by_ss ? A->ss.h_wg_halo : A->ring.h_run_halo;
            P->d_run_halo = by_ss ? A->ss.dev.wg_halo : A->ring.d_run_halo;
            std::vector<int> link(marks.size(), -1);
            // (alternately from the two ends of the row range: the ghost readers of a band are the first and the last few runs, and the
            // two neighbours' pushes should not both fall to the runs in front)
            std::vector<int> ghost_runs;
            for (size_t g = 0; g < marks.size(); g++)
                if (marks[g]) ghost_runs.push_back((int)g);
            int k = 0;
            for (size_t i = 0, lo_i = 0, hi_i = ghost_runs.size(); lo_i < hi_i && k < P->n_links; i++)
                link[ghost_runs[(i & 1) ? --hi_i : lo_i++]] = k++;
            P->npush_runs = k; // 0: no ghost runs in the plan -> dedicated push workgroups in front of the grid
            P->ghost_readers = false;
            for (int v : marks) P->ghost_readers = P->ghost_readers || v != 0;
            HIP_TRY(hipMalloc(&P->d_run_link, sizeof(int) * std::max<size_t>(link.size(), 1)));
            HIP_TRY(hipMemcpy(P->d_run_link, link.data(), sizeof(int) * link.size(), hipMemcpyHostToDevice));
            if (by_ss) { // the sliced stream reads its link out of the workgroup's record (one scalar load for the whole prologue)
                for (size_t g = 0; g < link.size(); g++) A->ss.h_wg[g].link = link[g];
                HIP_TRY(hipMemcpy(A->ss.dev.wg, A->ss.h_wg.data(), sizeof(SsWg) * A->ss.h_wg.size(), hipMemcpyHostToDevice));
            }
        }
        if (!P->fused) {
            mi_csr_destroy(P->piece_all);
            P->piece_all = nullptr;
        }
    }
    // ... and for SCALAR rows that got no one-launch form above (no 4x4 structure; a halo too wide for the sliced stream's and the ring's fused
    // forms — a 3-D mesh operator over ranks: a plane of ghosts each side): the staged step on the stream kernel's row blocks
    // (spmv_csr_fused_ext, spmv_bcsr4_ext.hpp).  MI355_PUSH_FUSED_CSR_EXT=0 keeps the four launches.
    const char* fc = getenv("MI355_PUSH_FUSED_CSR_EXT");
    if (!P->fused && !(fe && !strcmp(fe, "0")) && !(fx && !strcmp(fx, "0")) && !(fc && !strcmp(fc, "0")) && P->kernel == MI_KERNEL_AUTO && pl.n_local > 0 &&
        P->n_links > 0) {
        P->plan.build_all_ext();
        const LocalPiece& L = P->plan.all_ext;
        rc = csr_create_impl(pl.n_local, pl.n_local + pl.n_halo, L.ptrow.data(), L.indcol.data(), L.coef.data(), nullptr, &P->piece_all);
        if (rc) return rc;
        mi_csr_t A = P->piece_all;
        BlockTable* T = nullptr;
        if ((rc = get_table(A, 1024, &T))) return rc;
        std::vector<int2> blk((size_t)T->nblk + 1);
        HIP_TRY(hipMemcpy(blk.data(), T->d_blk, sizeof(int2) * blk.size(), hipMemcpyDeviceToHost));
        std::vector<unsigned> order;
        order.reserve((size_t)T->nblk);
        for (int pass = 0; pass < 2; pass++) { // the blocks that name a ghost LAST: dispatched behind the others, and a launch of their own in the two-launch form
            for (int b = 0; b < T->nblk; b++) {
                bool halo = false;
                for (int k = blk[b].y; k < blk[b + 1].y && !halo; k++) halo = L.indcol[k] >= pl.n_local;
                if (halo == (pass == 1)) order.push_back((unsigned)b | (halo ? 0x80000000u : 0u));
            }
            if (pass == 0) P->n_ext_plain = (int)order.size();
        }
        P->n_ext_units = (int)order.size();
        HIP_TRY(hipMalloc(&P->d_ext_order, sizeof(unsigned) * std::max<size_t>(order.size(), 1)));
        HIP_TRY(hipMemcpy(P->d_ext_order, order.data(), sizeof(unsigned) * order.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&P->d_stage, sizeof(double) * (size_t)std::max(pl.n_halo, 2)));
        HIP_TRY(hipMemset(P->d_stage, 0, sizeof(double) * (size_t)std::max(pl.n_halo, 2)));
        HIP_TRY(hipMalloc(&P->d_ready, sizeof(unsigned) * kExtReadyStride * (1 + kExtReadyLines)));
        HIP_TRY(hipMemset(P->d_ready, 0, sizeof(unsigned) * kExtReadyStride * (1 + kExtReadyLines)));
        int xw = (pl.n_halo + 8 * kWG - 1) / (8 * kWG);
        if (const char* e = getenv("MI355_PUSH_EXT_WGS")) xw = atoi(e);
        P->ext_wgs = std::max(1, std::min(256, xw));
        P->ext_split = P->peer_on_my_device;
        if (const char* e = getenv("MI355_PUSH_EXT_SPLIT")) P->ext_split = atoi(e) != 0;
        P->fused = P->fused_ext = P->ext_csr = true;
        P->ghost_readers = true;
    }
    return MI_OK;
}

// give the peer-push exchange up again (a failed collective self-check): windows and mappings released, a give-up counted
// during the check forgotten, so that the handle can go on with another exchange
extern "C" int mi_part_push_disable(mi_part_t P)
{
    CHECK_ARG(P, "null handle");
    HIP_TRY(hipDeviceSynchronize());
    for (void* m : P->ipc_opened) (void)hipIpcCloseMemHandle(m);
    P->ipc_opened.clear();
    if (P->win_registered) {
        std::lock_guard<std::mutex> lock(g_mu);
        g_win_registry.erase(P->win_key);
        P->win_registered = false;
    }
    dfree(P->win);
    dfree(P->d_links);
    dfree(P->d_push_work);
    dfree(P->d_link_chunks);
    dfree(P->d_tickets);
    P->d_push_work = nullptr;
    P->d_link_chunks = nullptr;
    P->d_tickets = nullptr;
    dfree(P->d_nb);
    dfree(P->d_run_link);
    dfree(P->d_stage);
    dfree(P->d_ready);
    dfree(P->d_ext_units);
    dfree(P->d_ext_order);
    P->d_stage = nullptr;
    P->d_ready = nullptr;
    P->d_ext_units = nullptr;
    P->d_ext_order = nullptr;
    P->ext_csr = false;
    P->d_run_link = nullptr;
    P->fused_ext = false;
    mi_csr_destroy(P->piece_all);
    P->win = nullptr;
    P->d_links = nullptr;
    P->d_nb = nullptr;
    P->piece_all = nullptr;
    P->fused = P->push_ready = false;
    P->n_links = P->n_nb = 0;
    if (P->h_timeouts) *P->h_timeouts = 0;
    return MI_OK;
}

// All ranks must drive the step the same way: whether the one-launch form is available is decided per rank (it needs the ring
// kernel to be the measured choice for the rank's combined piece), so the caller makes the decision collective and the ranks
// that could have fused step down to the four-launch form when a neighbour cannot.  (Ranks mixing the two forms passed the
// bitwise checks but, four processes sharing one card, a fused rank's in-kernel wait gave up in one run of four.)
extern "C" int mi_part_push_unfuse(mi_part_t P)
{
    CHECK_ARG(P, "null handle");
    if (!P->fused) return MI_OK;
    HIP_TRY(hipDeviceSynchronize());
    P->fused = P->fused_ext = false;
    dfree(P->d_run_link);
    dfree(P->d_stage);
    dfree(P->d_ready);
    dfree(P->d_ext_units);
    dfree(P->d_ext_order);
    P->d_stage = nullptr;
    P->d_ready = nullptr;
    P->d_ext_units = nullptr;
    P->d_ext_order = nullptr;
    P->ext_csr = false;
    P->d_run_link = nullptr;
    mi_csr_destroy(P->piece_all);
    P->piece_all = nullptr;
    return MI_OK;
}

// the kernel a piece's products launch: which = 0 interior rows, 1 boundary rows, 2 the combined piece of the one-launch push step ("" if none)
extern "C" const char* mi_part_kernel_name(mi_part_t P, int which)
{
    if (!P || which < 0 || which > 2) return "";
    if (which < 2) return P->piece[which] ? mi_csr_kernel_name(P->piece[which]) : "";
    if (!P->fused || !P->piece_all) return "";
    if (P->fused_ext && P->ext_csr) return P->ext_split ? "spmv_csr_fused_ext x2 (ranks share a device: two launches)" : "spmv_csr_fused_ext";
    if (P->fused_ext) return P->ext_split ? "spmv_bcsr4_fused_ext x2 (ranks share a device: two launches)" : "spmv_bcsr4_fused_ext";
    static thread_local char nm[160];
    const char* base = mi_csr_kernel_name(P->piece_all);
    if (resolve_kernel(P->piece_all) == MI_KERNEL_SSTREAM) snprintf(nm, sizeof nm, "spmv_sstream_fused<%d, %s>", P->piece_all->ss.deep ? 12 : 8, P->piece_all->ss.nt ? "true" : "false");
    else snprintf(nm, sizeof nm, "%s [FUSED]", base);
    return nm;
}

extern "C" int mi_part_push_info(mi_part_t P, int* ready, int* fused, int* neighbours)
{
    CHECK_ARG(P, "null handle");
    if (ready) *ready = P->push_ready ? 1 : 0;
    if (fused) *fused = P->fused ? 1 : 0;
    if (neighbours) *neighbours = P->n_nb;
    return MI_OK;
}
// ONE launch: the exchange workgroups in front, the blocked product over all rows behind them (spmv_bcsr4_ext.hpp)
int part_ext_launch(mi_part_s* P, const double* d_x_ext, double* d_y_local, unsigned step, unsigned spin_max, hipStream_t s, unsigned long long* trace, int* grid_out)
{
    const PartPlan& pl = P->plan;
    if (!P->ext_csr && (((uintptr_t)d_x_ext) & 15) != 0) return fail(MI_ERR_ARG, "the fused blocked step needs a 16-byte aligned x");
    ExtComm C;
    C.links = P->d_links;
    C.work = P->d_push_work;
    C.link_chunks = P->d_link_chunks;
    C.tickets = P->d_tickets;
    C.send_idx = P->d_send_idx;
    C.flags = P->win_flags;
    C.nb = P->d_nb;
    C.halo = P->win_data + (size_t)(step & 1u) * (size_t)pl.n_halo;
    C.stage = P->d_stage;
    C.ready = P->d_ready;
    C.timeouts = P->d_timeouts;
    C.n_work = P->n_push_work;
    C.n_nb = P->n_nb;
    C.n_local = pl.n_local;
    C.n_halo = pl.n_halo;
    C.xwgs = P->n_push_work + P->ext_wgs;
    C.step = step;
    C.spin_max = spin_max;
    C.trace = trace;
    C.nowait = 0;
    if (P->ext_debug & 2) C.links = nullptr; // (devtools: the push workgroups leave at once)
    if (P->ext_debug & 4) C.n_halo = 0;
    if (P->ext_debug & 8) C.n_nb = 0;
    const bool split = P->ext_split && !trace && P->n_ext_units > P->n_ext_plain;
    const dim3 grid((split ? P->n_ext_plain : P->n_ext_units) + C.xwgs);
    if (grid_out) *grid_out = (int)grid.x;
    if (P->ext_csr) { // scalar rows: the stream kernel's row blocks of piece_all
        mi_csr_t A = P->piece_all;
        BlockTable* T = nullptr;
        int rc = get_table(A, 1024, &T);
        if (rc) return rc;
        CsrView V{A->n, A->ncols, A->d_ptrow, A->d_indcol, A->d_coef, nullptr, T->d_blk, nullptr, T->nblk};
        if (trace) hipLaunchKernelGGL((spmv_csr_fused_ext<1024, false, true>), grid, dim3(kWG), 0, s, V, d_x_ext, d_y_local, C, P->d_ext_order);
        else if (A->stream_nt) hipLaunchKernelGGL((spmv_csr_fused_ext<1024, true>), grid, dim3(kWG), 0, s, V, d_x_ext, d_y_local, C, P->d_ext_order);
        else hipLaunchKernelGGL((spmv_csr_fused_ext<1024, false>), grid, dim3(kWG), 0, s, V, d_x_ext, d_y_local, C, P->d_ext_order);
        if (split) { // the blocks that name ghosts, behind the exchange by stream order
            ExtComm C2 = C;
            C2.xwgs = C2.n_work = 0;
            C2.nowait = 1;
            const dim3 g2(P->n_ext_units - P->n_ext_plain);
            if (A->stream_nt) hipLaunchKernelGGL((spmv_csr_fused_ext<1024, true>), g2, dim3(kWG), 0, s, V, d_x_ext, d_y_local, C2, P->d_ext_order + P->n_ext_plain);
            else hipLaunchKernelGGL((spmv_csr_fused_ext<1024, false>), g2, dim3(kWG), 0, s, V, d_x_ext, d_y_local, C2, P->d_ext_order + P->n_ext_plain);
        }
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    mi_bcsr4_t B = P->piece_all->blocked;
    Bcsr4View V{B->nbrows, B->nbcols, B->d_ptrow, B->d_indcol, B->d_coef, nullptr};
    if (trace) {
        hipLaunchKernelGGL((spmv_bcsr4_fused_ext<kBcsrDepth, 4, kWG, true>), grid, dim3(kWG), 0, s, V, d_x_ext, d_y_local, C, P->d_ext_units);
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    static const int depth = getenv("MI355_PUSH_EXT_DEPTH") ? atoi(getenv("MI355_PUSH_EXT_DEPTH")) : 24; // (A/B: blocks in flight, 4-lane rows / 16-lane rows)
#define EXT_LAUNCH(P_, U_) hipLaunchKernelGGL((spmv_bcsr4_fused_ext<P_, U_, kWG>), grid, dim3(kWG), 0, s, V, d_x_ext, d_y_local, C, P->d_ext_units)
    switch (depth) {
    case 22: EXT_LAUNCH(2, 2); break;
    case 23: EXT_LAUNCH(2, 3); break;
    case 27: EXT_LAUNCH(2, 7); break;
    default: EXT_LAUNCH(2, 4); break;
    }
#undef EXT_LAUNCH
    if (split) { // the units that name ghosts, behind the exchange by stream order
        ExtComm C2 = C;
        C2.xwgs = C2.n_work = 0;
        C2.nowait = 1;
        hipLaunchKernelGGL((spmv_bcsr4_fused_ext<kBcsrDepth, 4, kWG>), dim3(P->n_ext_units - P->n_ext_plain), dim3(kWG), 0, s, V, d_x_ext, d_y_local, C2,
                           P->d_ext_units + P->n_ext_plain);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_part_spmv_push_dev(mi_part_t P, double* d_x_ext, double* d_y_local, mi_stream_t s_)
{
    CHECK_ARG(P, "null handle");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    hipStream_t s = (hipStream_t)s_;
    const PartPlan& pl = P->plan;
    int rc;
    if (pl.nranks == 1) {
        if ((rc = mi_spmv_dev(P->piece[0], d_x_ext, d_y_local, s))) return rc;
        return mi_spmv_dev(P->piece[1], d_x_ext, d_y_local, s);
    }
    if (!P->push_ready) return fail(MI_ERR_STATE, "mi_part_push_connect was not called");
    if ((rc = part_handoff_status(P))) return rc;
    if (stream_is_capturing(s))
        return fail(MI_ERR_UNSUPPORTED, "the push step cannot be captured into a HIP graph: its step number is a kernel argument (a replay would "
                                        "present an old step and every wait would pass at once); capture mi_part_spmv_dev (RCCL, events) instead");
    const unsigned step = ++P->push_step;
    static const unsigned spin_max = 1u << (getenv("MI355_PUSH_SPIN_LOG2") ? std::max(8, std::min(30, atoi(getenv("MI355_PUSH_SPIN_LOG2")))) : kPushSpinLog2Default);
    if (P->fused_ext) return part_ext_launch(P, d_x_ext, d_y_local, step, spin_max, s, nullptr, nullptr);
    if (P->fused) { // ONE launch: push workgroups first, then the ring kernel over all rows, ghost readers waiting in-kernel
        RingComm C;
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
        C.push_wgs = (P->npush_runs == 0 && P->n_links > 0) ? kNXCD : 0; // fallback only; a multiple of the XCD count keeps the run-to-XCD mapping
        C.step = step;
        C.spin_max = spin_max;
        C.gate_push = P->ghost_readers ? 0 : 1; // nobody in this launch waits for the neighbours: the pushers do (push_exchange.hpp)
        if ((rc = launch_spmv(P->piece_all, d_x_ext, d_y_local, s, true, &C))) return rc;
        return MI_OK;
    }
    // one stream, four launches: my entries to the neighbours' windows, interior rows (need owned x only), wait for the
    // neighbours' entries and move them behind x_local, boundary rows
    if (P->n_links)
        hipLaunchKernelGGL(halo_push_kernel, dim3(P->n_push_work), dim3(256), 0, s, P->d_links, P->d_push_work, P->d_link_chunks, P->d_tickets,
                           P->d_send_idx, d_x_ext, step);
    if ((rc = mi_spmv_dev(P->piece[0], d_x_ext, d_y_local, s))) return rc;
    if (P->n_nb) {
        int grid = (pl.n_halo + 511) / 512; // one 16-byte load per thread: the window is uncached, so spread it wide
        grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
        hipLaunchKernelGGL(halo_wait_copy_kernel, dim3(grid), dim3(256), 0, s, P->win_flags, P->d_nb, P->n_nb, step,
                           P->win_data + (size_t)(step & 1u) * (size_t)(pl.n_halo > 0 ? pl.n_halo : 1), d_x_ext + pl.n_local, pl.n_halo,
                           P->d_timeouts, spin_max);
    }
    if ((rc = mi_spmv_dev(P->piece[1], d_x_ext, d_y_local, s))) return rc;
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// New coefficients for an unchanged pattern (see mi_csr_update_values): coef = this rank's values in the order of the arrays
// given to mi_part_create.  The interior / boundary pieces take theirs through the nonzero positions recorded at plan time;
// the fused step's piece holds all rows in the caller's order, so it takes the array as it is.
extern "C" int mi_part_update_values(mi_part_t P, const double* coef)
{
    CHECK_ARG(P, "null handle");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    for (int w = 0; w < 2; w++) {
        LocalPiece& L = P->plan.piece[w];
        if (L.src.empty()) continue;
        CHECK_ARG(coef, "null coef");
        for (size_t k = 0; k < L.src.size(); k++) L.coef[k] = coef[L.src[k]];
        int rc = mi_csr_update_values(P->piece[w], L.coef.data());
        if (rc) return rc;
    }
    if (P->piece_all) {
        LocalPiece& L = P->fused_ext ? P->plan.all_ext : P->plan.all;
        CHECK_ARG(coef || L.coef.empty(), "null coef");
        for (size_t k = 0; k < L.coef.size(); k++) L.coef[k] = coef[k];
        int rc = mi_csr_update_values(P->piece_all, coef);
        if (rc) return rc;
    }
    return MI_OK;
}

extern "C" int mi_part_set_kernel(mi_part_t P, int kernel_id)
{
    CHECK_ARG(P, "null handle");
    CHECK_ARG((kernel_id >= MI_KERNEL_AUTO && kernel_id <= MI_KERNEL_ROWPAR) || kernel_id == MI_KERNEL_SSTREAM, "unknown kernel id (a partition takes AUTO, STREAM, RING, ROWPAR, SSTREAM)");
    P->kernel = kernel_id;
    for (int w = 0; w < 2; w++)
        if (P->piece[w]) { // (a piece that holds no sliced copy — too few rows, ragged — keeps its next-best kernel: resolve_kernel)
            if (kernel_id == MI_KERNEL_SSTREAM) (void)mi_csr_set_kernel(P->piece[w], kernel_id);
            P->piece[w]->kernel = kernel_id;
        }
    return MI_OK;
}

extern "C" int mi_part_pack_dev(mi_part_t P, const double* d_x_ext, double* d_sendbuf, mi_stream_t s)
{
    CHECK_ARG(P, "null handle");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    return mi_gather_dev((int)P->plan.send_idx.size(), P->d_send_idx, d_x_ext, d_sendbuf, s);
}

extern "C" int mi_part_spmv_interior_dev(mi_part_t P, const double* d_x_ext, double* d_y_local, mi_stream_t s)
{
    CHECK_ARG(P, "null handle");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    return mi_spmv_dev(P->piece[0], d_x_ext, d_y_local, s);
}

extern "C" int mi_part_spmv_boundary_dev(mi_part_t P, const double* d_x_ext, double* d_y_local, mi_stream_t s)
{
    CHECK_ARG(P, "null handle");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    return mi_spmv_dev(P->piece[1], d_x_ext, d_y_local, s);
}
