// capi_dist.hip: mi_dist_* — ONE process, N devices: the multi-GPU split behind a single handle (SURVEY.md §8(b) export list:
// `mi_dist_create(ndev, …)`; "Threading: one host thread drives all GPUs (or one per GPU internally), hidden behind the ABI").
// Part of libmi355spmv.so (capi_internal.hpp has the layout of the library).  gfx950 only; no CPU fallback.
//
// New design (the reference is a single-process, single-device C++ program: mpk/2SpMV.cpp:43-296, mpk/SpM2V.cpp:804-987, seam
// mpk/SpMV.h:52-66; it has no distributed code).  What a harness linked against the mpk/SpMV.h shim calls is
// `SpMV_CSR(y, x, A)` from ONE thread; for that call to reach N GPUs the split must live under the C-ABI:
//   * mi_dist_create cuts the rows into N nnz-balanced contiguous ranges (node-aligned for 4x4-block FE matrices), builds every
//     rank's PartPlan (partition.hpp) in this address space — no id exchange, the send lists are read off the peers' plans —
//     and finalises rank r's pieces on device r;
//   * one WORKER THREAD per rank (bound to its device for life) enqueues that rank's launches, so that a step's N launches
//     are issued in parallel (a single host thread would serialise ≈5 µs of launch cost per device: 40 µs at N = 8 for a 22 µs step);
//     the calling thread posts a job and waits for the enqueue, not for the GPUs (the *_dev entry points are asynchronous);
//   * three exchanges drive the step, the first that comes up AND passes a bit-for-bit self-check against the event exchange is used:
//       push   (ranks on DISTINCT devices) the peer-push step of push_exchange.hpp — one launch per step and rank in its fused form —
//              with the neighbours' receive windows reached through peer pointers (hipDeviceEnablePeerAccess) instead of HIP IPC;
//       rccl   mi_part_comm_init + mi_part_spmv_dev per worker thread (ncclCommInitRank over the threads; grouped send/recv);
//       event  always available, any rank→device map, also N ranks on ONE device (the one-GPU development lease):
//              halo entries copied straight into the peers' [owned | halo] vectors with hipMemcpyPeerAsync on the sender's stream,
//              ordered by HIP events — ev_push[r] (my entries are on their way) and ev_done[r] (I have finished reading my halo);
//              the spinning push form stays refused for ranks that share a device (push_exchange.hpp: shared hardware queues).
// Every row of y is the same CSR-ordered fma chain as on one GPU (PartPlan keeps each row's nonzeros in the caller's order), so
// mi_dist_spmv returns mi_spmv's bits — and SpMV_CSR_FMA's (mpk/SpMV.cpp:41-56) — for every N.
#include "capi_internal.hpp"
#include "rccl_loader.hpp"

#include <atomic>
#include <condition_variable>
#include <functional>
#include <thread>

#include <sched.h>

namespace {

enum { kExEvent = 0, kExPush = 1, kExRccl = 2 };
constexpr long long kDistAllGatherHalo = 16384; // ghosts of the widest rank from which the RCCL step all-gathers boundary slices
const char* const kExNames[3] = {"event", "push", "rccl"};

struct DistRank {
    int device = 0;
    mi_part_t part = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_push = nullptr, ev_done = nullptr;
    double* d_sendbuf = nullptr; // event exchange: packed entries of non-contiguous send lists
    double* d_scal = nullptr;    // BLAS-1: this rank's partial [0], the global value as its update kernel summed it [1]
    double* h_scal = nullptr;    // ... and where they land on the host (pinned): [0] partial, [1] global
    double* d_parts = nullptr;   // RCCL exchange: every rank's partial, all-gathered (nranks doubles)
    hipEvent_t ev_dot[2] = {nullptr, nullptr}; // my partial of an orthogonalize has landed in the shared table (two parities)
    int n_local = 0, n_halo = 0;
    long long row0 = 0, nnz0 = 0, nnz_local = 0;
    std::vector<int> nb;         // neighbours: ranks I send to or receive from
    std::vector<int> ptrow;      // this rank's row pointers, relative to its first nonzero (kept for nothing but create)
};

// N worker threads, one job at a time: run(fn) executes fn(rank) on every worker and returns the first failure.
// Workers spin briefly for the next job (back-to-back steps find them awake) and then sleep on a condition variable.
struct Pool {
    int n = 0;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::atomic<unsigned> gen{0};
    std::atomic<int> remaining{0};
    std::function<int(int)> job;
    std::vector<int> rc;
    std::vector<std::string> err;
    std::atomic<bool> stop{false};
    // barrier among the workers inside a job (sense-reversing)
    std::atomic<int> bar_count{0};
    std::atomic<unsigned> bar_gen{0};

    void barrier()
    {
        if (n <= 1) return;
        const unsigned g = bar_gen.load(std::memory_order_acquire);
        if (bar_count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            bar_count.store(0, std::memory_order_relaxed);
            bar_gen.store(g + 1, std::memory_order_release);
        } else {
            unsigned spins = 0;
            while (bar_gen.load(std::memory_order_acquire) == g)
                if (++spins > 2000) sched_yield();
        }
    }

    void worker(int r, int device)
    {
        (void)hipSetDevice(device); // for the life of the thread
        unsigned seen = 0;
        for (;;) {
            unsigned spins = 0;
            while (gen.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_acquire)) {
                if (++spins < 20000) {
                    if ((spins & 63) == 0) sched_yield();
                    continue;
                }
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen || stop.load(std::memory_order_acquire); });
            }
            if (stop.load(std::memory_order_acquire)) return;
            seen = gen.load(std::memory_order_acquire);
            g_err.clear();
            const int c = job(r);
            rc[r] = c;
            err[r] = c ? g_err : std::string();
            if (remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu);
                cv_done.notify_all();
            }
        }
    }

    void start(const std::vector<int>& devices)
    {
        n = (int)devices.size();
        rc.assign((size_t)n, 0);
        err.assign((size_t)n, std::string());
        for (int r = 0; r < n; r++) th.emplace_back(&Pool::worker, this, r, devices[r]);
    }

    int run(std::function<int(int)> fn)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(fn);
            remaining.store(n, std::memory_order_release);
            gen.fetch_add(1, std::memory_order_acq_rel);
        }
        cv_job.notify_all();
        unsigned spins = 0;
        while (remaining.load(std::memory_order_acquire) != 0) {
            if (++spins < 20000) {
                if ((spins & 63) == 0) sched_yield();
                continue;
            }
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return remaining.load(std::memory_order_acquire) == 0; });
        }
        for (int r = 0; r < n; r++)
            if (rc[r]) return fail(rc[r], "rank " + std::to_string(r) + ": " + err[r]);
        return MI_OK;
    }

    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop.store(true, std::memory_order_release);
        }
        cv_job.notify_all();
        for (std::thread& t : th)
            if (t.joinable()) t.join();
        th.clear();
    }
};

} // namespace

struct mi_dist_vec_s {
    mi_dist_s* D = nullptr;
    std::vector<double*> ext; // per rank: [owned | halo] on the rank's device
};

struct mi_dist_s {
    int nranks = 0, n = 0;
    long long nnz = 0;
    std::vector<long long> rs; // [nranks + 1]
    std::vector<DistRank> R;
    int distinct_devices = 0;
    int exchange = kExEvent;
    bool fused = false;       // push: every rank runs the one-launch form
    bool allgather = false;   // rccl: the all-gather form of the exchange (wide halos)
    std::string note;         // how the exchange was chosen (what was tried, why it was dropped)
    Pool pool;
    std::mutex api_mu;        // one API call at a time per handle
    double* h_parts = nullptr;  // orthogonalize without RCCL: the ranks' partials, 2 parities x nranks doubles of pinned memory every device reads
    unsigned ortho_calls = 0;
    std::vector<mi_dist_vec_t> live; // every vector handed out and not yet destroyed (mi_dist_destroy releases their device memory)
    mi_dist_vec_t vx = nullptr, vy = nullptr, vz = nullptr; // scratch of the host-pointer entry points
    std::vector<mi_dist_vec_t> vpow;
};

namespace {

std::mutex g_dev_mu[64]; // create-time work (kernel timing at finalize) of ranks that share a device runs one after the other

// (skipped once rc is set: a failed rank does no further work, but still meets its barriers)
#define W_HIP(expr)                                                                                              \
    do {                                                                                                         \
        if (rc == MI_OK) {                                                                                       \
            hipError_t e_ = (expr);                                                                              \
            if (e_ != hipSuccess) rc = fail(e_ == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                                        \
    } while (0)

// One product step of rank r, enqueued on its stream: y_r = (A x)_r with x = [owned | halo] vectors of all ranks.
// Every rank's worker runs this for the same step; `ex` selects the exchange.  Workers call the pool barrier the same number of
// times whatever fails (a failed rank skips its work, not its barriers).
// rc_in: what an earlier step of the same job returned on this rank (a chain of steps stops working at its first failure)
int dist_step(mi_dist_s* D, int r, int ex, const mi_dist_vec_s* x, const mi_dist_vec_s* y, int rc_in = MI_OK, bool more_steps_in_job = false)
{
    DistRank& me = D->R[r];
    const PartPlan& pl = me.part->plan;
    double* x_ext = x->ext[r];
    double* y_loc = y->ext[r];
    if (D->nranks == 1 || ex != kExEvent) {
        if (rc_in) return rc_in;
        if (D->nranks == 1) return mi_part_spmv_dev(me.part, x_ext, y_loc, me.stream);
        if (ex == kExPush) return mi_part_spmv_push_dev(me.part, x_ext, y_loc, me.stream);
        return mi_part_spmv_dev(me.part, x_ext, y_loc, me.stream);
    }
    int rc = rc_in;
    // phase A: once every neighbour has finished reading its halo of the PREVIOUS step, my entries go out — straight into the
    // neighbours' vectors (their halo parts), on my stream
    for (int p : me.nb) W_HIP(hipStreamWaitEvent(me.stream, D->R[p].ev_done, 0));
    if (rc == MI_OK && !pl.sends_contiguous && !pl.send_idx.empty()) rc = mi_part_pack_dev(me.part, x_ext, me.d_sendbuf, me.stream);
    for (int p = 0; p < pl.nranks && rc == MI_OK; p++) {
        if (!pl.send_counts[p]) continue;
        const DistRank& peer = D->R[p];
        const double* src = pl.sends_contiguous ? x_ext + pl.send_lists[p][0] : me.d_sendbuf + pl.send_offsets[p];
        double* dst = x->ext[p] + peer.n_local + peer.part->plan.recv_offsets[r];
        const size_t bytes = sizeof(double) * (size_t)pl.send_counts[p];
        if (peer.device == me.device) W_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, me.stream));
        else W_HIP(hipMemcpyPeerAsync(dst, peer.device, src, me.device, bytes, me.stream));
    }
    W_HIP(hipEventRecord(me.ev_push, me.stream));
    D->pool.barrier(); // every rank's ev_push of THIS step is recorded before anybody waits for it
    // phase B: interior rows beside the neighbours' copies, boundary rows behind them
    if (rc == MI_OK) rc = mi_part_spmv_interior_dev(me.part, x_ext, y_loc, me.stream);
    for (int p : me.nb) W_HIP(hipStreamWaitEvent(me.stream, D->R[p].ev_push, 0));
    if (rc == MI_OK) rc = mi_part_spmv_boundary_dev(me.part, x_ext, y_loc, me.stream);
    W_HIP(hipEventRecord(me.ev_done, me.stream));
    // ... and every ev_done of this step before the next step's phase A: a barrier when that step follows in THIS job (a chain of
    // powers); the last step of a job needs none — the job's end is one (Pool::run returns when every worker has)
    if (more_steps_in_job) D->pool.barrier();
    return rc;
}

int vec_alloc(mi_dist_s* D, mi_dist_vec_t* out)
{
    mi_dist_vec_t v = new (std::nothrow) mi_dist_vec_s();
    if (!v) return fail(MI_ERR_ALLOC, "host allocation failed");
    v->D = D;
    v->ext.assign((size_t)D->nranks, nullptr);
    int rc = D->pool.run([&](int r) -> int {
        const DistRank& me = D->R[r];
        const size_t len = (size_t)me.n_local + (size_t)me.n_halo + 64; // (64 spare entries: 16-byte tails of the BLAS-1 kernels never straddle the end)
        HIP_TRY(hipMalloc(&v->ext[r], sizeof(double) * len));
        HIP_TRY(hipMemsetAsync(v->ext[r], 0, sizeof(double) * len, me.stream));
        // (create-time cost only) the zero fill is complete before anybody can use the vector: in the event exchange the PEERS write this
        // vector's halo part from their streams, ordered behind this rank's ev_done of the previous step only — a fill still queued here
        // could land on top of a neighbour's entries
        HIP_TRY(hipStreamSynchronize(me.stream));
        return MI_OK;
    });
    if (rc) {
        const std::string keep = g_err;
        D->pool.run([&](int r) -> int { dfree(v->ext[r]); return MI_OK; });
        delete v;
        return fail(rc, keep);
    }
    D->live.push_back(v);
    *out = v;
    return MI_OK;
}

void vec_free(mi_dist_s* D, mi_dist_vec_t v)
{
    if (!v) return;
    D->live.erase(std::remove(D->live.begin(), D->live.end(), v), D->live.end());
    D->pool.run([&](int r) -> int {
        (void)hipStreamSynchronize(D->R[r].stream);
        dfree(v->ext[r]);
        return MI_OK;
    });
    delete v;
}

int need_scratch(mi_dist_s* D, int npow)
{
    int rc;
    if (!D->vx && (rc = vec_alloc(D, &D->vx))) return rc;
    if (!D->vy && (rc = vec_alloc(D, &D->vy))) return rc;
    while ((int)D->vpow.size() < npow) {
        mi_dist_vec_t v = nullptr;
        if ((rc = vec_alloc(D, &v))) return rc;
        D->vpow.push_back(v);
    }
    return MI_OK;
}

// scatter / gather of a full-length host vector (owned parts only)
int vec_set(mi_dist_s* D, mi_dist_vec_t v, const double* host)
{
    return D->pool.run([&](int r) -> int {
        const DistRank& me = D->R[r];
        if (me.n_local) HIP_TRY(hipMemcpyAsync(v->ext[r], host + me.row0, sizeof(double) * (size_t)me.n_local, hipMemcpyHostToDevice, me.stream));
        return MI_OK;
    });
}

int vec_get(mi_dist_s* D, mi_dist_vec_t v, double* host)
{
    return D->pool.run([&](int r) -> int {
        const DistRank& me = D->R[r];
        if (me.n_local) HIP_TRY(hipMemcpyAsync(host + me.row0, v->ext[r], sizeof(double) * (size_t)me.n_local, hipMemcpyDeviceToHost, me.stream));
        HIP_TRY(hipStreamSynchronize(me.stream));
        return mi_part_status(me.part);
    });
}

int dist_sync(mi_dist_s* D)
{
    return D->pool.run([&](int r) -> int {
        HIP_TRY(hipStreamSynchronize(D->R[r].stream));
        return mi_part_status(D->R[r].part);
    });
}

// global dot: every rank's fixed-tree partial (mi_dot_dev), summed on the host in rank order — deterministic for a given N
int dist_dot(mi_dist_s* D, const mi_dist_vec_s* a, const mi_dist_vec_s* b, double* out)
{
    int rc = D->pool.run([&](int r) -> int {
        DistRank& me = D->R[r];
        *me.h_scal = 0.0;
        if (me.n_local) {
            int c = mi_dot_dev(me.n_local, a->ext[r], b->ext[r], me.d_scal, me.stream);
            if (c) return c;
            HIP_TRY(hipMemcpyAsync(me.h_scal, me.d_scal, sizeof(double), hipMemcpyDeviceToHost, me.stream));
        }
        HIP_TRY(hipStreamSynchronize(me.stream));
        return mi_part_status(me.part);
    });
    if (rc) return rc;
    double s = 0.0;
    for (int r = 0; r < D->nranks; r++) s += *D->R[r].h_scal;
    *out = s;
    return MI_OK;
}

// x3 = x1 - alpha (b . x1) b over all ranks (orthogonalize, mpk/SpMVmulti.cpp:146-151) WITHOUT the host in the middle (round 5; until
// round 4: partial dots to the host, a stream synchronise per rank, the sum on the host, then the updates).  Every rank's fixed-tree
// partial goes into a table every rank reads — one ncclAllGather of a double where the ranks hold a communicator (the RCCL exchange;
// SURVEY.md §8(e) "local two-stage reduce + a collective of one double"), else a slot of pinned host memory behind an event — and each
// rank's update kernel adds the table up in RANK ORDER itself (the host's order: the same beta bit for bit as before) and applies it to
// its slice.  Only a caller that wants beta on the host waits, at the end, for rank 0's copy of it.
int dist_ortho(mi_dist_s* D, const mi_dist_vec_s* b, const mi_dist_vec_s* x1, mi_dist_vec_s* x3, double alpha, double* beta_out)
{
    const int N = D->nranks;
    const unsigned par = D->ortho_calls++ & 1u;
    const bool by_rccl = N > 1 && D->exchange == kExRccl && rccl_state().ok && rccl_state().AllGather;
    int rc = D->pool.run([&](int r) -> int {
        DistRank& me = D->R[r];
        int rc = MI_OK; // (a failed rank skips its work, not its barrier)
        const double* parts = nullptr;
        if (me.n_local) rc = mi_dot_dev(me.n_local, b->ext[r], x1->ext[r], me.d_scal, me.stream);
        else W_HIP(hipMemsetAsync(me.d_scal, 0, sizeof(double), me.stream));
        if (N == 1) parts = me.d_scal;
        else if (by_rccl) {
            if (rc == MI_OK && rccl_state().AllGather(me.d_scal, me.d_parts, 1, kNcclDouble, me.part->comm, me.stream) != 0) rc = fail(MI_ERR_HIP, "ncclAllGather of the partial dots failed");
            parts = me.d_parts;
        } else {
            double* slot = D->h_parts + (size_t)par * N;
            W_HIP(hipMemcpyAsync(slot + r, me.d_scal, sizeof(double), hipMemcpyDeviceToHost, me.stream));
            W_HIP(hipEventRecord(me.ev_dot[par], me.stream));
            D->pool.barrier(); // every rank's event of THIS call is recorded before anybody waits for it
            for (int p = 0; p < N; p++)
                if (p != r) W_HIP(hipStreamWaitEvent(me.stream, D->R[p].ev_dot[par], 0));
            parts = slot;
            // (two parities: a rank overwrites its slot of parity `par` two calls later, after it has waited for every peer's event of the
            // call in between — which each peer recorded behind its own update kernel of this call)
        }
        if (rc == MI_OK) rc = ortho_update_from_parts(me.n_local, N, parts, alpha, b->ext[r], x1->ext[r], x3->ext[r], me.d_scal + 1, me.stream);
        if (beta_out && r == 0) {
            W_HIP(hipMemcpyAsync(me.h_scal + 1, me.d_scal + 1, sizeof(double), hipMemcpyDeviceToHost, me.stream));
            W_HIP(hipStreamSynchronize(me.stream));
            if (rc == MI_OK) rc = mi_part_status(me.part);
        }
        return rc;
    });
    if (rc) return rc;
    if (beta_out) *beta_out = D->R[0].h_scal[1];
    return MI_OK;
}

void dist_release(mi_dist_s* D)
{
    if (!D->pool.th.empty()) {
        for (mi_dist_vec_t v : {D->vx, D->vy, D->vz}) vec_free(D, v);
        for (mi_dist_vec_t v : D->vpow) vec_free(D, v);
        // vectors the caller still holds: their device memory goes with the handle; the small host object stays for the caller's
        // mi_dist_vec_destroy, which finds D == nullptr and only deletes it (it used to dereference the freed handle: ADVICE r4)
        const std::vector<mi_dist_vec_t> orphans = D->live;
        for (mi_dist_vec_t v : orphans) {
            D->pool.run([&](int r) -> int {
                (void)hipStreamSynchronize(D->R[r].stream);
                dfree(v->ext[r]);
                v->ext[r] = nullptr;
                return MI_OK;
            });
            v->D = nullptr;
        }
        D->live.clear();
        D->pool.run([&](int r) -> int {
            DistRank& me = D->R[r];
            if (me.stream) (void)hipStreamSynchronize(me.stream);
            (void)mi_part_destroy(me.part);
            me.part = nullptr;
            dfree(me.d_sendbuf);
            dfree(me.d_scal);
            dfree(me.d_parts);
            if (me.h_scal) (void)hipHostFree(me.h_scal);
            for (hipEvent_t& e : me.ev_dot)
                if (e) (void)hipEventDestroy(e);
            if (me.ev_push) (void)hipEventDestroy(me.ev_push);
            if (me.ev_done) (void)hipEventDestroy(me.ev_done);
            if (me.stream) (void)hipStreamDestroy(me.stream);
            return MI_OK;
        });
        D->pool.shutdown();
    }
    if (D->h_parts) (void)hipHostFree(D->h_parts);
    delete D;
}

// bring the peer-push exchange up between ranks on distinct devices; MI_OK + *ok = false when it is not available
int try_push(mi_dist_s* D, bool* ok, std::string* why)
{
    *ok = false;
    const int N = D->nranks;
    for (int r = 0; r < N; r++)
        for (int p : D->R[r].nb)
            if (D->R[p].device == D->R[r].device) {
                *why = "neighbouring ranks share a device (the spinning push form is refused there)";
                return MI_OK;
            }
    std::vector<long long> layouts((size_t)N * (2 * N + 1), 0);
    int rc = D->pool.run([&](int r) -> int {
        DistRank& me = D->R[r];
        for (int p : me.nb) { // my kernels store into p's window: peer access from my device to p's
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, me.device, D->R[p].device));
            if (!can) return fail(MI_ERR_UNSUPPORTED, "no peer access between devices " + std::to_string(me.device) + " and " + std::to_string(D->R[p].device));
            hipError_t e = hipDeviceEnablePeerAccess(D->R[p].device, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e != hipSuccess) return fail(MI_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        }
        int c = part_push_window(me.part);
        if (c) return c;
        part_push_layout(me.part, layouts.data() + (size_t)r * (2 * N + 1));
        return MI_OK;
    });
    if (rc == MI_OK) {
        std::vector<void*> bases((size_t)N, nullptr);
        for (int r = 0; r < N; r++) bases[r] = D->R[r].part->win;
        rc = D->pool.run([&](int r) -> int { return part_push_connect_bases(D->R[r].part, bases.data(), layouts.data()); });
    }
    if (rc == MI_OK) { // one form for everybody (mixed forms are legal; equal forms keep the step times balanced)
        bool all_fused = true;
        for (int r = 0; r < N; r++) all_fused = all_fused && D->R[r].part->fused;
        if (!all_fused) rc = D->pool.run([&](int r) -> int { return mi_part_push_unfuse(D->R[r].part); });
        D->fused = all_fused;
    }
    if (rc) {
        *why = g_err;
        D->pool.run([&](int r) -> int { return mi_part_push_disable(D->R[r].part); });
        D->fused = false;
        return MI_OK;
    }
    *ok = true;
    return MI_OK;
}

int try_rccl(mi_dist_s* D, bool* ok, std::string* why)
{
    *ok = false;
    if (mi_comm_available() != MI_OK) {
        *why = g_err;
        return MI_OK;
    }
    char id[MI_COMM_ID_BYTES];
    if (mi_comm_unique_id(id) != MI_OK) {
        *why = g_err;
        return MI_OK;
    }
    // ncclCommInitRank is collective over the worker threads
    int rc = D->pool.run([&](int r) -> int { return mi_part_comm_init(D->R[r].part, id); });
    if (rc) {
        *why = g_err;
        return MI_OK;
    }
    // wide halos (FE slab partitions) take the all-gather form of the step: one ncclAllGather of every rank's boundary slice instead
    // of a send / recv pair per neighbour.  One address space: every rank's union list is read off its plan.
    // MI355_PART_EXCHANGE=allgather | sendrecv forces the form.
    const char* fe = getenv("MI355_PART_EXCHANGE");
    long long hmax = 0;
    for (const DistRank& r : D->R) hmax = std::max<long long>(hmax, r.part->plan.n_halo);
    const bool want_ag = fe ? !strcmp(fe, "allgather") : hmax >= kDistAllGatherHalo;
    if (want_ag) {
        const int N = D->nranks;
        std::vector<int> counts((size_t)N, 0);
        std::vector<long long> ids;
        for (int r = 0; r < N; r++) {
            const int* loc = nullptr;
            if (mi_part_send_union(D->R[r].part, &counts[r], &loc) != MI_OK) {
                *why = g_err;
                return MI_OK;
            }
            for (int i = 0; i < counts[r]; i++) ids.push_back(D->R[r].row0 + loc[i]);
        }
        rc = D->pool.run([&](int r) -> int {
            int c = mi_part_allgather_setup(D->R[r].part, counts.data(), ids.data());
            return c ? c : mi_part_set_allgather(D->R[r].part, 1);
        });
        if (rc) {
            *why = "all-gather form: " + g_err;
            return MI_OK;
        }
        D->allgather = true;
    }
    *ok = true;
    return MI_OK;
}

// one product through exchange `ex` and one through the event exchange on the same (counter-based) vector: equal bits on every
// rank, or `ex` is not used
int selfcheck(mi_dist_s* D, int ex, bool* same)
{
    *same = false;
    int rc = need_scratch(D, 0);
    if (rc) return rc;
    if (!D->vz && (rc = vec_alloc(D, &D->vz))) return rc;
    std::vector<double> hx((size_t)D->n), y0((size_t)D->n), y1((size_t)D->n);
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < D->n; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        hx[i] = (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
    if ((rc = vec_set(D, D->vx, hx.data()))) return rc;
    for (int rep = 0; rep < 3 && rc == MI_OK; rep++) // three steps back to back: both parities of a window, a step on top of a step
        rc = D->pool.run([&](int r) -> int { return dist_step(D, r, ex, D->vx, D->vy); });
    if (rc == MI_OK) rc = vec_get(D, D->vy, y1.data());
    if (rc) return rc;
    if ((rc = D->pool.run([&](int r) -> int { return dist_step(D, r, kExEvent, D->vx, D->vz); }))) return rc;
    if ((rc = vec_get(D, D->vz, y0.data()))) return rc;
    *same = memcmp(y0.data(), y1.data(), sizeof(double) * (size_t)D->n) == 0;
    return MI_OK;
}

} // namespace

// ---------------------------------------------------------------- create / destroy
extern "C" int mi_dist_create(int ndev, int n, const int* ptrow, const int* indcol, const double* coef, mi_dist_t* out)
{
    CHECK_ARG(out, "out is null");
    *out = nullptr;
    CHECK_ARG(ndev >= 1 && ndev <= 64, "ndev must be 1..64");
    CHECK_ARG(n >= 0 && ptrow && ptrow[0] == 0, "bad n / ptrow");
    for (int i = 0; i < n; i++) CHECK_ARG(ptrow[i] <= ptrow[i + 1], "ptrow must be non-decreasing");
    const long long nnz = ptrow[n];
    CHECK_ARG(nnz == 0 || (indcol && coef), "indcol/coef is null");
    int rc = need_device();
    if (rc) return rc;
    int ndevices = 0;
    HIP_TRY(hipGetDeviceCount(&ndevices));

    mi_dist_s* D = new (std::nothrow) mi_dist_s();
    if (!D) return fail(MI_ERR_ALLOC, "host allocation failed");
    const int N = ndev;
    D->nranks = N;
    D->n = n;
    D->nnz = nnz;
    D->R.resize((size_t)N);
    // rank -> device: MI355_DIST_DEVICES="0,1,2,3" (one entry per rank, reused cyclically), else rank r on device r mod (devices present):
    // with fewer devices than ranks (the one-GPU development lease) several ranks share a device and the event exchange drives the step
    std::vector<int> map;
    if (const char* e = getenv("MI355_DIST_DEVICES")) {
        for (const char* p = e; *p;) {
            char* q = nullptr;
            const long v = strtol(p, &q, 10);
            if (q == p) break;
            if (v < 0 || v >= ndevices) {
                delete D;
                return fail(MI_ERR_ARG, "MI355_DIST_DEVICES names a device that does not exist");
            }
            map.push_back((int)v);
            p = *q == ',' ? q + 1 : q;
        }
    }
    std::vector<int> devices((size_t)N);
    std::vector<char> used((size_t)ndevices, 0);
    for (int r = 0; r < N; r++) {
        devices[r] = map.empty() ? r % ndevices : map[(size_t)r % map.size()];
        D->R[r].device = devices[r];
        if (!used[devices[r]]) D->distinct_devices++;
        used[devices[r]] = 1;
    }
    // nnz-balanced contiguous row ranges; FE matrices (exact 4x4 node blocks) are cut at node boundaries, so that every rank's rows
    // keep their block structure and run the blocked kernel
    const int align = (N > 1 && csr_has_block4_pattern(n, ptrow, indcol)) ? 4 : 1;
    D->rs.assign((size_t)N + 1, 0);
    for (int r = 1; r < N; r++) {
        const long long target = nnz * r / N;
        long long cut = std::lower_bound(ptrow, ptrow + n + 1, (int)std::min<long long>(target, 0x7fffffff)) - ptrow;
        if (align > 1) cut = std::min<long long>((cut + align / 2) / align * align, (long long)n / align * align);
        D->rs[r] = std::max(D->rs[r - 1], std::min<long long>(cut, n));
    }
    D->rs[N] = n;
    for (int r = 0; r < N; r++) {
        DistRank& me = D->R[r];
        me.row0 = D->rs[r];
        me.n_local = (int)(D->rs[r + 1] - D->rs[r]);
        me.nnz0 = ptrow[me.row0];
        me.nnz_local = ptrow[D->rs[r + 1]] - me.nnz0;
    }
    D->pool.start(devices);
    // plans, in parallel (host-only integer work)
    rc = D->pool.run([&](int r) -> int {
        DistRank& me = D->R[r];
        me.ptrow.resize((size_t)me.n_local + 1);
        for (int i = 0; i <= me.n_local; i++) me.ptrow[i] = (int)(ptrow[me.row0 + i] - me.nnz0);
        return mi_part_create(N, r, D->rs.data(), me.ptrow.data(), indcol ? indcol + me.nnz0 : nullptr, coef ? coef + me.nnz0 : nullptr, &me.part);
    });
    // what rank r wants from q is what q sends to r: read off the plans (one address space: no id exchange)
    if (rc == MI_OK)
        for (int q = 0; q < N && rc == MI_OK; q++)
            for (int r = 0; r < N && rc == MI_OK; r++) {
                const PartPlan& want = D->R[r].part->plan;
                const int cnt = r == q ? 0 : want.recv_counts[q];
                rc = mi_part_set_send_ids(D->R[q].part, r, cnt, cnt ? want.halo_ids.data() + want.recv_offsets[q] : nullptr);
            }
    if (rc == MI_OK)
        rc = D->pool.run([&](int r) -> int {
            DistRank& me = D->R[r];
            {
                std::lock_guard<std::mutex> lk(g_dev_mu[me.device % 64]); // ranks sharing a device: one create-time measurement at a time
                int c = mi_part_finalize(me.part);
                if (c) return c;
            }
            const PartPlan& pl = me.part->plan;
            me.n_halo = pl.n_halo;
            for (int p = 0; p < N; p++)
                if (p != r && (pl.send_counts[p] || pl.recv_counts[p])) me.nb.push_back(p);
            HIP_TRY(hipStreamCreateWithFlags(&me.stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&me.ev_push, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&me.ev_done, hipEventDisableTiming));
            HIP_TRY(hipMalloc(&me.d_sendbuf, sizeof(double) * std::max<size_t>(pl.send_idx.size(), 1)));
            HIP_TRY(hipMalloc(&me.d_scal, 2 * sizeof(double)));
            HIP_TRY(hipMalloc(&me.d_parts, sizeof(double) * (size_t)std::max(N, 1)));
            HIP_TRY(hipHostMalloc((void**)&me.h_scal, 2 * sizeof(double), hipHostMallocDefault));
            for (hipEvent_t& e : me.ev_dot) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            std::vector<int>().swap(me.ptrow);
            return MI_OK;
        });
    if (rc == MI_OK && hipHostMalloc((void**)&D->h_parts, sizeof(double) * 2 * (size_t)std::max(N, 1), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        D->h_parts = nullptr;
        rc = fail(MI_ERR_ALLOC, "pinned table of the ranks' partial dots");
    }
    if (rc) {
        const std::string keep = g_err;
        dist_release(D);
        return fail(rc, "mi_dist_create: " + keep);
    }
    memset(D->h_parts, 0, sizeof(double) * 2 * (size_t)std::max(N, 1));
    // the exchange: MI355_DIST_EXCHANGE=event|push|rccl forces one (an unavailable one fails the create); unset / auto: push where
    // every rank has its own device, then RCCL, then events — each candidate only after the bit-for-bit self-check
    const char* xe = getenv("MI355_DIST_EXCHANGE");
    const std::string want = xe ? xe : "auto";
    if (want != "auto" && want != "event" && want != "push" && want != "rccl") {
        dist_release(D);
        return fail(MI_ERR_ARG, "MI355_DIST_EXCHANGE must be auto, event, push or rccl");
    }
    D->exchange = kExEvent;
    if (N > 1 && want != "event") {
        const bool own_devices = D->distinct_devices == N;
        for (int cand : {kExPush, kExRccl}) {
            if (want != "auto" && want != kExNames[cand]) continue;
            if (want == "auto" && !own_devices) { // ranks share devices: push is refused there, real RCCL refuses duplicate GPUs
                D->note += std::string(kExNames[cand]) + ": skipped (ranks share devices); ";
                continue;
            }
            bool ok = false, same = false;
            std::string why;
            rc = cand == kExPush ? try_push(D, &ok, &why) : try_rccl(D, &ok, &why);
            if (rc == MI_OK && ok) {
                rc = selfcheck(D, cand, &same);
                if (rc) { // the step itself failed (a wait gave up, an RCCL error): not usable
                    why = "self-check failed to run: " + g_err;
                    rc = MI_OK;
                } else if (!same) why = "self-check: results differ from the event exchange";
                if (!same && cand == kExPush) D->pool.run([&](int r) -> int { return mi_part_push_disable(D->R[r].part); });
            }
            if (rc) break;
            if (ok && same) {
                D->exchange = cand;
                D->note += std::string(kExNames[cand]) + ": ok; ";
                break;
            }
            D->note += std::string(kExNames[cand]) + ": " + (why.empty() ? "not available" : why) + "; ";
            if (want != "auto") {
                const std::string keep = D->note;
                dist_release(D);
                return fail(MI_ERR_UNSUPPORTED, "mi_dist_create: the requested exchange is not usable — " + keep);
            }
        }
        if (rc) {
            const std::string keep = g_err;
            dist_release(D);
            return fail(rc, "mi_dist_create: " + keep);
        }
    }
    if (D->exchange == kExEvent) D->note += "event: in use";
    *out = D;
    return MI_OK;
}

extern "C" int mi_dist_destroy(mi_dist_t D)
{
    if (!D) return MI_OK;
    dist_release(D);
    return MI_OK;
}

extern "C" int mi_dist_info(mi_dist_t D, int* nranks, int* distinct_devices, int* exchange, int* fused, long long* halo_total, long long* halo_max)
{
    CHECK_ARG(D, "null handle");
    if (nranks) *nranks = D->nranks;
    if (distinct_devices) *distinct_devices = D->distinct_devices;
    if (exchange) *exchange = D->exchange;
    if (fused) *fused = D->fused ? 1 : 0;
    long long tot = 0, mx = 0;
    for (const DistRank& r : D->R) {
        tot += r.n_halo;
        mx = std::max<long long>(mx, r.n_halo);
    }
    if (halo_total) *halo_total = tot;
    if (halo_max) *halo_max = mx;
    return MI_OK;
}

extern "C" const char* mi_dist_exchange_name(mi_dist_t D)
{
    if (!D) return "";
    return D->exchange == kExRccl && D->allgather ? "rccl-allgather" : kExNames[D->exchange];
}
extern "C" const char* mi_dist_exchange_note(mi_dist_t D) { return D ? D->note.c_str() : ""; }

extern "C" int mi_dist_rank_info(mi_dist_t D, int rank, int* device, long long* row_start, int* n_local, int* n_halo, long long* nnz_local,
                                 mi_stream_t* stream)
{
    CHECK_ARG(D && rank >= 0 && rank < D->nranks, "bad handle / rank");
    const DistRank& r = D->R[rank];
    if (device) *device = r.device;
    if (row_start) *row_start = r.row0;
    if (n_local) *n_local = r.n_local;
    if (n_halo) *n_halo = r.n_halo;
    if (nnz_local) *nnz_local = r.nnz_local;
    if (stream) *stream = (mi_stream_t)r.stream;
    return MI_OK;
}

extern "C" int mi_dist_update_values(mi_dist_t D, const double* coef)
{
    CHECK_ARG(D && (coef || D->nnz == 0), "null argument");
    std::lock_guard<std::mutex> lk(D->api_mu);
    return D->pool.run([&](int r) -> int {
        const DistRank& me = D->R[r];
        HIP_TRY(hipStreamSynchronize(me.stream));
        return mi_part_update_values(me.part, coef + me.nnz0);
    });
}

// ---------------------------------------------------------------- distributed vectors
extern "C" int mi_dist_vec_create(mi_dist_t D, mi_dist_vec_t* out)
{
    CHECK_ARG(D && out, "null argument");
    std::lock_guard<std::mutex> lk(D->api_mu);
    return vec_alloc(D, out);
}

extern "C" int mi_dist_vec_destroy(mi_dist_vec_t v)
{
    if (!v) return MI_OK;
    if (!v->D) { // its handle was destroyed first (mi_dist_destroy released the device memory)
        delete v;
        return MI_OK;
    }
    std::lock_guard<std::mutex> lk(v->D->api_mu);
    vec_free(v->D, v);
    return MI_OK;
}

extern "C" int mi_dist_vec_set(mi_dist_vec_t v, const double* host)
{
    CHECK_ARG(v && host, "null argument");
    std::lock_guard<std::mutex> lk(v->D->api_mu);
    int rc = vec_set(v->D, v, host);
    return rc ? rc : dist_sync(v->D); // the caller may reuse `host` at once
}

extern "C" int mi_dist_vec_get(mi_dist_vec_t v, double* host)
{
    CHECK_ARG(v && host, "null argument");
    std::lock_guard<std::mutex> lk(v->D->api_mu);
    return vec_get(v->D, v, host);
}

extern "C" int mi_dist_vec_ptr(mi_dist_vec_t v, int rank, double** d_ptr)
{
    CHECK_ARG(v && d_ptr && rank >= 0 && rank < v->D->nranks, "bad argument");
    *d_ptr = v->ext[rank];
    return MI_OK;
}

// ---------------------------------------------------------------- products on distributed vectors (asynchronous)
extern "C" int mi_dist_spmv_dev(mi_dist_t D, mi_dist_vec_t x, mi_dist_vec_t y)
{
    CHECK_ARG(D && x && y && x->D == D && y->D == D, "vectors of another handle");
    CHECK_ARG(x != y, "x and y must be different vectors");
    std::lock_guard<std::mutex> lk(D->api_mu);
    const int ex = D->exchange;
    return D->pool.run([&](int r) -> int { return dist_step(D, r, ex, x, y); });
}

extern "C" int mi_dist_spmk_dev(mi_dist_t D, int k, mi_dist_vec_t x, const mi_dist_vec_t* y_out)
{
    CHECK_ARG(D && x && y_out && x->D == D, "bad argument");
    if (k < 1 || k > MI_MAX_POWERS) return fail(MI_ERR_UNSUPPORTED, "k outside 1..MI_MAX_POWERS");
    for (int p = 0; p < k; p++) {
        CHECK_ARG(y_out[p] && y_out[p]->D == D && y_out[p] != x, "bad output vector");
        for (int q = 0; q < p; q++) CHECK_ARG(y_out[p] != y_out[q], "output vectors must be distinct");
    }
    std::lock_guard<std::mutex> lk(D->api_mu);
    const int ex = D->exchange;
    // one exchange per power (SURVEY §8(e) "k exchanges"): power p + 1 reads the owned part AND the ghosts of power p
    return D->pool.run([&](int r) -> int {
        int rc = MI_OK;
        const mi_dist_vec_s* src = x;
        for (int p = 0; p < k; p++) {
            rc = dist_step(D, r, ex, src, y_out[p], rc, p + 1 < k);
            src = y_out[p];
        }
        return rc;
    });
}

extern "C" int mi_dist_synchronize(mi_dist_t D)
{
    CHECK_ARG(D, "null handle");
    std::lock_guard<std::mutex> lk(D->api_mu);
    return dist_sync(D);
}

extern "C" int mi_dist_dot_dev(mi_dist_t D, mi_dist_vec_t a, mi_dist_vec_t b, double* out)
{
    CHECK_ARG(D && a && b && out && a->D == D && b->D == D, "bad argument");
    std::lock_guard<std::mutex> lk(D->api_mu);
    return dist_dot(D, a, b, out);
}

extern "C" int mi_dist_orthogonalize_dev(mi_dist_t D, mi_dist_vec_t b, mi_dist_vec_t x1, mi_dist_vec_t x3, double alpha, double* beta_out)
{
    CHECK_ARG(D && b && x1 && x3 && b->D == D && x1->D == D && x3->D == D && x3 != b, "bad argument");
    std::lock_guard<std::mutex> lk(D->api_mu);
    return dist_ortho(D, b, x1, x3, alpha, beta_out);
}

// ---------------------------------------------------------------- host vectors: the reference's calling convention
extern "C" int mi_dist_spmv(mi_dist_t D, const double* x, double* y)
{
    CHECK_ARG(D && (D->n == 0 || (x && y)), "null argument");
    std::lock_guard<std::mutex> lk(D->api_mu);
    int rc = need_scratch(D, 0);
    if (rc) return rc;
    if (D->n == 0) return MI_OK;
    if ((rc = vec_set(D, D->vx, x))) return rc;
    const int ex = D->exchange;
    if ((rc = D->pool.run([&](int r) -> int { return dist_step(D, r, ex, D->vx, D->vy); }))) return rc;
    return vec_get(D, D->vy, y);
}

extern "C" int mi_dist_spmk(mi_dist_t D, int k, const double* x, double* const* y_out)
{
    CHECK_ARG(D && y_out && (D->n == 0 || x), "null argument");
    if (k < 1 || k > MI_MAX_POWERS) return fail(MI_ERR_UNSUPPORTED, "k outside 1..MI_MAX_POWERS");
    for (int p = 0; p < k; p++) CHECK_ARG(y_out[p] || D->n == 0, "null output");
    std::lock_guard<std::mutex> lk(D->api_mu);
    int rc = need_scratch(D, k);
    if (rc) return rc;
    if (D->n == 0) return MI_OK;
    if ((rc = vec_set(D, D->vx, x))) return rc;
    const int ex = D->exchange;
    const mi_dist_vec_s* src = D->vx;
    for (int p = 0; p < k; p++) {
        mi_dist_vec_s* dst = D->vpow[p];
        if ((rc = D->pool.run([&](int r) -> int { return dist_step(D, r, ex, src, dst); }))) return rc;
        src = dst;
    }
    for (int p = 0; p < k; p++)
        if ((rc = vec_get(D, D->vpow[p], y_out[p]))) return rc;
    return MI_OK;
}

extern "C" int mi_dist_dot(mi_dist_t D, const double* x, const double* y, double* out)
{
    CHECK_ARG(D && out && (D->n == 0 || (x && y)), "null argument");
    std::lock_guard<std::mutex> lk(D->api_mu);
    int rc = need_scratch(D, 0);
    if (rc) return rc;
    if ((rc = vec_set(D, D->vx, x)) || (rc = vec_set(D, D->vy, y))) return rc;
    return dist_dot(D, D->vx, D->vy, out);
}

extern "C" int mi_dist_orthogonalize(mi_dist_t D, const double* b, const double* x1, double* x3, double alpha, double* beta_out)
{
    CHECK_ARG(D && (D->n == 0 || (b && x1 && x3)), "null argument");
    std::lock_guard<std::mutex> lk(D->api_mu);
    int rc = need_scratch(D, 0);
    if (rc) return rc;
    if (D->n == 0) {
        if (beta_out) *beta_out = 0.0;
        return MI_OK;
    }
    if ((rc = vec_set(D, D->vx, b)) || (rc = vec_set(D, D->vy, x1))) return rc;
    if ((rc = dist_ortho(D, D->vx, D->vy, D->vy, alpha, beta_out))) return rc; // in place on the device copy of x1
    return vec_get(D, D->vy, x3);
}
