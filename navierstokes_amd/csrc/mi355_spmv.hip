// mi355_spmv.hip — implementation of the C-ABI declared in include/mi355_spmv.h.
//
// One translation unit: kernels (spmv_kernels.hpp, blas1_kernels.hpp), the
// host-side partition planner (partition.hpp) and the handle/launch code below.
// Built for gfx950 only (navierstokes_amd/csrc/Makefile).  There is no CPU
// fallback anywhere in this file: every compute entry point needs a HIP device.
#include "mi355_spmv.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "blas1_kernels.hpp"
#include "handoff_kernels.hpp"
#include "partition.hpp"
#include "push_exchange.hpp"
#include "rccl_loader.hpp"
#include "reorder.hpp"
#include "ring_plan.hpp"
#include "spmv_kernels.hpp"
#include "spmv_ring.hpp"
#include "spmv_tile.hpp"
#include "spmv_mring.hpp"

using namespace mi355;

// ---------------------------------------------------------------- errors
static thread_local std::string g_err;

static inline void dfree(void* p)
{
    if (p) (void)hipFree(p);
}

static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            int code_ = (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorInsufficientDriver) \
                            ? MI_ERR_NODEVICE                                                           \
                            : (e_ == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP);                  \
            return fail(code_, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
        }                                                                                               \
    } while (0)

#define CHECK_ARG(cond, msg)                        \
    do {                                            \
        if (!(cond)) return fail(MI_ERR_ARG, msg);  \
    } while (0)

static int need_device()
{
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(MI_ERR_NODEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0") +
                                         " (libmi355spmv has no CPU fallback)");
    return MI_OK;
}

// ---------------------------------------------------------------- handles
struct BlockTable {
    int nnzb = 0;
    int nblk = 0;
    int2* d_blk = nullptr;  // [nblk+1]
};

struct RingTable {
    RingConfig cfg{};
    int nblk = 0, wgs = 0, bpw = 0, bad_runs = 0;
    double ok_fraction = 0.0; // share of the nonzeros in ring-served runs
    int* d_plan = nullptr; // 8 ints per block, read as two int4
    int* d_ok = nullptr;
    int* d_rng = nullptr;      // {first block, end block} per run
    int* d_run_halo = nullptr; // per run: touches a ghost column (fused multi-GPU step)
    std::vector<int> h_run_halo;
    bool uniform = true;       // runs are consecutive ranges of bpw blocks (the kernel then computes them)
    bool lean = false;         // the plan allows the LEAN instantiation (spmv_ring.hpp)
    unsigned short* d_slots = nullptr; // 16-bit column stream (ring slots), nnzb per block
    bool nt = false;                   // non-temporal loads of the values (chosen by measurement)
    bool skew = false;                 // padded staging layout (many rows with a length that is a multiple of 8)
};

// plan of the multi-window ring kernel (mring_plan.hpp); valid iff d_plan != nullptr
struct MringTable {
    int nblk = 0, wgs = 0, nruns = 0, bpw = 0, bad_runs = 0, depth = 2;
    double ok_fraction = 0.0;
    long long restarts = 0;
    int* d_plan = nullptr;             // kMringRec ints per block, read as int4
    int* d_first = nullptr;            // kMringFirst ints per run: the first block's windows
    int* d_ok = nullptr;
    int* d_rng = nullptr;
    unsigned short* d_slots = nullptr;
    bool nt = false, skew = false;
};

// plan of the tile kernel (tile_plan.hpp); valid iff d_desc != nullptr
struct TileTable {
    int nblk = 0;
    int* d_desc = nullptr;             // 4 ints per block, read as int4
    unsigned* d_ulist = nullptr;       // distinct columns per block
    unsigned short* d_slots = nullptr; // 16-bit column stream: position in the block's list
    double unique_per_nnz = 0.0;       // distinct columns per nonzero, averaged over the matrix
    bool nt = false;                   // non-temporal loads of the values
    bool skew = false;                 // padded staging layout (see RingTable::skew)
};

struct mi_csr_s {
    int device = 0;
    int n = 0, ncols = 0;
    long long nnz = 0;
    int* d_ptrow = nullptr;
    int* d_indcol = nullptr;
    double* d_coef = nullptr;
    int* d_rowmap = nullptr;
    bool mapped = false;  // created with a rowmap (device-only entry points, no powers)
    int y_offset = 0;     // a rowmap that is just "row r -> y[r + offset]" is applied as a pointer offset, not as a gather
    std::vector<int> h_ptrow; // kept to (re)build row-block tables
    std::map<int, BlockTable> tables;
    RingTable ring;           // valid iff ring.d_plan != nullptr
    TileTable tile;           // valid iff tile.d_desc != nullptr
    double tune_us_tile = 0.0, tune_us_tile_nt = 0.0;
    MringTable mring;         // valid iff mring.d_plan != nullptr
    double tune_us_mring = 0.0, tune_us_mring_nt = 0.0;
    int kernel = MI_KERNEL_AUTO;
    int auto_kernel = MI_KERNEL_STREAM;
    double tune_us_ring = 0.0, tune_us_ring_nt = 0.0, tune_us_stream = 0.0, tune_us_stream_nt = 0.0;
    double tune_us_ring_aligned = 0.0, tune_us_ring_unaligned = 0.0; // large matrices: the two block shapes (0 = not compared)
    bool stream_nt = false; // non-temporal matrix loads in the stream kernel
    mi_bcsr4_t blocked = nullptr; // BCSR 4x4 copy (exact 4x4 node-block structure only), else null
    double tune_us_bcsr = 0.0;
    int n_out = 0; // length of the y a launch may write (n, or max rowmap + 1)
    // Locality reordering (reorder.hpp): when `inner` is set, this handle is a front for A' = P A P^T, a row-mapped
    // handle in the new numbering; products gather x into d_xp (new numbering) and inner writes y through its row
    // map straight into the caller's numbering.  The natural-order device arrays are released then.
    mi_csr_t inner = nullptr;
    int* d_iperm = nullptr;     // [n] caller's index of new row / column
    int* d_src_start = nullptr; // [n] offset of new row r' in the caller's coef (values refresh)
    double* d_xp = nullptr;     // x in the new numbering (one product at a time per handle)
    std::vector<double*> d_pp;  // powers in the new numbering
    double* d_vtmp = nullptr;   // staging for mi_csr_update_values (host values)
    double spread_before = 0.0, spread_after = 0.0, us_natural = 0.0, us_reordered = 0.0;
    int reorder_block = 0;      // 0: no reordering attempted
    // scratch for the host-pointer entry points
    double* d_x = nullptr;
    double* d_y = nullptr;
    std::vector<double*> d_pow;
};

struct mi_bcsr4_s {
    int device = 0;
    int nbrows = 0, nbcols = 0;
    long long nblocks = 0;
    int* d_ptrow = nullptr;
    int* d_indcol = nullptr;
    double* d_coef = nullptr;
    int* d_browmap = nullptr; // block-row map of a reordered matrix's blocked copy, else null
    // x tile per workgroup (spmv_bcsr4_tile): lists of distinct block columns and 16-bit positions; null if not built
    int* d_tl_ptr = nullptr;
    unsigned* d_tl_nodes = nullptr;
    unsigned short* d_tl_slots = nullptr;
    bool use_tile = false;    // the measured choice between the two kernels (MI355_BCSR_TILE=0|1 forces)
    double tune_us_plain = 0.0, tune_us_tile = 0.0;
    double* d_x = nullptr;
    double* d_y = nullptr;
    std::vector<double*> d_pow;
};

struct mi_part_s {
    PartPlan plan;
    mi_csr_t piece[2] = {nullptr, nullptr};
    int* d_send_idx = nullptr;
    bool finalized = false;
    int kernel = MI_KERNEL_AUTO;
    // native exchange (mi_part_comm_init)
    void* comm = nullptr; // ncclComm_t
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_pack = nullptr, ev_comm = nullptr;
    double* d_sendbuf = nullptr;
    unsigned* d_flags = nullptr; // [0] "x ready" (set on the caller's stream), [1] "halo rows done" (set on the comm stream)
    unsigned* h_timeouts = nullptr; // pinned, device-mapped: hand-off waits that gave up (read by the host at every entry point)
    unsigned* d_timeouts = nullptr; // the device address of h_timeouts
    unsigned step_no = 0;
    // Hand-offs between the two streams: HIP events by default.  Flag kernels (handoff_kernels.hpp) are ~8 us per
    // step cheaper in the one-GPU harness but have not run against real multi-GPU RCCL yet: opt in with
    // MI355_PART_HANDOFF=flags.
    bool flag_handoff = false;
    // peer-push exchange (push_exchange.hpp): my receive window, the peers' windows I write to
    void* win = nullptr;          // [flags: nranks x 64 B][pad][data: 2 x n_halo doubles]
    bool win_uncached = false, win_registered = false;
    std::string win_key;          // the IPC handle bytes (key of the in-process registry)
    unsigned* win_flags = nullptr;
    double* win_data = nullptr;
    std::vector<void*> ipc_opened; // mappings to close
    PushLink* d_links = nullptr;
    int n_links = 0;
    int2* d_push_work = nullptr;   // stand-alone push kernel: {link, chunk} per workgroup
    int* d_link_chunks = nullptr;  // chunks per link
    unsigned* d_tickets = nullptr; // per link: chunks out so far
    int n_push_work = 0;
    int* d_nb = nullptr;           // ranks whose flags I wait for
    int n_nb = 0;
    unsigned push_step = 0;
    bool push_ready = false;
    // the one-launch form of the push step (spmv_ring.hpp, FUSED): all local rows in one ring-served, row-mapped piece
    mi_csr_t piece_all = nullptr;
    int* d_run_link = nullptr; // per run of piece_all: first push link it serves, or -1
    int npush_runs = 0;
    bool fused = false;
    bool fused_bcsr = false;   // piece_all is served by the BCSR kernel: spmv_bcsr4_fused
    int* d_wg_halo = nullptr;  // per workgroup of that launch: its block rows touch a ghost node
};

// windows of ranks living in THIS process (rank threads; hipIpcOpenMemHandle refuses a handle of the opening process)
static std::map<std::string, void*> g_win_registry;
static size_t win_data_offset(int nranks) { return ((size_t)nranks * kWinFlagStride * sizeof(unsigned) + 255) / 256 * 256; }

static void part_comm_release(mi_part_s* P);

// Reduction workspace: partials of the two-stage reductions, one per (device, stream) so that
// reductions enqueued on different streams (or by different rank threads of one process) never share
// partials.  4 * kMaxPartials doubles: [0, 2K) the two partial arrays of a reduction (or the
// ping-pong pair of the Gram-Schmidt sweep), the rest spare.  32 KB per stream that ever reduced; a
// destroyed stream's slot is simply reused if the runtime hands the same handle out again.
// g_mu guards every process-wide table of this file (workspaces, flush buffers).
static std::mutex g_mu;
static std::map<std::pair<int, hipStream_t>, double*> g_ws;

static int get_ws(hipStream_t s, double** out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mu);
    double*& w = g_ws[std::make_pair(dev, s)];
    if (!w) HIP_TRY(hipMalloc(&w, sizeof(double) * (4 * kMaxPartials + 8)));
    *out = w;
    return MI_OK;
}

// ---------------------------------------------------------------- library
extern "C" int mi_version(void) { return MI355_SPMV_VERSION; }

extern "C" const char* mi_strerror(int status)
{
    switch (status) {
    case MI_OK: return "ok";
    case MI_ERR_ARG: return "invalid argument";
    case MI_ERR_NODEVICE: return "no HIP device (no CPU fallback)";
    case MI_ERR_HIP: return "HIP runtime error";
    case MI_ERR_ALLOC: return "allocation failed";
    case MI_ERR_UNSUPPORTED: return "unsupported";
    case MI_ERR_STATE: return "bad handle state";
    default: return "unknown status";
    }
}

extern "C" const char* mi_last_error(void) { return g_err.c_str(); }

extern "C" int mi_device_count(int* count)
{
    CHECK_ARG(count, "count is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *count = (e == hipSuccess) ? c : 0;
    return MI_OK;
}

extern "C" int mi_set_device(int device)
{
    int rc = need_device();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    return MI_OK;
}

extern "C" int mi_device_synchronize(void)
{
    int rc = need_device();
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return MI_OK;
}

static std::map<int, void*> g_flush;

// read sweep: leaves the caches full of CLEAN lines of a buffer nobody uses
__global__ __launch_bounds__(256) void flush_read_kernel(const double2* __restrict__ p, size_t n16, double* __restrict__ sink)
{
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const double2 v = p[i];
        s += v.x + v.y;
    }
    if (s == 123.456) sink[0] = s; // never true: keeps the loads alive
}

static int flush_cache_on(hipStream_t st, bool sync);

extern "C" int mi_flush_cache(void) { return flush_cache_on(nullptr, true); }

extern "C" int mi_flush_cache_async(mi_stream_t s) { return flush_cache_on((hipStream_t)s, false); }

static int flush_cache_on(hipStream_t st, bool sync)
{
    int rc = need_device();
    if (rc) return rc;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const size_t bytes = (size_t)512 << 20;
    void* buf = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        void*& slot = g_flush[dev];
        if (!slot) HIP_TRY(hipMalloc(&slot, 2 * bytes + 256));
        buf = slot;
    }
    // Write 512 MiB (as the reference's flush_cache writes its buffer, mpk/utils.cpp:146-154), then READ another 512 MiB:
    // the fill alone would leave the 256 MiB Infinity Cache full of DIRTY lines whose write-back the next kernel then pays
    // for (measured: a cold C4 product 210 us behind the fill alone); behind the read sweep the caches hold clean lines of
    // a buffer nobody uses, i.e. "nothing of the caller's data is cached" and nothing else.
    HIP_TRY(hipMemsetAsync(buf, 1, bytes, st));
    const char* only_fill = getenv("MI355_FLUSH_FILL_ONLY");
    if (!(only_fill && !strcmp(only_fill, "1")))
        hipLaunchKernelGGL(flush_read_kernel, dim3(4096), dim3(256), 0, st, reinterpret_cast<const double2*>((char*)buf + bytes), bytes / 16,
                           reinterpret_cast<double*>((char*)buf + 2 * bytes));
    HIP_TRY(hipGetLastError());
    if (sync) HIP_TRY(hipDeviceSynchronize());
    return MI_OK;
}

// ---------------------------------------------------------------- CSR create
static int get_table(mi_csr_t A, int nnzb, BlockTable** out)
{
    BlockTable& T = A->tables[nnzb];
    if (!T.d_blk) {
        std::vector<int> rows, ptrs;
        // whole waves of rows for the one-thread-per-row chain phase where that keeps 7/8 of the block (ring_plan.hpp) — for
        // matrices that live in the Infinity Cache: S15 1 M rows 61 -> 56 us, but the 5 M-row mesh 180 -> 193 us
        // (tools/stream_align_ab.py); MI355_STREAM_ROW_ALIGN=1|64 forces (A/B)
        int row_align = A->nnz < 20000000 ? 64 : 1;
        if (const char* e = getenv("MI355_STREAM_ROW_ALIGN")) row_align = std::max(1, atoi(e));
        build_row_blocks(A->n, A->h_ptrow.data(), nnzb, 4 * kWG, rows, ptrs, row_align, 7);
        T.nnzb = nnzb;
        T.nblk = (int)rows.size() - 1;
        std::vector<int2> h(rows.size());
        for (size_t i = 0; i < rows.size(); i++) h[i] = make_int2(rows[i], ptrs[i]);
        HIP_TRY(hipMalloc(&T.d_blk, sizeof(int2) * h.size()));
        HIP_TRY(hipMemcpy(T.d_blk, h.data(), sizeof(int2) * h.size(), hipMemcpyHostToDevice));
    }
    *out = &T;
    return MI_OK;
}

static void free_tile(mi_csr_t A)
{
    dfree(A->tile.d_desc);
    dfree(A->tile.d_ulist);
    dfree(A->tile.d_slots);
    A->tile = TileTable();
}

static void free_mring(mi_csr_t A)
{
    dfree(A->mring.d_plan);
    dfree(A->mring.d_first);
    dfree(A->mring.d_ok);
    dfree(A->mring.d_rng);
    dfree(A->mring.d_slots);
    A->mring = MringTable();
}

// Plan of the multi-window ring kernel (host arrays of the caller, or nullptr: the handle's device copy is read back)
static int build_mring(mi_csr_t A, const int* indcol, int row_align = 0)
{
    if (A->mring.d_plan || A->n == 0 || A->nnz == 0) return MI_OK;
    std::vector<int> back;
    if (!indcol) {
        if (!A->d_indcol) return fail(MI_ERR_STATE, "mring plan: the handle no longer holds its column indices");
        back.resize((size_t)A->nnz);
        HIP_TRY(hipMemcpy(back.data(), A->d_indcol, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToHost));
        indcol = back.data();
    }
    MringPlanHost P;
    build_mring_plan(A->n, A->h_ptrow.data(), indcol, P, row_align);
    MringTable& M = A->mring;
    hipError_t e;
    if ((e = hipMalloc(&M.d_plan, sizeof(int) * P.plan.size())) != hipSuccess ||
        (e = hipMalloc(&M.d_first, sizeof(int) * P.first.size())) != hipSuccess ||
        (e = hipMemcpy(M.d_first, P.first.data(), sizeof(int) * P.first.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMalloc(&M.d_ok, sizeof(int) * P.run_ok.size())) != hipSuccess ||
        (e = hipMalloc(&M.d_rng, sizeof(int) * P.run_rng.size())) != hipSuccess ||
        (e = hipMalloc(&M.d_slots, sizeof(unsigned short) * P.slots.size())) != hipSuccess ||
        (e = hipMemcpy(M.d_plan, P.plan.data(), sizeof(int) * P.plan.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(M.d_ok, P.run_ok.data(), sizeof(int) * P.run_ok.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(M.d_rng, P.run_rng.data(), sizeof(int) * P.run_rng.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(M.d_slots, P.slots.data(), sizeof(unsigned short) * P.slots.size(), hipMemcpyHostToDevice)) != hipSuccess) {
        free_mring(A);
        return fail(e == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP, std::string("mring plan upload: ") + hipGetErrorString(e));
    }
    M.nblk = P.nblk;
    M.wgs = P.wgs;
    M.nruns = P.nruns;
    M.bpw = P.bpw;
    M.bad_runs = P.bad_runs;
    M.restarts = P.restarts;
    M.ok_fraction = 1.0 - (double)P.bad_nnz / (double)A->nnz;
    M.depth = P.bpw >= 40 ? 4 : 2; // as for the single ring (tools/depth_ab.py)
    long long mult8 = 0;
    for (int i = 0; i < A->n; i++) {
        const int len = A->h_ptrow[i + 1] - A->h_ptrow[i];
        mult8 += len > 0 && len % 8 == 0;
    }
    M.skew = 10 * mult8 > A->n;
    M.nt = 10.0 * (double)A->nnz + 16.0 * (double)A->n > 0.75 * 256e6;
    return MI_OK;
}

// Plan of the tile kernel for this handle's pattern (host arrays of the caller, or nullptr: the handle's own device
// copy is read back — explicit MI_KERNEL_TILE requests on a handle created without it).
static int build_tile(mi_csr_t A, const int* indcol)
{
    if (A->tile.d_desc || A->n == 0 || A->nnz == 0) return MI_OK;
    std::vector<int> back;
    if (!indcol) {
        if (!A->d_indcol) return fail(MI_ERR_STATE, "tile plan: the handle no longer holds its column indices");
        back.resize((size_t)A->nnz);
        HIP_TRY(hipMemcpy(back.data(), A->d_indcol, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToHost));
        indcol = back.data();
    }
    TilePlanHost P;
    build_tile_plan(A->n, A->h_ptrow.data(), indcol, P, kTileNnzb, 0, A->nnz < 20000000 ? 64 : 1);
    TileTable& T = A->tile;
    hipError_t e;
    if ((e = hipMalloc(&T.d_desc, sizeof(int) * P.desc.size())) != hipSuccess ||
        (e = hipMalloc(&T.d_ulist, sizeof(unsigned) * P.ulist.size())) != hipSuccess ||
        (e = hipMalloc(&T.d_slots, sizeof(unsigned short) * P.slots.size())) != hipSuccess ||
        (e = hipMemcpy(T.d_desc, P.desc.data(), sizeof(int) * P.desc.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(T.d_ulist, P.ulist.data(), sizeof(unsigned) * P.ulist.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(T.d_slots, P.slots.data(), sizeof(unsigned short) * P.slots.size(), hipMemcpyHostToDevice)) != hipSuccess) {
        free_tile(A);
        return fail(e == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP, std::string("tile plan upload: ") + hipGetErrorString(e));
    }
    T.nblk = P.nblk;
    T.unique_per_nnz = (double)(P.ulist.size() - kTileThreads) / (double)A->nnz;
    long long mult8 = 0;
    for (int i = 0; i < A->n; i++) {
        const int len = A->h_ptrow[i + 1] - A->h_ptrow[i];
        mult8 += len > 0 && len % 8 == 0;
    }
    T.skew = 10 * mult8 > A->n;
    T.nt = 10.0 * (double)A->nnz + 16.0 * (double)A->n > 0.75 * 256e6;
    return MI_OK;
}

static int launch_spmv(mi_csr_t A, const double* d_x, double* d_y, hipStream_t s, bool use_map = true, const RingComm* comm = nullptr);
static int launch_bcsr4(mi_bcsr4_t A, const double* d_x, double* d_y, mi_stream_t s, bool use_map);
static int resolve_kernel(const mi_csr_s* A);

static void free_ring_table(RingTable& R)
{
    dfree(R.d_plan);
    dfree(R.d_ok);
    dfree(R.d_rng);
    dfree(R.d_run_halo);
    dfree(R.d_slots);
    R = RingTable();
}

// device copy of a ring plan (plan records, run tables, 16-bit column stream) and what the launch needs to know about it
static int fill_ring_table(RingTable& R, const RingPlanHost& best, int n, const int* ptrow, const int* indcol, long long nnz, bool ghosts)
{
    R.cfg = best.cfg;
    // Blocks of prefetch: with long runs (C4: 72 blocks per workgroup) four blocks in flight instead of two hide more of
    // the HBM latency — same handle, same box, back to back 172.8 / 168.8 / 167.5 us at depth 2 / 3 / 4, cold caches
    // 198.2 / 194.2 / 191.9 us; with short runs (1 M rows: 14 blocks) the longer pipeline fill costs more than it hides:
    // 34.7 / 35.1 / 37.3 us (tools/depth_ab.py, profiles/r02_ring_depth_ab.txt).  Configuration 4 only.
    if (best.cfg.id == 4 && best.bpw >= 40) R.cfg.depth = 4;
    if (const char* e = getenv("MI355_RING_DEPTH")) {
        const int d = atoi(e);
        if (best.cfg.id == 4 && d >= 2 && d <= 4) R.cfg.depth = d;
    }
    R.nblk = best.nblk;
    R.wgs = best.wgs;
    R.bpw = best.bpw;
    R.bad_runs = best.bad_runs;
    R.ok_fraction = nnz ? 1.0 - (double)best.bad_nnz / (double)nnz : 0.0;
    R.lean = best.lean && best.cfg.id == 4 && !(getenv("MI355_RING_LEAN") && !strcmp(getenv("MI355_RING_LEAN"), "0"));
    if (best.nblk <= 0) return MI_OK;
    hipError_t e;
#define RING_TRY(expr)                                                                                                      \
    if ((e = (expr)) != hipSuccess) {                                                                                       \
        free_ring_table(R);                                                                                                 \
        return fail(e == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e)); \
    }
    RING_TRY(hipMalloc(&R.d_plan, sizeof(int) * best.plan.size()));
    RING_TRY(hipMemcpy(R.d_plan, best.plan.data(), sizeof(int) * best.plan.size(), hipMemcpyHostToDevice));
    RING_TRY(hipMalloc(&R.d_ok, sizeof(int) * best.run_ok.size()));
    RING_TRY(hipMemcpy(R.d_ok, best.run_ok.data(), sizeof(int) * best.run_ok.size(), hipMemcpyHostToDevice));
    for (int g = 0; g < best.wgs; g++)
        R.uniform = R.uniform && best.run_rng[2 * g] == std::min(best.nblk, g * best.bpw) &&
                    best.run_rng[2 * g + 1] == std::min(best.nblk, (g + 1) * best.bpw);
    RING_TRY(hipMalloc(&R.d_rng, sizeof(int) * best.run_rng.size()));
    RING_TRY(hipMemcpy(R.d_rng, best.run_rng.data(), sizeof(int) * best.run_rng.size(), hipMemcpyHostToDevice));
    if (ghosts) {
        R.h_run_halo = best.run_halo;
        RING_TRY(hipMalloc(&R.d_run_halo, sizeof(int) * best.run_halo.size()));
        RING_TRY(hipMemcpy(R.d_run_halo, best.run_halo.data(), sizeof(int) * best.run_halo.size(), hipMemcpyHostToDevice));
    }
    {
        std::vector<unsigned short> slots;
        build_ring_slots(best, indcol, slots);
        RING_TRY(hipMalloc(&R.d_slots, sizeof(unsigned short) * slots.size()));
        RING_TRY(hipMemcpy(R.d_slots, slots.data(), sizeof(unsigned short) * slots.size(), hipMemcpyHostToDevice));
    }
#undef RING_TRY
    // staging layout of the row chains: plain unless more than a tenth of the rows have a length
    // that is a multiple of 8 (their LDS segments would start on the same two banks)
    long long mult8 = 0;
    for (int i = 0; i < n; i++) {
        const int len = ptrow[i + 1] - ptrow[i];
        mult8 += len > 0 && len % 8 == 0;
    }
    R.skew = 10 * mult8 > n;
    if (const char* e2 = getenv("MI355_RING_SKEW")) R.skew = atoi(e2) != 0;
    return MI_OK;
}

static int time_handle(mi_csr_t A, int warm, int timed, double* us);

static int csr_create_impl(int n, int ncols, const int* ptrow, const int* indcol, const double* coef,
                           const int* rowmap, mi_csr_t* out, int ghost_lo = 0, int ghost_hi = 0)
{
    CHECK_ARG(out, "out is null");
    *out = nullptr;
    CHECK_ARG(n >= 0 && ncols >= 0, "negative dimension");
    CHECK_ARG(ptrow, "ptrow is null");
    CHECK_ARG(ptrow[0] == 0, "ptrow[0] must be 0");
    for (int i = 0; i < n; i++) CHECK_ARG(ptrow[i] <= ptrow[i + 1], "ptrow must be non-decreasing");
    const long long nnz = ptrow[n];
    CHECK_ARG(nnz == 0 || (indcol && coef), "indcol/coef is null");
    // 32-bit element offsets inside the kernels, padding included (ring_plan.hpp)
    CHECK_ARG(nnz <= 0x7fffffffLL - 2 * kRingPadNnz && n <= 0x7fffffff - 2 * kRingPadRows, "matrix too large for 32-bit offsets: partition it (mi_part_*)");
    std::vector<int> row_min((size_t)n), row_max((size_t)n);
    for (int i = 0; i < n; i++) {
        int lo = 0x7fffffff, hi = -1;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) {
            const int c = indcol[k];
            CHECK_ARG(c >= 0 && c < ncols, "column index outside [0, ncols)");
            lo = c < lo ? c : lo;
            hi = c > hi ? c : hi;
        }
        row_min[i] = lo;
        row_max[i] = hi;
    }
    int rc = need_device();
    if (rc) return rc;

    mi_csr_t A = new (std::nothrow) mi_csr_s();
    if (!A) return fail(MI_ERR_ALLOC, "host allocation failed");
    A->n = n;
    A->ncols = ncols;
    A->nnz = nnz;
    A->h_ptrow.assign(ptrow, ptrow + n + 1);
    hipError_t e = hipGetDevice(&A->device);
    // zero padding behind the arrays: the kernels' unclamped / vector loads may touch it (ring_plan.hpp)
    const size_t pad = 8, padv = kRingPadNnz, padr = kRingPadRows;
    auto cleanup = [&]() { mi_csr_destroy(A); };
#define TRY_OR_CLEAN(expr)                                                          \
    do {                                                                            \
        hipError_t e2_ = (expr);                                                    \
        if (e2_ != hipSuccess) {                                                    \
            cleanup();                                                              \
            return fail(e2_ == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP,     \
                        std::string(#expr) + ": " + hipGetErrorString(e2_));        \
        }                                                                           \
    } while (0)
    TRY_OR_CLEAN(e);
    TRY_OR_CLEAN(hipMalloc(&A->d_ptrow, sizeof(int) * ((size_t)n + 1 + padr)));
    TRY_OR_CLEAN(hipMalloc(&A->d_indcol, sizeof(int) * ((size_t)nnz + pad)));
    TRY_OR_CLEAN(hipMalloc(&A->d_coef, sizeof(double) * ((size_t)nnz + padv)));
    TRY_OR_CLEAN(hipMemset(A->d_ptrow + n + 1, 0, sizeof(int) * padr));
    TRY_OR_CLEAN(hipMemset(A->d_indcol + nnz, 0, sizeof(int) * pad));
    TRY_OR_CLEAN(hipMemset(A->d_coef + nnz, 0, sizeof(double) * padv));
    TRY_OR_CLEAN(hipMemcpy(A->d_ptrow, ptrow, sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice));
    if (nnz) {
        TRY_OR_CLEAN(hipMemcpy(A->d_indcol, indcol, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        TRY_OR_CLEAN(hipMemcpy(A->d_coef, coef, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
    }
    A->mapped = rowmap != nullptr;
    bool offset_only = rowmap != nullptr && n > 0;
    for (int i = 1; offset_only && i < n; i++) offset_only = rowmap[i] == rowmap[0] + i;
    if (offset_only) A->y_offset = rowmap[0]; // e.g. the interior rows of a banded partition: one contiguous range
    if (rowmap && n > 0 && !offset_only) {
        TRY_OR_CLEAN(hipMalloc(&A->d_rowmap, sizeof(int) * ((size_t)n + padr)));
        TRY_OR_CLEAN(hipMemset(A->d_rowmap + n, 0, sizeof(int) * padr));
        TRY_OR_CLEAN(hipMemcpy(A->d_rowmap, rowmap, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    }
    // window plan of the ring kernel: first configuration (in preference order) that
    // serves at least 90 % of the nonzeros; MI355_RING_CONFIG=1..4 forces one
    if (n > 0 && nnz > 0) {
        const int* order = kRingConfigOrder;
        int forced = 0;
        if (const char* e = getenv("MI355_RING_CONFIG")) forced = atoi(e);
        RingPlanHost best;
        bool have = false;
        for (int t = 0; t < kNumRingConfigs && !have; t++) {
            const int id = forced >= 1 && forced <= kNumRingConfigs ? forced : order[t];
            RingPlanHost P;
            build_ring_plan(kRingConfigs[id - 1], n, ptrow, row_min.data(), row_max.data(), P, ghost_lo, ghost_hi);
            const double okf = 1.0 - (double)P.bad_nnz / (double)nnz;
            if (forced || okf >= 0.90) {
                best = std::move(P);
                have = true;
            } else if (t == 0) {
                best = std::move(P); // remember the preferred one for explicit MI_KERNEL_RING requests
            }
            if (forced) break;
        }
        {
            const int rcr = fill_ring_table(A->ring, best, n, ptrow, indcol, nnz, ghost_lo < ghost_hi);
            if (rcr != MI_OK) {
                mi_csr_destroy(A);
                return rcr;
            }
        }
        A->auto_kernel = (have && A->ring.ok_fraction >= 0.90) ? MI_KERNEL_RING : MI_KERNEL_STREAM;
    }
    // wide-band matrices (the ring does not serve them): the tile kernel's plan, kept if neighbouring rows share
    // enough columns for it to pay (tile_plan.hpp); MI355_TILE=0 never, =1 always
    {
        const char* te = getenv("MI355_TILE");
        const char* ke = getenv("MI355_SPMV_KERNEL");
        const bool asked = (te && !strcmp(te, "1")) || (ke && !strcmp(ke, "tile"));
        if (n > 0 && nnz > 0 && !(te && !strcmp(te, "0")) && (asked || (A->auto_kernel != MI_KERNEL_RING && nnz >= 200000))) {
            const int rct = build_tile(A, indcol);
            if (rct != MI_OK) {
                mi_csr_destroy(A);
                return rct;
            }
            if (!asked && A->tile.unique_per_nnz > 0.6) free_tile(A); // little sharing: nothing to gain over the stream kernel
        }
    }
    // ... and the multi-window ring's (3-D mesh operators: a few narrow column clusters far apart), kept if it serves >= 90 %
    {
        const char* me = getenv("MI355_MRING");
        const char* ke = getenv("MI355_SPMV_KERNEL");
        const bool asked = (me && !strcmp(me, "1")) || (ke && !strcmp(ke, "mring"));
        if (n > 0 && nnz > 0 && !(me && !strcmp(me, "0")) && (asked || (A->auto_kernel != MI_KERNEL_RING && nnz >= 200000))) {
            const int rcm = build_mring(A, indcol);
            if (rcm != MI_OK) {
                mi_csr_destroy(A);
                return rcm;
            }
            if (!asked && A->mring.ok_fraction < 0.90) free_mring(A);
        }
    }
    // FE matrices: a blocked copy for the BCSR 4x4 kernel (same bits, 8.25 instead of 12 B per nonzero)
    // (a row map that moves whole nodes — rowmap[4b + q] = rowmap[4b] + q, 4-aligned — becomes a block-row map)
    bool node_map = rowmap != nullptr && !offset_only && n % 4 == 0;
    for (int b = 0; node_map && b < n / 4; b++)
        node_map = rowmap[4 * b] % 4 == 0 && rowmap[4 * b + 1] == rowmap[4 * b] + 1 && rowmap[4 * b + 2] == rowmap[4 * b] + 2 &&
                   rowmap[4 * b + 3] == rowmap[4 * b] + 3;
    if ((!rowmap || offset_only || node_map) && n >= 4 && nnz >= 16 && ncols % 4 == 0 && !(getenv("MI355_AUTO_BCSR") && !strcmp(getenv("MI355_AUTO_BCSR"), "0"))) {
        std::vector<int> bptr, bcol;
        std::vector<double> bval;
        if (csr_to_bcsr4_exact(n, ptrow, indcol, coef, bptr, bcol, bval)) {
            const int rcb = mi_bcsr4_create(n / 4, ncols / 4, bptr.data(), bcol.data(), bval.data(), &A->blocked);
            if (rcb != MI_OK) {
                mi_csr_destroy(A);
                return rcb;
            }
            if (node_map) {
                std::vector<int> bmap((size_t)n / 4);
                for (int b = 0; b < n / 4; b++) bmap[b] = rowmap[4 * b] / 4;
                TRY_OR_CLEAN(hipMalloc(&A->blocked->d_browmap, sizeof(int) * bmap.size()));
                TRY_OR_CLEAN(hipMemcpy(A->blocked->d_browmap, bmap.data(), sizeof(int) * bmap.size(), hipMemcpyHostToDevice));
            }
        }
    }
    // default for the value loads when nothing is measured: non-temporal once the matrix stream
    // (10 B per nonzero) no longer fits the 256 MB Infinity Cache with room for the vectors
    A->ring.nt = A->ring.d_slots && 10.0 * (double)nnz + 16.0 * (double)n > 0.75 * 256e6;
    if (const char* e = getenv("MI355_RING_NT")) A->ring.nt = A->ring.d_slots && atoi(e) != 0;
    A->stream_nt = 12.0 * (double)nnz + 16.0 * (double)n > 0.75 * 256e6;
    if (const char* e = getenv("MI355_STREAM_NT")) A->stream_nt = atoi(e) != 0;
    if (const char* e = getenv("MI355_TILE_NT")) A->tile.nt = atoi(e) != 0;
    if (const char* e = getenv("MI355_MRING_NT")) A->mring.nt = atoi(e) != 0;
    if (A->blocked) A->auto_kernel = MI_KERNEL_BCSR4; // unless measured otherwise below
    A->n_out = n;
    if (rowmap)
        for (int i = 0; i < n; i++) A->n_out = rowmap[i] + 1 > A->n_out ? rowmap[i] + 1 : A->n_out;
    bool forced_kernel = false;
    if (const char* e = getenv("MI355_SPMV_KERNEL")) {
        forced_kernel = true;
        if (!strcmp(e, "stream")) A->auto_kernel = MI_KERNEL_STREAM;
        else if (!strcmp(e, "ring") && A->ring.d_plan) A->auto_kernel = MI_KERNEL_RING;
        else if (!strcmp(e, "rowpar")) A->auto_kernel = MI_KERNEL_ROWPAR;
        else if (!strcmp(e, "bcsr4") && A->blocked) A->auto_kernel = MI_KERNEL_BCSR4;
        else if (!strcmp(e, "tile") && A->tile.d_desc) A->auto_kernel = MI_KERNEL_TILE;
        else if (!strcmp(e, "mring") && A->mring.d_plan) A->auto_kernel = MI_KERNEL_MRING;
        else forced_kernel = false;
    }
    const char* at = getenv("MI355_SPMV_AUTOTUNE");
    if (!forced_kernel && !(at && !strcmp(at, "0")) && nnz >= 200000) {
        // measure the candidates on this very matrix (x = 0: timing does not depend on the values):
        // ring (if it serves the matrix) and stream, each with temporal and non-temporal matrix loads
        struct TuneScratch { // released on every exit path, the early error returns of TRY_OR_CLEAN included
            double *tx = nullptr, *ty = nullptr;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            ~TuneScratch()
            {
                dfree(tx);
                dfree(ty);
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
            }
        } ts;
        double *&tx = ts.tx, *&ty = ts.ty;
        hipEvent_t &e0 = ts.e0, &e1 = ts.e1;
        TRY_OR_CLEAN(hipMalloc(&tx, sizeof(double) * (size_t)(ncols > 0 ? ncols : 1)));
        TRY_OR_CLEAN(hipMalloc(&ty, sizeof(double) * (size_t)(A->n_out > 0 ? A->n_out : 1)));
        TRY_OR_CLEAN(hipMemset(tx, 0, sizeof(double) * (size_t)(ncols > 0 ? ncols : 1)));
        TRY_OR_CLEAN(hipEventCreate(&e0));
        TRY_OR_CLEAN(hipEventCreate(&e1));
        const bool ring_ok = A->auto_kernel == MI_KERNEL_RING;
        const bool ring_nt_forced = getenv("MI355_RING_NT") != nullptr, stream_nt_forced = getenv("MI355_STREAM_NT") != nullptr;
        const bool ring_nt0 = A->ring.nt, stream_nt0 = A->stream_nt;
        const bool tile_nt_forced = getenv("MI355_TILE_NT") != nullptr;
        const bool tile_nt0 = A->tile.nt;
        const bool mring_nt_forced = getenv("MI355_MRING_NT") != nullptr;
        const bool mring_nt0 = A->mring.nt;
        double us[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // ring, ring nt, stream, stream nt, tile, tile nt, mring, mring nt
        // two interleaved rounds, the faster of the two counts: one round is not enough to tell two
        // candidates 5 % apart from each other (clock ramps, what the previous candidate left in the caches)
        for (int round = 0; round < 2; round++)
            for (int c = 0; c < 8; c++) {
                const bool nt = c & 1;
                if (c >= 6) {
                    if (!A->mring.d_plan || (mring_nt_forced && nt != mring_nt0)) continue;
                    A->mring.nt = nt;
                    A->kernel = MI_KERNEL_MRING;
                } else if (c >= 4) {
                    if (!A->tile.d_desc || (tile_nt_forced && nt != tile_nt0)) continue;
                    A->tile.nt = nt;
                    A->kernel = MI_KERNEL_TILE;
                } else if (c < 2) {
                    if (!ring_ok || (nt && !A->ring.d_slots) || (ring_nt_forced && nt != ring_nt0)) continue;
                    A->ring.nt = nt;
                    A->kernel = MI_KERNEL_RING;
                } else {
                    if (stream_nt_forced && nt != stream_nt0) continue;
                    if (round == 1 && ring_ok && us[c] > 1.25 * std::min(us[0] > 0 ? us[0] : us[1], us[1] > 0 ? us[1] : us[0]))
                        continue; // stream is out of the race already
                    A->stream_nt = nt;
                    A->kernel = MI_KERNEL_STREAM;
                }
                // warm launches first: a temporal candidate is judged with the Infinity Cache holding
                // what it can of the matrix, as it would between the iterations of a solver
                const int warm = 3, timed = nnz < 40000000 ? 12 : 6;
                for (int w = 0; w < warm; w++)
                    if (launch_spmv(A, tx, ty, nullptr) != MI_OK) break;
                TRY_OR_CLEAN(hipEventRecord(e0, nullptr));
                for (int w = 0; w < timed; w++)
                    if (launch_spmv(A, tx, ty, nullptr) != MI_OK) break;
                TRY_OR_CLEAN(hipEventRecord(e1, nullptr));
                TRY_OR_CLEAN(hipEventSynchronize(e1));
                float ms = 0.f;
                TRY_OR_CLEAN(hipEventElapsedTime(&ms, e0, e1));
                const double t = ms * 1e3 / timed;
                us[c] = us[c] > 0 ? std::min(us[c], t) : t;
            }
        A->kernel = MI_KERNEL_AUTO;
        A->tune_us_ring = us[0];
        A->tune_us_ring_nt = us[1];
        A->tune_us_stream = us[2];
        A->tune_us_stream_nt = us[3];
        auto better = [](double a, double b) { return a > 0 && (b <= 0 || a < b); }; // a measured and faster than b
        A->ring.nt = ring_ok ? better(us[1], us[0]) : ring_nt0;
        A->stream_nt = better(us[3], us[2]);
        A->tune_us_tile = us[4];
        A->tune_us_tile_nt = us[5];
        A->tile.nt = A->tile.d_desc ? better(us[5], us[4]) : tile_nt0;
        const double best_ring = A->ring.nt ? us[1] : us[0], best_stream = A->stream_nt ? us[3] : us[2];
        const double best_tile = A->tile.nt ? us[5] : us[4];
        if (ring_ok && better(best_stream, best_ring)) A->auto_kernel = MI_KERNEL_STREAM;
        if (better(best_tile, A->auto_kernel == MI_KERNEL_RING ? best_ring : best_stream)) A->auto_kernel = MI_KERNEL_TILE;
        A->tune_us_mring = us[6];
        A->tune_us_mring_nt = us[7];
        A->mring.nt = A->mring.d_plan ? better(us[7], us[6]) : mring_nt0;
        const double best_mring = A->mring.nt ? us[7] : us[6];
        if (better(best_mring, A->auto_kernel == MI_KERNEL_RING ? best_ring : (A->auto_kernel == MI_KERNEL_TILE ? best_tile : best_stream)))
            A->auto_kernel = MI_KERNEL_MRING;
        if (A->blocked) { // the blocked copy against the best CSR kernel
            A->kernel = MI_KERNEL_BCSR4;
            for (int w = 0; w < 3; w++)
                if (launch_spmv(A, tx, ty, nullptr) != MI_OK) break;
            TRY_OR_CLEAN(hipEventRecord(e0, nullptr));
            for (int w = 0; w < 6; w++)
                if (launch_spmv(A, tx, ty, nullptr) != MI_OK) break;
            TRY_OR_CLEAN(hipEventRecord(e1, nullptr));
            TRY_OR_CLEAN(hipEventSynchronize(e1));
            float ms = 0.f;
            TRY_OR_CLEAN(hipEventElapsedTime(&ms, e0, e1));
            A->tune_us_bcsr = ms * 1e3 / 6;
            A->kernel = MI_KERNEL_AUTO;
            const double best_csr = A->auto_kernel == MI_KERNEL_RING ? best_ring : (A->auto_kernel == MI_KERNEL_TILE ? best_tile : (A->auto_kernel == MI_KERNEL_MRING ? best_mring : best_stream));
            if (better(A->tune_us_bcsr, best_csr)) A->auto_kernel = MI_KERNEL_BCSR4;
        }
        // Large ring-served matrices: blocks ending on multiples of 64 rows (the default plan) against unaligned blocks — which
        // is faster depends on the box (ring_plan.hpp), so both are built and timed; the loser is released.
        // (a rank's combined piece of the fused multi-GPU step included: timed here without the exchange, as a plain product)
        if (A->auto_kernel == MI_KERNEL_RING && A->ring.cfg.id == 4 && nnz >= 20000000 && !getenv("MI355_RING_ROW_ALIGN")) {
            RingPlanHost alt;
            build_ring_plan(kRingConfigs[3], n, ptrow, row_min.data(), row_max.data(), alt, ghost_lo, ghost_hi, 1);
            RingTable T2;
            if (nnz > 0 && 1.0 - (double)alt.bad_nnz / (double)nnz >= 0.90 &&
                fill_ring_table(T2, alt, n, ptrow, indcol, nnz, ghost_lo < ghost_hi) == MI_OK) {
                T2.nt = A->ring.nt;
                double us64 = 0.0, us1 = 0.0;
                A->kernel = MI_KERNEL_RING;
                int rct = time_handle(A, 3, 8, &us64);
                std::swap(A->ring, T2);
                if (rct == MI_OK) rct = time_handle(A, 3, 8, &us1);
                A->kernel = MI_KERNEL_AUTO;
                A->tune_us_ring_aligned = us64;
                A->tune_us_ring_unaligned = us1;
                if (rct != MI_OK || !(us1 < 0.98 * us64)) std::swap(A->ring, T2); // keep the default unless the other is clearly faster
                free_ring_table(T2);
            }
        }
        if (A->auto_kernel == MI_KERNEL_MRING && nnz >= 20000000 && !getenv("MI355_RING_ROW_ALIGN")) { // the same for the multi-window ring
            MringTable keep = A->mring;
            A->mring = MringTable();
            if (build_mring(A, indcol, 1) == MI_OK && A->mring.d_plan && A->mring.ok_fraction >= 0.90) {
                A->mring.nt = keep.nt;
                double us64 = 0.0, us1 = 0.0;
                A->kernel = MI_KERNEL_MRING;
                int rct = time_handle(A, 3, 8, &us1);
                std::swap(A->mring, keep);
                if (rct == MI_OK) rct = time_handle(A, 3, 8, &us64);
                A->kernel = MI_KERNEL_AUTO;
                A->tune_us_ring_aligned = us64;
                A->tune_us_ring_unaligned = us1;
                if (rct == MI_OK && us1 < 0.98 * us64) std::swap(A->mring, keep); // the unaligned plan is clearly faster here
            } else {
                std::swap(A->mring, keep);
            }
            MringTable loser = keep; // release the plan not kept
            keep = A->mring;
            A->mring = loser;
            free_mring(A);
            A->mring = keep;
        }
    }
#undef TRY_OR_CLEAN
    *out = A;
    return MI_OK;
}

// mean launch time of A's current choice over `timed` launches after `warm` (x = 0: timing does not depend on values)
static int time_handle(mi_csr_t A, int warm, int timed, double* us)
{
    struct Scratch2 {
        double *tx = nullptr, *ty = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Scratch2()
        {
            dfree(tx);
            dfree(ty);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } t;
    const size_t nx = (size_t)(A->ncols > 0 ? A->ncols : 1), ny = (size_t)(A->n_out > 0 ? A->n_out : 1);
    HIP_TRY(hipMalloc(&t.tx, sizeof(double) * nx));
    HIP_TRY(hipMalloc(&t.ty, sizeof(double) * ny));
    HIP_TRY(hipMemset(t.tx, 0, sizeof(double) * nx));
    HIP_TRY(hipEventCreate(&t.e0));
    HIP_TRY(hipEventCreate(&t.e1));
    int rc;
    for (int w = 0; w < warm; w++)
        if ((rc = launch_spmv(A, t.tx, t.ty, nullptr))) return rc;
    HIP_TRY(hipEventRecord(t.e0, nullptr));
    for (int w = 0; w < timed; w++)
        if ((rc = launch_spmv(A, t.tx, t.ty, nullptr))) return rc;
    HIP_TRY(hipEventRecord(t.e1, nullptr));
    HIP_TRY(hipEventSynchronize(t.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t.e0, t.e1));
    *us = ms * 1e3 / timed;
    return MI_OK;
}

static void release_natural_arrays(mi_csr_t A)
{
    dfree(A->d_ptrow);
    dfree(A->d_indcol);
    dfree(A->d_coef);
    A->d_ptrow = A->d_indcol = nullptr;
    A->d_coef = nullptr;
    for (auto& kv : A->tables) dfree(kv.second.d_blk);
    A->tables.clear();
    dfree(A->ring.d_plan);
    dfree(A->ring.d_ok);
    dfree(A->ring.d_rng);
    dfree(A->ring.d_run_halo);
    dfree(A->ring.d_slots);
    A->ring = RingTable();
    free_tile(A);
    free_mring(A);
    mi_bcsr4_destroy(A->blocked);
    A->blocked = nullptr;
}

// Locality reordering at create time (reorder.hpp).  Tried when the matrix is square, not row-mapped, large enough
// to matter, NOT already served by the ring kernel, and its nonzeros lie far from the diagonal for its size (an
// unstructured node numbering); kept when the reordered twin — x gather included — measures faster.
// MI355_REORDER=0 never, =1 always try and keep (tests), unset: as described.
static int maybe_reorder(mi_csr_t A, const int* ptrow, const int* indcol, const double* coef)
{
    const char* env = getenv("MI355_REORDER");
    const bool force = env && !strcmp(env, "1");
    if (env && !strcmp(env, "0")) return MI_OK;
    const int n = A->n;
    if (A->mapped || n != A->ncols || n < 8 || A->nnz == 0) return MI_OK;
    if (!force && (n < 100000 || resolve_kernel(A) == MI_KERNEL_RING)) return MI_OK;
    const int block = csr_has_block4_pattern(n, ptrow, indcol) ? 4 : 1;
    const double nn = (double)n / block;
    const double spread = mean_column_distance(n, ptrow, indcol, block);
    A->spread_before = spread;
    // a mesh of nn nodes in d >= 2 dimensions cannot be numbered with a mean distance much below nn^(1 - 1/d);
    // far above the 3-D figure means the numbering, not the mesh, spreads the columns
    if (!force && spread < 2.0 * std::pow(nn, 2.0 / 3.0)) return MI_OK;
    Reorder R;
    rcm_reorder(n, ptrow, indcol, block, R);
    A->reorder_block = block;
    A->spread_after = R.spread_after;
    if (!force && R.spread_after > 0.5 * spread) return MI_OK; // nothing gained
    std::vector<int> p2, c2, src_start;
    std::vector<double> v2;
    permute_csr(n, ptrow, indcol, coef, R, p2, c2, v2, src_start);
    mi_csr_t inner = nullptr;
    int rc = csr_create_impl(n, n, p2.data(), c2.data(), v2.data(), R.iperm.data(), &inner);
    if (rc) return rc;
    A->inner = inner; // from here on launch_spmv(A) goes through the twin; destroy releases it
    hipError_t e;
    if ((e = hipMalloc(&A->d_iperm, sizeof(int) * (size_t)n)) != hipSuccess ||
        (e = hipMalloc(&A->d_src_start, sizeof(int) * (size_t)n)) != hipSuccess ||
        (e = hipMalloc(&A->d_xp, sizeof(double) * (size_t)n)) != hipSuccess ||
        (e = hipMemcpy(A->d_iperm, R.iperm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(A->d_src_start, src_start.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP, std::string("reorder upload: ") + hipGetErrorString(e));
    const char* at = getenv("MI355_SPMV_AUTOTUNE");
    bool keep = true;
    if (!force && !(at && !strcmp(at, "0"))) { // measure: natural choice against the twin (gather included)
        const int timed = A->nnz < 40000000 ? 12 : 6;
        A->inner = nullptr;
        rc = time_handle(A, 3, timed, &A->us_natural);
        A->inner = inner;
        if (rc) return rc;
        if ((rc = time_handle(A, 3, timed, &A->us_reordered))) return rc;
        keep = A->us_reordered < 0.97 * A->us_natural;
    }
    if (!keep) {
        mi_csr_destroy(A->inner);
        A->inner = nullptr;
        dfree(A->d_iperm);
        dfree(A->d_src_start);
        dfree(A->d_xp);
        A->d_iperm = A->d_src_start = nullptr;
        A->d_xp = nullptr;
        return MI_OK;
    }
    release_natural_arrays(A);
    return MI_OK;
}

extern "C" int mi_csr_create(int n, int ncols, const int* ptrow, const int* indcol, const double* coef, mi_csr_t* out)
{
    int rc = csr_create_impl(n, ncols, ptrow, indcol, coef, nullptr, out);
    if (rc) return rc;
    if ((rc = maybe_reorder(*out, ptrow, indcol, coef))) {
        mi_csr_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

// host-only: the relabelling maybe_reorder would compute (reverse Cuthill-McKee on the node graph), for CPU tests
extern "C" int mi_reorder_probe(int n, const int* ptrow, const int* indcol, int* block, int* perm, double* spread_before,
                                double* spread_after)
{
    CHECK_ARG(n >= 0 && ptrow && (ptrow[n] == 0 || indcol), "bad argument");
    for (int i = 0; i < n; i++) {
        CHECK_ARG(ptrow[i] <= ptrow[i + 1], "ptrow must be non-decreasing");
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) CHECK_ARG(indcol[k] >= 0 && indcol[k] < n, "column index outside [0, n)");
    }
    const int b = csr_has_block4_pattern(n, ptrow, indcol) ? 4 : 1;
    Reorder R;
    rcm_reorder(n, ptrow, indcol, b, R);
    if (block) *block = b;
    if (perm)
        for (int i = 0; i < n; i++) perm[i] = R.perm[i];
    if (spread_before) *spread_before = R.spread_before;
    if (spread_after) *spread_after = R.spread_after;
    return MI_OK;
}

extern "C" int mi_csr_reorder_info(mi_csr_t A, int* reordered, int* block, double* spread_before, double* spread_after,
                                   double* us_natural, double* us_reordered)
{
    CHECK_ARG(A, "null handle");
    if (reordered) *reordered = A->inner ? 1 : 0;
    if (block) *block = A->reorder_block;
    if (spread_before) *spread_before = A->spread_before;
    if (spread_after) *spread_after = A->spread_after;
    if (us_natural) *us_natural = A->us_natural;
    if (us_reordered) *us_reordered = A->us_reordered;
    return MI_OK;
}

extern "C" int mi_csr_create_mapped(int n, int ncols, const int* ptrow, const int* indcol, const double* coef,
                                    const int* rowmap, mi_csr_t* out)
{
    return csr_create_impl(n, ncols, ptrow, indcol, coef, rowmap, out);
}

extern "C" int mi_csr_destroy(mi_csr_t A)
{
    if (!A) return MI_OK;
    dfree(A->d_ptrow);
    dfree(A->d_indcol);
    dfree(A->d_coef);
    dfree(A->d_rowmap);
    dfree(A->d_x);
    dfree(A->d_y);
    for (double* p : A->d_pow) dfree(p);
    for (auto& kv : A->tables) {
        dfree(kv.second.d_blk);
    }
    dfree(A->ring.d_plan);
    dfree(A->ring.d_ok);
    dfree(A->ring.d_rng);
    dfree(A->ring.d_run_halo);
    dfree(A->ring.d_slots);
    free_tile(A);
    free_mring(A);
    mi_bcsr4_destroy(A->blocked);
    mi_csr_destroy(A->inner);
    dfree(A->d_iperm);
    dfree(A->d_src_start);
    dfree(A->d_xp);
    dfree(A->d_vtmp);
    for (double* p : A->d_pp) dfree(p);
    delete A;
    return MI_OK;
}

// Blocked copy's values from the CSR values already on the device: lane q of block row bi copies its
// row's four coefficients of every block (32 B per lane and block; setup-time traffic).
__global__ __launch_bounds__(kWG) void bcsr4_values_from_csr_kernel(int nbrows, const int* __restrict__ csr_ptrow,
                                                                     const double* __restrict__ csr_coef,
                                                                     const int* __restrict__ bptr, double* __restrict__ bval)
{
    const int g = blockIdx.x * kWG + threadIdx.x;
    const int bi = g >> 2, q = g & 3;
    if (bi >= nbrows) return;
    const double* src = csr_coef + csr_ptrow[4 * bi + q];
    const int b0 = bptr[bi], b1 = bptr[bi + 1];
    for (int blk = b0; blk < b1; blk++) {
        double* dst = bval + 16 * (size_t)blk + 4 * q;
        const double* sp = src + 4 * (size_t)(blk - b0);
        dst[0] = sp[0]; dst[1] = sp[1]; dst[2] = sp[2]; dst[3] = sp[3];
    }
}

static int refresh_blocked_values(mi_csr_t A, hipStream_t s)
{
    if (!A->blocked || A->blocked->nbrows == 0) return MI_OK;
    const long long threads = 4LL * A->blocked->nbrows;
    hipLaunchKernelGGL(bcsr4_values_from_csr_kernel, dim3((unsigned)((threads + kWG - 1) / kWG)), dim3(kWG), 0, s,
                       A->blocked->nbrows, A->d_ptrow, A->d_coef, A->blocked->d_ptrow, A->blocked->d_coef);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// New coefficients for an unchanged sparsity pattern (what a Newton loop does to its Jacobian every
// iteration, src/solve_newton.c:1245-1247): only the value array is replaced.  Row-block tables, the
// ring plan, the 16-bit column stream and the kernel choice depend on the pattern alone and are kept;
// the blocked copy's values are regenerated on the device.
// values of a reordered twin from the caller's (original order) values: new row r' copies its segment
__global__ __launch_bounds__(kWG) void permute_values_kernel(int n, const int* __restrict__ new_ptrow, const int* __restrict__ src_start,
                                                              const double* __restrict__ src, double* __restrict__ dst)
{
    const int r = blockIdx.x * kWG + threadIdx.x;
    if (r >= n) return;
    const int b = new_ptrow[r], len = new_ptrow[r + 1] - b, a = src_start[r];
    for (int k = 0; k < len; k++) dst[b + k] = src[a + k];
}

extern "C" int mi_csr_update_values_dev(mi_csr_t A, const double* d_coef, mi_stream_t s_)
{
    CHECK_ARG(A, "null handle");
    if (A->nnz == 0) return MI_OK;
    CHECK_ARG(d_coef, "null coef");
    hipStream_t s = (hipStream_t)s_;
    if (A->inner) {
        mi_csr_t I = A->inner;
        hipLaunchKernelGGL(permute_values_kernel, dim3((A->n + kWG - 1) / kWG), dim3(kWG), 0, s, A->n, I->d_ptrow, A->d_src_start, d_coef, I->d_coef);
        HIP_TRY(hipGetLastError());
        return refresh_blocked_values(I, s);
    }
    HIP_TRY(hipMemcpyAsync(A->d_coef, d_coef, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToDevice, s));
    return refresh_blocked_values(A, s);
}

extern "C" int mi_csr_update_values(mi_csr_t A, const double* coef)
{
    CHECK_ARG(A, "null handle");
    if (A->nnz == 0) return MI_OK;
    CHECK_ARG(coef, "null coef");
    if (A->inner) {
        if (!A->d_vtmp) HIP_TRY(hipMalloc(&A->d_vtmp, sizeof(double) * (size_t)A->nnz));
        HIP_TRY(hipMemcpy(A->d_vtmp, coef, sizeof(double) * (size_t)A->nnz, hipMemcpyHostToDevice));
        int rc = mi_csr_update_values_dev(A, A->d_vtmp, nullptr);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(nullptr));
        return MI_OK;
    }
    HIP_TRY(hipMemcpy(A->d_coef, coef, sizeof(double) * (size_t)A->nnz, hipMemcpyHostToDevice));
    int rc = refresh_blocked_values(A, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    return MI_OK;
}

extern "C" int mi_csr_dims(mi_csr_t A, int* n, int* ncols, long long* nnz)
{
    CHECK_ARG(A, "null handle");
    if (n) *n = A->n;
    if (ncols) *ncols = A->ncols;
    if (nnz) *nnz = A->nnz;
    return MI_OK;
}

static int resolve_kernel(const mi_csr_s* A)
{
    int k = A->kernel != MI_KERNEL_AUTO ? A->kernel : A->auto_kernel;
    if (k == MI_KERNEL_RING && !A->ring.d_plan) k = MI_KERNEL_STREAM; // empty matrix: nothing to plan
    if (k == MI_KERNEL_BCSR4 && !A->blocked) k = MI_KERNEL_STREAM;
    if (k == MI_KERNEL_TILE && !A->tile.d_desc) k = MI_KERNEL_STREAM;
    if (k == MI_KERNEL_MRING && !A->mring.d_plan) k = MI_KERNEL_STREAM;
    return k;
}

extern "C" int mi_csr_tune_info(mi_csr_t A, double* us_ring, double* us_stream)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    if (us_ring) *us_ring = A->ring.nt ? A->tune_us_ring_nt : A->tune_us_ring;
    if (us_stream) *us_stream = A->stream_nt ? A->tune_us_stream_nt : A->tune_us_stream;
    return MI_OK;
}

extern "C" int mi_csr_tune_detail(mi_csr_t A, double us[5], int* ring_nt, int* stream_nt)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    if (us) {
        us[0] = A->tune_us_ring;
        us[1] = A->tune_us_ring_nt;
        us[2] = A->tune_us_stream;
        us[3] = A->tune_us_stream_nt;
        us[4] = A->tune_us_bcsr;
    }
    if (ring_nt) *ring_nt = A->ring.nt ? 1 : 0;
    if (stream_nt) *stream_nt = A->stream_nt ? 1 : 0;
    return MI_OK;
}

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

// plain read sweep, 16 bytes per lane and step, grid-stride: what this very GPU streams from HBM when nothing else is asked of it
template <bool NT>
__global__ __launch_bounds__(256) void stream_read_kernel(const double2* __restrict__ p, size_t n16, double* __restrict__ sink)
{
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        double2 v;
        if (NT) {
            v.x = __builtin_nontemporal_load(&p[i].x);
            v.y = __builtin_nontemporal_load(&p[i].y);
        } else v = p[i];
        s += v.x + v.y;
    }
    if (s == 123.456) sink[0] = s; // never true: keeps the loads alive
}

extern "C" int mi_debug_stream_read(long long bytes, int launches, double* us_per_launch)
{
    CHECK_ARG(bytes >= (1 << 20) && launches >= 1 && us_per_launch, "bad argument");
    int rc = need_device();
    if (rc) return rc;
    struct Scratch {
        void* buf = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Scratch()
        {
            dfree(buf);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } t;
    HIP_TRY(hipMalloc(&t.buf, (size_t)bytes + 256));
    HIP_TRY(hipMemset(t.buf, 1, (size_t)bytes + 256));
    HIP_TRY(hipEventCreate(&t.e0));
    HIP_TRY(hipEventCreate(&t.e1));
    double* sink = reinterpret_cast<double*>((char*)t.buf + ((size_t)bytes / 16) * 16);
    auto launch = [&]() {
        hipLaunchKernelGGL(stream_read_kernel<true>, dim3(2048), dim3(256), 0, nullptr, (const double2*)t.buf, (size_t)bytes / 16, sink);
    };
    for (int i = 0; i < 3; i++) launch();
    HIP_TRY(hipEventRecord(t.e0, nullptr));
    for (int i = 0; i < launches; i++) launch();
    HIP_TRY(hipEventRecord(t.e1, nullptr));
    HIP_TRY(hipEventSynchronize(t.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t.e0, t.e1));
    *us_per_launch = ms * 1e3 / launches;
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

// host-only: build the ring plan and the 16-bit column stream for one configuration exactly as
// mi_csr_create would, and check their invariants (every block of a served run keeps its columns
// inside one window of at most RING entries, distinct columns of a block get distinct slots, slots
// are < RING, rows/nonzeros are covered once and in order).  For CPU-side tests of the planner.
extern "C" int mi_ring_plan_probe(int n, const int* ptrow, const int* indcol, int config_id, int* nblk, int* runs,
                                  int* runs_not_ringable, double* nnz_fraction_ringable, int* max_slot)
{
    CHECK_ARG(n >= 0 && ptrow && config_id >= 1 && config_id <= kNumRingConfigs, "bad argument");
    const long long nnz = ptrow[n];
    CHECK_ARG(nnz == 0 || indcol, "indcol is null");
    std::vector<int> row_min((size_t)n), row_max((size_t)n);
    for (int i = 0; i < n; i++) {
        int lo = 0x7fffffff, hi = -1;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) {
            lo = std::min(lo, indcol[k]);
            hi = std::max(hi, indcol[k]);
        }
        row_min[i] = lo;
        row_max[i] = hi;
    }
    RingPlanHost P;
    build_ring_plan(kRingConfigs[config_id - 1], n, ptrow, row_min.data(), row_max.data(), P);
    std::vector<unsigned short> slots;
    if (P.nblk > 0) build_ring_slots(P, indcol, slots);
    const RingConfig& c = P.cfg;
    const int T = c.threads, per = c.nnzb / T;
    int next_row = 0, mslot = -1;
    long long next_nz = 0;
    // replay of the window as the kernel keeps it: content[s] = the column whose x value slot s holds
    std::vector<int> content((size_t)c.ring, -1);
    int cur_run = -1;
    // the runs: every block in exactly one, none longer than the kernel's LDS plan, dealt out by weight (ring_plan.hpp)
    std::vector<int> run_of((size_t)P.nblk, -1);
    long long wmax = 0, wsum = 0;
    for (int g = 0; g < P.wgs; g++) {
        const int b0 = P.run_rng[2 * g], b1 = P.run_rng[2 * g + 1];
        if (b0 < 0 || b1 < b0 || b1 > P.nblk || b1 - b0 > kRingMaxB) return fail(MI_ERR_STATE, "run range out of bounds or longer than the kernel's plan");
        long long w = 0;
        for (int b = b0; b < b1; b++) {
            if (run_of[b] >= 0) return fail(MI_ERR_STATE, "a block belongs to two runs");
            run_of[b] = g;
            w += !P.run_ok[g] || P.plan[(size_t)8 * b + 7] == 2 ? kRingPlainWeight : 1;
        }
        wmax = std::max(wmax, w);
        wsum += w;
    }
    for (int b = 0; b < P.nblk; b++)
        if (run_of[b] < 0) return fail(MI_ERR_STATE, "a block belongs to no run");
    if (P.wgs > 0 && wmax > 2 * (wsum / P.wgs) + 4 * kRingPlainWeight) return fail(MI_ERR_STATE, "one run carries more than twice the mean weight");
    for (int b = 0; b < P.nblk; b++) {
        const int* Q = &P.plan[(size_t)8 * b];
        if (Q[0] != next_row || Q[1] != next_nz) return fail(MI_ERR_STATE, "plan does not cover rows / nonzeros in order");
        const int brows = Q[7] == 2 ? Q[4] : Q[2]; // a PLAIN block keeps its row count out of the loop's sight
        next_row += brows;
        next_nz += Q[3];
        if (Q[3] != ptrow[Q[0] + brows] - ptrow[Q[0]]) return fail(MI_ERR_STATE, "block nonzero count disagrees with ptrow");
        const int run = run_of[b];
        if (P.lean && P.run_ok[run] && brows > T) return fail(MI_ERR_STATE, "a LEAN plan holds a block of more than T rows");
        if (!P.run_ok[run] || Q[3] == 0) {
            if (Q[7] == 2 || (Q[7] && Q[3] == 0)) return fail(MI_ERR_STATE, "flags of a block outside the ring loop");
            continue;
        }
        if (Q[7] == 2) { // computed behind the loop: the loop must see an empty block
            if (Q[2] != 0 || Q[5] != 0) return fail(MI_ERR_STATE, "a PLAIN block is visible to the ring loop");
            if (P.run_ok[run] != 3) return fail(MI_ERR_STATE, "a run with a PLAIN block does not tell the kernel to look behind its loop");
            continue;
        }
        if (Q[7] != 1 || Q[3] > c.nnzb || Q[2] > 2 * T) return fail(MI_ERR_STATE, "a served run holds a block the kernel cannot take");
        int cmin = 0x7fffffff, cmax = -1;
        for (long long k = Q[1]; k < (long long)Q[1] + Q[3]; k++) {
            cmin = std::min(cmin, indcol[k]);
            cmax = std::max(cmax, indcol[k]);
        }
        if (cmax - cmin + 1 > c.ring) return fail(MI_ERR_STATE, "block window wider than the ring");
        if (cmin - Q[6] < 0 || cmax - Q[6] >= 2 * c.ring) return fail(MI_ERR_STATE, "ring base out of range for the block's columns");
        if (Q[4] + Q[5] < cmax + 1) return fail(MI_ERR_STATE, "window does not reach the block's last column");
        if (P.lean && Q[5] > T && b != P.run_rng[2 * run]) return fail(MI_ERR_STATE, "a LEAN plan brings more than T new columns into a block inside a run");
        if (run != cur_run) { // a new workgroup: nothing in its ring yet
            std::fill(content.begin(), content.end(), -1);
            cur_run = run;
        }
        for (int col = Q[4]; col < Q[4] + Q[5]; col++) { // the columns this block brings in
            int sl = col - Q[6];
            if (sl >= c.ring) sl -= c.ring;
            if (sl < 0 || sl >= c.ring) return fail(MI_ERR_STATE, "a new column falls outside the ring");
            content[sl] = col;
        }
        for (int k = 0; k < Q[3]; k++)
            if (content[(indcol[Q[1] + k] - Q[6]) >= c.ring ? indcol[Q[1] + k] - Q[6] - c.ring : indcol[Q[1] + k] - Q[6]] != indcol[Q[1] + k])
                return fail(MI_ERR_STATE, "a nonzero's column is not in the window when its block runs");
        for (int k = 0; k < Q[3]; k++) { // slot of nonzero k as the kernel's thread (k % T), element k / T reads it
            const int slot = slots[(size_t)b * c.nnzb + (size_t)(k % T) * per + k / T];
            int want = indcol[Q[1] + k] - Q[6];
            if (want >= c.ring) want -= c.ring;
            if (slot != want || slot < 0 || slot >= c.ring) return fail(MI_ERR_STATE, "16-bit slot disagrees with the column");
            mslot = std::max(mslot, slot);
        }
    }
    if (next_row != n || next_nz != nnz) return fail(MI_ERR_STATE, "plan does not cover the matrix");
    if (nblk) *nblk = P.nblk;
    if (runs) *runs = P.wgs;
    if (runs_not_ringable) *runs_not_ringable = P.bad_runs;
    if (nnz_fraction_ringable) *nnz_fraction_ringable = nnz ? 1.0 - (double)P.bad_nnz / (double)nnz : 0.0;
    if (max_slot) *max_slot = mslot;
    return MI_OK;
}

extern "C" int mi_csr_block4_structure(int n, const int* ptrow, const int* indcol, int* is_blocked, long long* nblocks)
{
    CHECK_ARG(n >= 0 && ptrow && is_blocked, "bad argument");
    CHECK_ARG(ptrow[n] == 0 || indcol, "indcol is null");
    // values are irrelevant to the structure test: hand the converter a dummy array of the right length
    std::vector<double> dummy((size_t)ptrow[n], 0.0);
    std::vector<int> bptr, bcol;
    std::vector<double> bval;
    const bool ok = csr_to_bcsr4_exact(n, ptrow, indcol, dummy.data(), bptr, bcol, bval);
    *is_blocked = ok ? 1 : 0;
    if (nblocks) *nblocks = ok ? (long long)bcol.size() : 0;
    return MI_OK;
}

extern "C" int mi_csr_set_nontemporal(mi_csr_t A, int ring_nt, int stream_nt)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    if (ring_nt >= 0) A->ring.nt = ring_nt != 0 && A->ring.d_slots;
    if (stream_nt >= 0) A->stream_nt = stream_nt != 0;
    return MI_OK;
}

extern "C" int mi_csr_ring_info(mi_csr_t A, int* config_id, int* runs, int* runs_not_ringable, double* nnz_fraction_ringable)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    if (config_id) *config_id = A->ring.cfg.id;
    if (runs) *runs = A->ring.wgs;
    if (runs_not_ringable) *runs_not_ringable = A->ring.bad_runs;
    if (nnz_fraction_ringable) *nnz_fraction_ringable = A->ring.ok_fraction;
    return MI_OK;
}

extern "C" int mi_csr_tile_info(mi_csr_t A, int* built, int* nblk, double* unique_per_nnz, double us[2], int* nt)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    if (built) *built = A->tile.d_desc != nullptr;
    if (nblk) *nblk = A->tile.nblk;
    if (unique_per_nnz) *unique_per_nnz = A->tile.unique_per_nnz;
    if (us) {
        us[0] = A->tune_us_tile;
        us[1] = A->tune_us_tile_nt;
    }
    if (nt) *nt = A->tile.nt;
    return MI_OK;
}

extern "C" int mi_csr_mring_info(mi_csr_t A, int* built, int* runs, int* runs_not_served, double* nnz_fraction_served, double us[2], int* nt)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    if (built) *built = A->mring.d_plan != nullptr;
    if (runs) *runs = A->mring.nruns;
    if (runs_not_served) *runs_not_served = A->mring.bad_runs;
    if (nnz_fraction_served) *nnz_fraction_served = A->mring.ok_fraction;
    if (us) {
        us[0] = A->tune_us_mring;
        us[1] = A->tune_us_mring_nt;
    }
    if (nt) *nt = A->mring.nt;
    return MI_OK;
}

extern "C" int mi_mring_plan_probe(int n, const int* ptrow, const int* indcol, int* nblk, int* runs, int* runs_not_served,
                                   double* nnz_fraction_served, long long* window_restarts)
{
    CHECK_ARG(n >= 0 && ptrow && ptrow[0] == 0, "bad matrix");
    CHECK_ARG(ptrow[n] == 0 || indcol, "indcol is null");
    MringPlanHost P;
    build_mring_plan(n, ptrow, indcol, P);
    if (const char* bad = check_mring_plan(P, n, ptrow, indcol)) return fail(MI_ERR_STATE, std::string("mring plan: ") + bad);
    if (nblk) *nblk = P.nblk;
    if (runs) *runs = P.nruns;
    if (runs_not_served) *runs_not_served = P.bad_runs;
    if (nnz_fraction_served) *nnz_fraction_served = ptrow[n] ? 1.0 - (double)P.bad_nnz / (double)ptrow[n] : 0.0;
    if (window_restarts) *window_restarts = P.restarts;
    return MI_OK;
}

extern "C" int mi_tile_plan_probe(int n, const int* ptrow, const int* indcol, int threads, int* nblk, long long* distinct_total,
                                  int* max_distinct, long long* nnz_listed)
{
    CHECK_ARG(n >= 0 && ptrow && ptrow[0] == 0, "bad matrix");
    CHECK_ARG(ptrow[n] == 0 || indcol, "indcol is null");
    TilePlanHost P;
    build_tile_plan(n, ptrow, indcol, P, kTileNnzb, threads);
    if (const char* bad = check_tile_plan(P, n, ptrow, indcol)) return fail(MI_ERR_STATE, std::string("tile plan: ") + bad);
    if (nblk) *nblk = P.nblk;
    if (distinct_total) *distinct_total = P.nblk > 0 ? (long long)P.ulist.size() - kTileThreads : 0;
    if (max_distinct) *max_distinct = P.max_unique;
    if (nnz_listed) *nnz_listed = P.listed;
    return MI_OK;
}

extern "C" int mi_ring_plan_lean(int n, const int* ptrow, const int* indcol, int config_id, int* lean)
{
    CHECK_ARG(n >= 0 && ptrow && lean && config_id >= 1 && config_id <= kNumRingConfigs, "bad argument");
    std::vector<int> row_min((size_t)n), row_max((size_t)n);
    for (int i = 0; i < n; i++) {
        int lo = 0x7fffffff, hi = -1;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) {
            lo = std::min(lo, indcol[k]);
            hi = std::max(hi, indcol[k]);
        }
        row_min[i] = lo;
        row_max[i] = hi;
    }
    RingPlanHost P;
    build_ring_plan(kRingConfigs[config_id - 1], n, ptrow, row_min.data(), row_max.data(), P);
    *lean = P.lean && P.cfg.id == 4 && P.bad_runs == 0;
    return MI_OK;
}

extern "C" int mi_csr_ring_shape_info(mi_csr_t A, int* blocks, int* lean, int* depth, double* us_aligned, double* us_unaligned)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    if (blocks) *blocks = A->ring.nblk;
    if (lean) *lean = A->ring.lean;
    if (depth) *depth = A->ring.cfg.depth;
    if (us_aligned) *us_aligned = A->tune_us_ring_aligned;
    if (us_unaligned) *us_unaligned = A->tune_us_ring_unaligned;
    return MI_OK;
}

extern "C" int mi_csr_set_kernel(mi_csr_t A, int kernel_id)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    CHECK_ARG(kernel_id >= MI_KERNEL_AUTO && kernel_id <= MI_KERNEL_MRING, "unknown kernel id");
    if (kernel_id == MI_KERNEL_BCSR4 && !A->blocked)
        return fail(MI_ERR_UNSUPPORTED, "MI_KERNEL_BCSR4: this matrix has no exact 4x4 block structure (or is row-mapped)");
    if (kernel_id == MI_KERNEL_TILE) { // the plan is built on first request if mi_csr_create did not keep one
        int rc = need_device();
        if (rc) return rc;
        if ((rc = build_tile(A, nullptr))) return rc;
    }
    if (kernel_id == MI_KERNEL_MRING) {
        int rc = need_device();
        if (rc) return rc;
        if ((rc = build_mring(A, nullptr))) return rc;
    }
    A->kernel = kernel_id;
    return MI_OK;
}

extern "C" int mi_csr_get_kernel(mi_csr_t A, int* kernel_id)
{
    CHECK_ARG(A && kernel_id, "null argument");
    if (A->inner) A = A->inner;
    *kernel_id = resolve_kernel(A);
    return MI_OK;
}

extern "C" const char* mi_csr_kernel_name(mi_csr_t A)
{
    if (!A) return "";
    if (A->inner) A = A->inner;
    switch (resolve_kernel(A)) {
    case MI_KERNEL_STREAM: return A->stream_nt ? "spmv_csr_stream<1024, true>" : "spmv_csr_stream<1024, false>";
    case MI_KERNEL_RING: { // the name rocprofv3 prints for the instantiation launch_ring picks
        static thread_local char nm[112];
        const RingConfig& c = A->ring.cfg;
        snprintf(nm, sizeof nm, "spmv_csr_ring<%d, %d, %d, %d, %d, %s, %s, %s, false, %s>", c.threads, c.nnzb, c.ring, c.depth, kRingMaxB,
                 A->d_rowmap ? "true" : "false", A->ring.nt ? "true" : "false", A->ring.skew ? "true" : "false",
                 A->ring.lean && c.threads == 256 && c.depth != 3 ? "true" : "false");
        return nm;
    }
    case MI_KERNEL_ROWPAR: return "spmv_csr_rowpar";
    case MI_KERNEL_BCSR4: return A->blocked && A->blocked->use_tile && A->blocked->d_tl_ptr ? "spmv_bcsr4_tile<2>" : "spmv_bcsr4<2>";
    case MI_KERNEL_MRING: {
        static thread_local char nm[96];
        snprintf(nm, sizeof nm, "spmv_csr_mring<%d, %d, %d, %d, %s, %s, %s>", kMringThreads, kMringNnzb, A->mring.depth, kMringMaxB,
                 A->d_rowmap ? "true" : "false", A->mring.nt ? "true" : "false", A->mring.skew ? "true" : "false");
        return nm;
    }
    case MI_KERNEL_TILE: {
        static thread_local char nm[64];
        snprintf(nm, sizeof nm, "spmv_csr_tile<%d, %s, %s>", kTileNnzb, A->tile.nt ? "true" : "false", A->tile.skew ? "true" : "false");
        return nm;
    }
    default: return "";
    }
}

// ---------------------------------------------------------------- SpMV launch
template <int T, int NNZB, int RING, int D, bool MAPPED, bool NT, bool SKEW, bool LEAN>
static void launch_ring3(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm)
{
    if (!MAPPED && comm) { // the fused multi-GPU step: push workgroups in front of the grid (spmv_ring.hpp)
        hipLaunchKernelGGL((spmv_csr_ring<T, NNZB, RING, D, kRingMaxB, MAPPED, NT, SKEW, true, LEAN>), dim3(A->ring.wgs + comm->push_wgs), dim3(T), 0, s,
                           V, reinterpret_cast<const int4*>(A->ring.d_plan), A->ring.d_ok, A->ring.d_slots, d_x, d_y, reinterpret_cast<const int2*>(A->ring.d_rng), A->ring.uniform ? A->ring.bpw : 0, *comm);
        return;
    }
    hipLaunchKernelGGL((spmv_csr_ring<T, NNZB, RING, D, kRingMaxB, MAPPED, NT, SKEW, false, LEAN>), dim3(A->ring.wgs), dim3(T), 0, s, V,
                       reinterpret_cast<const int4*>(A->ring.d_plan), A->ring.d_ok, A->ring.d_slots, d_x, d_y, reinterpret_cast<const int2*>(A->ring.d_rng), A->ring.uniform ? A->ring.bpw : 0, RingComm{});
}

template <int T, int NNZB, int RING, int D, bool MAPPED, bool NT, bool SKEW>
static void launch_ring2(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm)
{
    // the LEAN instantiation exists for the configuration that runs in practice (4: 256 threads) at depths 2 and 4
    if (T == 256 && D != 3 && A->ring.lean) launch_ring3<T, NNZB, RING, D, MAPPED, NT, SKEW, (T == 256 && D != 3)>(A, V, d_x, d_y, s, comm);
    else launch_ring3<T, NNZB, RING, D, MAPPED, NT, SKEW, false>(A, V, d_x, d_y, s, comm);
}

template <int T, int NNZB, int RING, int D, bool MAPPED>
static void launch_ring1(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm)
{
    if (A->ring.nt) {
        if (A->ring.skew) launch_ring2<T, NNZB, RING, D, MAPPED, true, true>(A, V, d_x, d_y, s, comm);
        else launch_ring2<T, NNZB, RING, D, MAPPED, true, false>(A, V, d_x, d_y, s, comm);
    } else {
        if (A->ring.skew) launch_ring2<T, NNZB, RING, D, MAPPED, false, true>(A, V, d_x, d_y, s, comm);
        else launch_ring2<T, NNZB, RING, D, MAPPED, false, false>(A, V, d_x, d_y, s, comm);
    }
}

template <int T, int NNZB, int RING, int D>
static void launch_ring(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm)
{
    static_assert(NNZB <= kRingPadNnz && 2 * T + 1 <= kRingPadRows, "device arrays are padded for the kernel's unclamped loads");
    if (V.rowmap) launch_ring1<T, NNZB, RING, D, true>(A, V, d_x, d_y, s, comm);
    else launch_ring1<T, NNZB, RING, D, false>(A, V, d_x, d_y, s, comm);
}

// dst[idx[i]] = src[i]
__global__ __launch_bounds__(256) void scatter_kernel(int m, const int* __restrict__ idx, const double* __restrict__ src,
                                                      double* __restrict__ dst)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) dst[idx[i]] = src[i];
}

// x into a reordered handle's numbering (whole nodes at a time when nodes were moved and x allows 16-byte accesses)
static int gather_perm(mi_csr_t A, const double* d_x, double* d_xp, hipStream_t s)
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

static int launch_spmv(mi_csr_t A, const double* d_x, double* d_y, hipStream_t s, bool use_map, const RingComm* comm)
{
    if (A->n == 0) return MI_OK;
    if (A->inner) { // reordered: x into the new numbering, then the twin writes y through its row map
        int rc = gather_perm(A, d_x, A->d_xp, s);
        if (rc) return rc;
        return launch_spmv(A->inner, A->d_xp, d_y, s, true);
    }
    const int kid = resolve_kernel(A);
    if (use_map) d_y += A->y_offset;
    CsrView V;
    V.n = A->n;
    V.ncols = A->ncols;
    V.ptrow = A->d_ptrow;
    V.indcol = A->d_indcol;
    V.coef = A->d_coef;
    V.rowmap = use_map ? A->d_rowmap : nullptr;
    V.blk = nullptr;
    V.blk_span = nullptr;
    V.nblk = 0;
    if (kid == MI_KERNEL_BCSR4 && (((uintptr_t)d_x) & 15) == 0) return launch_bcsr4(A->blocked, d_x, d_y, (mi_stream_t)s, use_map);
    if (kid == MI_KERNEL_BCSR4) { // x not 16-byte aligned: the blocked kernel's paired loads cannot be used
        BlockTable* T = nullptr;
        int rc = get_table(A, 1024, &T);
        if (rc) return rc;
        V.blk = T->d_blk;
        V.nblk = T->nblk;
        const int grid = kNXCD * ((T->nblk + kNXCD - 1) / kNXCD);
        hipLaunchKernelGGL((spmv_csr_stream<1024, false>), dim3(grid), dim3(kWG), 0, s, V, d_x, d_y);
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    if (kid == MI_KERNEL_MRING) {
        const MringTable& M = A->mring;
        V.nblk = M.nblk;
        const int4* plan = reinterpret_cast<const int4*>(M.d_plan);
        const int2* rng = reinterpret_cast<const int2*>(M.d_rng);
#define MRING_L(D_, MP_, NT_, SK_) hipLaunchKernelGGL((spmv_csr_mring<kMringThreads, kMringNnzb, D_, kMringMaxB, MP_, NT_, SK_>), dim3(kNXCD * ((M.nruns + kNXCD - 1) / kNXCD)), dim3(kMringThreads), 0, s, V, plan, reinterpret_cast<const int4*>(M.d_first), M.d_ok, M.d_slots, d_x, d_y, rng, M.nruns)
#define MRING_L3(D_, MP_) do { if (M.nt) { if (M.skew) MRING_L(D_, MP_, true, true); else MRING_L(D_, MP_, true, false); } \
                               else { if (M.skew) MRING_L(D_, MP_, false, true); else MRING_L(D_, MP_, false, false); } } while (0)
#define MRING_L2(D_) do { if (V.rowmap) MRING_L3(D_, true); else MRING_L3(D_, false); } while (0)
        int depth = M.depth;
        if (const char* e = getenv("MI355_RING_DEPTH")) depth = atoi(e);
        if (depth == 4) MRING_L2(4);
        else MRING_L2(2);
#undef MRING_L2
#undef MRING_L3
#undef MRING_L
    } else if (kid == MI_KERNEL_TILE) {
        const TileTable& T = A->tile;
        const int grid = kNXCD * ((T.nblk + kNXCD - 1) / kNXCD);
        const int4* desc = reinterpret_cast<const int4*>(T.d_desc);
#define TILE_LAUNCH(NT_, SK_) hipLaunchKernelGGL((spmv_csr_tile<kTileNnzb, NT_, SK_>), dim3(grid), dim3(kTileThreads), 0, s, V, desc, T.nblk, T.d_ulist, T.d_slots, d_x, d_y)
        if (T.nt) { if (T.skew) TILE_LAUNCH(true, true); else TILE_LAUNCH(true, false); }
        else { if (T.skew) TILE_LAUNCH(false, true); else TILE_LAUNCH(false, false); }
#undef TILE_LAUNCH
    } else if (kid == MI_KERNEL_ROWPAR) {
        hipLaunchKernelGGL(spmv_csr_rowpar, dim3((A->n + kWG - 1) / kWG), dim3(kWG), 0, s, V, d_x, d_y);
    } else if (kid == MI_KERNEL_RING) {
        V.nblk = A->ring.nblk;
        switch (A->ring.cfg.id) {
        case 1: launch_ring<512, 2048, 5120, 2>(A, V, d_x, d_y, s, comm); break;
        case 2: launch_ring<512, 4096, 5120, 2>(A, V, d_x, d_y, s, comm); break;
        case 3: launch_ring<512, 4096, 11264, 2>(A, V, d_x, d_y, s, comm); break;
        default: {
            int depth = A->ring.cfg.depth; // chosen at create (long runs: 4 blocks of prefetch); MI355_RING_DEPTH is read per
                                           // launch so that tools/depth_ab.py can compare the depths on one handle
            if (const char* e = getenv("MI355_RING_DEPTH")) depth = atoi(e);
            if (depth == 3) launch_ring<256, 2048, 5120, 3>(A, V, d_x, d_y, s, comm);
            else if (depth == 4) launch_ring<256, 2048, 5120, 4>(A, V, d_x, d_y, s, comm);
            else launch_ring<256, 2048, 5120, 2>(A, V, d_x, d_y, s, comm);
        } break;
        }
    } else {
        BlockTable* T = nullptr;
        int rc = get_table(A, 1024, &T);
        if (rc) return rc;
        V.blk = T->d_blk;
        V.nblk = T->nblk;
        const int grid = kNXCD * ((T->nblk + kNXCD - 1) / kNXCD);
        if (A->stream_nt) hipLaunchKernelGGL((spmv_csr_stream<1024, true>), dim3(grid), dim3(kWG), 0, s, V, d_x, d_y);
        else hipLaunchKernelGGL((spmv_csr_stream<1024, false>), dim3(grid), dim3(kWG), 0, s, V, d_x, d_y);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_spmv_dev(mi_csr_t A, const double* d_x, double* d_y, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(A->n == 0 || (d_x && d_y), "null vector");
    return launch_spmv(A, d_x, d_y, (hipStream_t)s);
}

extern "C" int mi_spmv(mi_csr_t A, const double* x, double* y)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(A->n == 0 || (x && y), "null vector");
    if (A->mapped) return fail(MI_ERR_UNSUPPORTED, "mapped matrices are device-only (use mi_spmv_dev)");
    if (A->n == 0) return MI_OK;
    if (!A->d_x) HIP_TRY(hipMalloc(&A->d_x, sizeof(double) * (size_t)(A->ncols > 0 ? A->ncols : 1)));
    if (!A->d_y) HIP_TRY(hipMalloc(&A->d_y, sizeof(double) * (size_t)A->n));
    HIP_TRY(hipMemcpy(A->d_x, x, sizeof(double) * (size_t)A->ncols, hipMemcpyHostToDevice));
    int rc = launch_spmv(A, A->d_x, A->d_y, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(y, A->d_y, sizeof(double) * (size_t)A->n, hipMemcpyDeviceToHost));
    return MI_OK;
}

// ---------------------------------------------------------------- matrix powers
extern "C" int mi_spmk_dev(mi_csr_t A, int k, const double* d_x, double* const* d_y_out, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    if (k < 1 || k > MI_MAX_POWERS) return fail(MI_ERR_UNSUPPORTED, "k must be in 1..MI_MAX_POWERS");
    CHECK_ARG(A->n == A->ncols, "matrix powers need a square matrix");
    CHECK_ARG(!A->mapped, "matrix powers need an unmapped matrix");
    CHECK_ARG(d_y_out, "null output array");
    if (A->inner && A->n > 0) {
        // the whole chain in the new numbering (each power feeds the next without leaving it), every power scattered
        // to the caller's numbering as it completes
        while ((int)A->d_pp.size() < k) {
            double* p = nullptr;
            HIP_TRY(hipMalloc(&p, sizeof(double) * (size_t)A->n));
            A->d_pp.push_back(p);
        }
        int rc = gather_perm(A, d_x, A->d_xp, (hipStream_t)s);
        if (rc) return rc;
        const double* src = A->d_xp;
        int grid = (A->n + 255) / 256;
        if (grid > 2048) grid = 2048;
        for (int p = 0; p < k; p++) {
            CHECK_ARG(d_y_out[p], "null output vector");
            if ((rc = launch_spmv(A->inner, src, A->d_pp[p], (hipStream_t)s, false))) return rc;
            hipLaunchKernelGGL(scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, A->n, A->d_iperm, A->d_pp[p], d_y_out[p]);
            src = A->d_pp[p];
        }
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    const double* src = d_x;
    for (int p = 0; p < k; p++) {
        CHECK_ARG(A->n == 0 || d_y_out[p], "null output vector");
        int rc = launch_spmv(A, src, d_y_out[p], (hipStream_t)s);
        if (rc) return rc;
        src = d_y_out[p];
    }
    return MI_OK;
}

extern "C" int mi_spmk(mi_csr_t A, int k, const double* x, double* const* y_out)
{
    CHECK_ARG(A, "null handle");
    if (k < 1 || k > MI_MAX_POWERS) return fail(MI_ERR_UNSUPPORTED, "k must be in 1..MI_MAX_POWERS");
    CHECK_ARG(A->n == A->ncols, "matrix powers need a square matrix");
    CHECK_ARG(A->n == 0 || (x && y_out), "null vector");
    if (A->n == 0) return MI_OK;
    if (!A->d_x) HIP_TRY(hipMalloc(&A->d_x, sizeof(double) * (size_t)A->ncols));
    while ((int)A->d_pow.size() < k) {
        double* p = nullptr;
        HIP_TRY(hipMalloc(&p, sizeof(double) * (size_t)A->n));
        A->d_pow.push_back(p);
    }
    HIP_TRY(hipMemcpy(A->d_x, x, sizeof(double) * (size_t)A->ncols, hipMemcpyHostToDevice));
    int rc = mi_spmk_dev(A, k, A->d_x, A->d_pow.data(), nullptr);
    if (rc) return rc;
    for (int p = 0; p < k; p++) {
        CHECK_ARG(y_out[p], "null output vector");
        HIP_TRY(hipMemcpy(y_out[p], A->d_pow[p], sizeof(double) * (size_t)A->n, hipMemcpyDeviceToHost));
    }
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

// host-pointer helper: upload up to three vectors, run f on the device copies, download
struct Scratch {
    std::vector<double*> bufs;
    ~Scratch()
    {
        for (double* p : bufs) dfree(p);
    }
    int up(const double* h, size_t n, double** d)
    {
        *d = nullptr;
        HIP_TRY(hipMalloc(d, sizeof(double) * (n ? n : 1)));
        bufs.push_back(*d);
        if (h && n) HIP_TRY(hipMemcpy(*d, h, sizeof(double) * n, hipMemcpyHostToDevice));
        return MI_OK;
    }
};

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

// ---------------------------------------------------------------- BCSR 4x4
extern "C" int mi_bcsr4_create(int nbrows, int nbcols, const int* ptrow, const int* indcol, const double* coef,
                               mi_bcsr4_t* out)
{
    CHECK_ARG(out, "out is null");
    *out = nullptr;
    CHECK_ARG(nbrows >= 0 && nbcols >= 0 && ptrow && ptrow[0] == 0, "bad argument");
    for (int i = 0; i < nbrows; i++) CHECK_ARG(ptrow[i] <= ptrow[i + 1], "ptrow must be non-decreasing");
    const long long nb = ptrow[nbrows];
    CHECK_ARG(nb == 0 || (indcol && coef), "indcol/coef is null");
    for (long long k = 0; k < nb; k++) CHECK_ARG(indcol[k] >= 0 && indcol[k] < nbcols, "block column outside [0, nbcols)");
    int rc = need_device();
    if (rc) return rc;
    mi_bcsr4_t A = new (std::nothrow) mi_bcsr4_s();
    if (!A) return fail(MI_ERR_ALLOC, "host allocation failed");
    A->nbrows = nbrows;
    A->nbcols = nbcols;
    A->nblocks = nb;
    hipError_t e;
    if ((e = hipGetDevice(&A->device)) != hipSuccess ||
        (e = hipMalloc(&A->d_ptrow, sizeof(int) * ((size_t)nbrows + 1))) != hipSuccess ||
        (e = hipMalloc(&A->d_indcol, sizeof(int) * ((size_t)nb + 1))) != hipSuccess ||
        (e = hipMalloc(&A->d_coef, sizeof(double) * 16 * ((size_t)nb + 1))) != hipSuccess ||
        (e = hipMemcpy(A->d_ptrow, ptrow, sizeof(int) * ((size_t)nbrows + 1), hipMemcpyHostToDevice)) != hipSuccess ||
        (nb && (e = hipMemcpy(A->d_indcol, indcol, sizeof(int) * (size_t)nb, hipMemcpyHostToDevice)) != hipSuccess) ||
        (nb && (e = hipMemcpy(A->d_coef, coef, sizeof(double) * 16 * (size_t)nb, hipMemcpyHostToDevice)) != hipSuccess)) {
        mi_bcsr4_destroy(A);
        return fail(e == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP, std::string("bcsr4 upload: ") + hipGetErrorString(e));
    }
    // The x tile per workgroup (spmv_bcsr4_tile): the distinct block columns of each group of 64 block rows, and every block's
    // position in its group's list.  Built when no group needs more than the tile holds; then both kernels are timed and the
    // faster is kept (MI355_BCSR_TILE=0 never builds it, =1 takes it unmeasured).
    const char* te = getenv("MI355_BCSR_TILE");
    if (nb >= 4096 && !(te && !strcmp(te, "0"))) {
        const int per = kWG / 4, nwg = (nbrows + per - 1) / per;
        std::vector<int> wg_ptr((size_t)nwg + 1, 0);
        std::vector<unsigned> nodes;
        std::vector<unsigned short> slots((size_t)nb + 1, 0);
        std::vector<unsigned> u;
        bool fits = true;
        for (int w = 0; w < nwg && fits; w++) {
            const int b0 = ptrow[(size_t)w * per], b1 = ptrow[std::min<long long>((long long)(w + 1) * per, nbrows)];
            u.assign(indcol + b0, indcol + b1);
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            fits = (int)u.size() <= kBtileNodes;
            for (int k = b0; k < b1 && fits; k++) slots[k] = (unsigned short)(std::lower_bound(u.begin(), u.end(), (unsigned)indcol[k]) - u.begin());
            nodes.insert(nodes.end(), u.begin(), u.end());
            wg_ptr[w + 1] = (int)nodes.size();
        }
        if (fits) {
            nodes.push_back(0);
            if ((e = hipMalloc(&A->d_tl_ptr, sizeof(int) * wg_ptr.size())) != hipSuccess ||
                (e = hipMalloc(&A->d_tl_nodes, sizeof(unsigned) * nodes.size())) != hipSuccess ||
                (e = hipMalloc(&A->d_tl_slots, sizeof(unsigned short) * slots.size())) != hipSuccess ||
                (e = hipMemcpy(A->d_tl_ptr, wg_ptr.data(), sizeof(int) * wg_ptr.size(), hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(A->d_tl_nodes, nodes.data(), sizeof(unsigned) * nodes.size(), hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(A->d_tl_slots, slots.data(), sizeof(unsigned short) * slots.size(), hipMemcpyHostToDevice)) != hipSuccess) {
                mi_bcsr4_destroy(A);
                return fail(e == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP, std::string("bcsr4 tile upload: ") + hipGetErrorString(e));
            }
            const char* at = getenv("MI355_SPMV_AUTOTUNE");
            if (te && !strcmp(te, "1")) A->use_tile = true;
            else if (!(at && !strcmp(at, "0")) && nb >= 100000) { // measure both (x = 0: timing does not depend on the values)
                double *tx = nullptr, *ty = nullptr;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                const size_t nx = 4 * (size_t)std::max(nbcols, 1), ny = 4 * (size_t)std::max(nbrows, 1);
                if (hipMalloc(&tx, sizeof(double) * nx) == hipSuccess && hipMalloc(&ty, sizeof(double) * ny) == hipSuccess &&
                    hipMemset(tx, 0, sizeof(double) * nx) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                    double us[2] = {0, 0};
                    for (int round = 0; round < 2; round++)
                        for (int c = 0; c < 2; c++) {
                            A->use_tile = c == 1;
                            for (int w = 0; w < 3; w++) (void)launch_bcsr4(A, tx, ty, nullptr, false);
                            (void)hipEventRecord(e0, nullptr);
                            for (int w = 0; w < 8; w++) (void)launch_bcsr4(A, tx, ty, nullptr, false);
                            (void)hipEventRecord(e1, nullptr);
                            (void)hipEventSynchronize(e1);
                            float ms = 0.f;
                            (void)hipEventElapsedTime(&ms, e0, e1);
                            const double t = ms * 1e3 / 8;
                            us[c] = us[c] > 0 ? std::min(us[c], t) : t;
                        }
                    A->tune_us_plain = us[0];
                    A->tune_us_tile = us[1];
                    A->use_tile = us[1] > 0 && us[1] < us[0];
                }
                dfree(tx);
                dfree(ty);
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
            }
        }
    }
    *out = A;
    return MI_OK;
}

extern "C" int mi_bcsr4_tile_info(mi_bcsr4_t A, int* built, int* in_use, double* us_plain, double* us_tile)
{
    CHECK_ARG(A, "null handle");
    if (built) *built = A->d_tl_ptr != nullptr;
    if (in_use) *in_use = A->use_tile && A->d_tl_ptr;
    if (us_plain) *us_plain = A->tune_us_plain;
    if (us_tile) *us_tile = A->tune_us_tile;
    return MI_OK;
}

// new block values (16 per block, row-major) for an unchanged block pattern
extern "C" int mi_bcsr4_update_values(mi_bcsr4_t A, const double* coef)
{
    CHECK_ARG(A, "null handle");
    if (A->nblocks == 0) return MI_OK;
    CHECK_ARG(coef, "null coef");
    HIP_TRY(hipMemcpy(A->d_coef, coef, sizeof(double) * 16 * (size_t)A->nblocks, hipMemcpyHostToDevice));
    return MI_OK;
}

extern "C" int mi_bcsr4_update_values_dev(mi_bcsr4_t A, const double* d_coef, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    if (A->nblocks == 0) return MI_OK;
    CHECK_ARG(d_coef, "null coef");
    HIP_TRY(hipMemcpyAsync(A->d_coef, d_coef, sizeof(double) * 16 * (size_t)A->nblocks, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return MI_OK;
}

extern "C" int mi_bcsr4_destroy(mi_bcsr4_t A)
{
    if (!A) return MI_OK;
    dfree(A->d_ptrow);
    dfree(A->d_indcol);
    dfree(A->d_coef);
    dfree(A->d_browmap);
    dfree(A->d_tl_ptr);
    dfree(A->d_tl_nodes);
    dfree(A->d_tl_slots);
    dfree(A->d_x);
    dfree(A->d_y);
    for (double* p : A->d_pow) dfree(p);
    delete A;
    return MI_OK;
}

static int launch_bcsr4(mi_bcsr4_t A, const double* d_x, double* d_y, mi_stream_t s, bool use_map)
{
    CHECK_ARG(A, "null handle");
    if (A->nbrows == 0) return MI_OK;
    CHECK_ARG(d_x && d_y, "null vector");
    CHECK_ARG((((uintptr_t)d_x) & 15) == 0, "x must be 16-byte aligned");
    Bcsr4View V{A->nbrows, A->nbcols, A->d_ptrow, A->d_indcol, A->d_coef, use_map ? A->d_browmap : nullptr};
    const long long threads = 4LL * A->nbrows;
    const int nwg = (int)((threads + kWG - 1) / kWG);
    static const int chunk = getenv("MI355_BCSR_XCD_CHUNK") ? atoi(getenv("MI355_BCSR_XCD_CHUNK")) : 0;
    const int grid = nwg;
    if (A->use_tile && A->d_tl_ptr) {
        Bcsr4Tile Tl{A->d_tl_ptr, A->d_tl_nodes, A->d_tl_slots};
        hipLaunchKernelGGL(spmv_bcsr4_tile<kBcsrDepth>, dim3((unsigned)grid), dim3(kWG), 0, (hipStream_t)s, V, Tl, d_x, d_y, nwg);
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    hipLaunchKernelGGL(spmv_bcsr4<kBcsrDepth>, dim3((unsigned)grid), dim3(kWG), 0, (hipStream_t)s, V, d_x, d_y, chunk, nwg);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_bcsr4_spmv_dev(mi_bcsr4_t A, const double* d_x, double* d_y, mi_stream_t s)
{
    return launch_bcsr4(A, d_x, d_y, s, true);
}

extern "C" int mi_bcsr4_spmv(mi_bcsr4_t A, const double* x, double* y)
{
    CHECK_ARG(A, "null handle");
    if (A->nbrows == 0) return MI_OK;
    CHECK_ARG(x && y, "null vector");
    if (!A->d_x) HIP_TRY(hipMalloc(&A->d_x, sizeof(double) * 4 * (size_t)(A->nbcols > 0 ? A->nbcols : 1)));
    if (!A->d_y) HIP_TRY(hipMalloc(&A->d_y, sizeof(double) * 4 * (size_t)A->nbrows));
    HIP_TRY(hipMemcpy(A->d_x, x, sizeof(double) * 4 * (size_t)A->nbcols, hipMemcpyHostToDevice));
    int rc = mi_bcsr4_spmv_dev(A, A->d_x, A->d_y, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(y, A->d_y, sizeof(double) * 4 * (size_t)A->nbrows, hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" int mi_bcsr4_spmk_dev(mi_bcsr4_t A, int k, const double* d_x, double* const* d_y_out, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    if (k < 1 || k > MI_MAX_POWERS) return fail(MI_ERR_UNSUPPORTED, "k must be in 1..MI_MAX_POWERS");
    CHECK_ARG(A->nbrows == A->nbcols, "matrix powers need a square matrix");
    CHECK_ARG(d_y_out, "null output array");
    const double* src = d_x;
    for (int p = 0; p < k; p++) {
        CHECK_ARG(A->nbrows == 0 || d_y_out[p], "null output vector");
        int rc = mi_bcsr4_spmv_dev(A, src, d_y_out[p], s);
        if (rc) return rc;
        src = d_y_out[p];
    }
    return MI_OK;
}

extern "C" int mi_bcsr4_spmk(mi_bcsr4_t A, int k, const double* x, double* const* y_out)
{
    CHECK_ARG(A, "null handle");
    if (k < 1 || k > MI_MAX_POWERS) return fail(MI_ERR_UNSUPPORTED, "k must be in 1..MI_MAX_POWERS");
    CHECK_ARG(A->nbrows == A->nbcols, "matrix powers need a square matrix");
    if (A->nbrows == 0) return MI_OK;
    CHECK_ARG(x && y_out, "null vector");
    const size_t n = 4 * (size_t)A->nbrows;
    if (!A->d_x) HIP_TRY(hipMalloc(&A->d_x, sizeof(double) * n));
    while ((int)A->d_pow.size() < k) {
        double* p = nullptr;
        HIP_TRY(hipMalloc(&p, sizeof(double) * n));
        A->d_pow.push_back(p);
    }
    HIP_TRY(hipMemcpy(A->d_x, x, sizeof(double) * n, hipMemcpyHostToDevice));
    int rc = mi_bcsr4_spmk_dev(A, k, A->d_x, A->d_pow.data(), nullptr);
    if (rc) return rc;
    for (int p = 0; p < k; p++) {
        CHECK_ARG(y_out[p], "null output vector");
        HIP_TRY(hipMemcpy(y_out[p], A->d_pow[p], sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    return MI_OK;
}

// ---------------------------------------------------------------- multi-vector products, Krylov basis
template <int S>
static void launch_spmm_s(const Bcsr4View& V, int arith, const double* X, long long ldx, double* Y, long long ldy, hipStream_t s)
{
    const long long threads = 4LL * V.nbrows;
    const int nwg = (int)((threads + kWG - 1) / kWG);
    // measured on the FE matrix (bench.py --workload fe_spmm4 / fe_spmm8, MI355_SPMM_XCD=0|1): 8 columns 284 us in XCD order
    // against 343 in dispatch order (x traffic is 8x a single product's and every L2 fetched all of it); 4 columns 179
    // against 173 (not bound by x traffic yet) — so XCD order from five columns on.  MI355_SPMM_XCD=0|1 forces.
    static const int xcd_env = getenv("MI355_SPMM_XCD") ? atoi(getenv("MI355_SPMM_XCD")) : -1;
    const bool xcd = xcd_env >= 0 ? xcd_env != 0 : S > 4;
    const dim3 grid((unsigned)(xcd ? kNXCD * ((nwg + kNXCD - 1) / kNXCD) : nwg)), block(kWG);
    constexpr bool PF = S <= 4; // beyond four columns the prefetch stage costs more occupancy than it hides latency
    static const bool quad = !(getenv("MI355_SPMM_QUAD") && !strcmp(getenv("MI355_SPMM_QUAD"), "0"));
    if (S % 4 == 0 && quad) { // the quad of a block row shares its x blocks through DPP (spmv_kernels.hpp: spmm_bcsr4_quad)
        constexpr int SQ = S % 4 == 0 ? S : 4;
        static const int depth_env = getenv("MI355_SPMM_DEPTH") ? atoi(getenv("MI355_SPMM_DEPTH")) : 0;
        // measured on the FE matrix (bench.py fe_spmm4 / fe_spmm8, MI355_SPMM_DEPTH): 4 columns 254 / 168 / 168 / 173 us at depth 1 / 2 / 3 / 4
        // (registers cost occupancy: 7 / 5 / 4 / 3 waves per SIMD), 8 columns 223 / 232 / 237 us at depth 1 / 2 / 3
        const int depth = depth_env >= 1 && depth_env <= 4 ? depth_env : (S == 4 ? 2 : 1);
#define MI_SPMM_QUAD(PD)                                                                                                               \
    do {                                                                                                                               \
        if (xcd) {                                                                                                                     \
            if (arith == MI_ARITH_BLOCKACC) hipLaunchKernelGGL((spmm_bcsr4_quad<SQ, 1, true, PD>), grid, block, 0, s, V, X, ldx, Y, ldy, nwg); \
            else hipLaunchKernelGGL((spmm_bcsr4_quad<SQ, 0, true, PD>), grid, block, 0, s, V, X, ldx, Y, ldy, nwg);                    \
        } else {                                                                                                                       \
            if (arith == MI_ARITH_BLOCKACC) hipLaunchKernelGGL((spmm_bcsr4_quad<SQ, 1, false, PD>), grid, block, 0, s, V, X, ldx, Y, ldy, nwg); \
            else hipLaunchKernelGGL((spmm_bcsr4_quad<SQ, 0, false, PD>), grid, block, 0, s, V, X, ldx, Y, ldy, nwg);                   \
        }                                                                                                                              \
    } while (0)
        if (depth == 1) MI_SPMM_QUAD(1);
        else if (depth == 2) MI_SPMM_QUAD(2);
        else if (depth == 3) MI_SPMM_QUAD(3);
        else MI_SPMM_QUAD(4);
#undef MI_SPMM_QUAD
        return;
    }
    if (xcd) {
        if (arith == MI_ARITH_BLOCKACC) hipLaunchKernelGGL((spmm_bcsr4<S, 1, PF, true>), grid, block, 0, s, V, X, ldx, Y, ldy, nwg);
        else hipLaunchKernelGGL((spmm_bcsr4<S, 0, PF, true>), grid, block, 0, s, V, X, ldx, Y, ldy, nwg);
    } else {
        if (arith == MI_ARITH_BLOCKACC) hipLaunchKernelGGL((spmm_bcsr4<S, 1, PF, false>), grid, block, 0, s, V, X, ldx, Y, ldy, nwg);
        else hipLaunchKernelGGL((spmm_bcsr4<S, 0, PF, false>), grid, block, 0, s, V, X, ldx, Y, ldy, nwg);
    }
}

static int launch_spmm(mi_bcsr4_t A, int s, int arith, const double* X, long long ldx, double* Y, long long ldy, hipStream_t st,
                       bool use_map)
{
    Bcsr4View V{A->nbrows, A->nbcols, A->d_ptrow, A->d_indcol, A->d_coef, use_map ? A->d_browmap : nullptr};
    for (int j0 = 0; j0 < s; j0 += 8) { // more than eight columns: batches of eight (the matrix is read once per batch)
        const int m = std::min(8, s - j0);
        const double* Xj = X + (size_t)j0 * ldx;
        double* Yj = Y + (size_t)j0 * ldy;
        switch (m) {
        case 1: launch_spmm_s<1>(V, arith, Xj, ldx, Yj, ldy, st); break;
        case 2: launch_spmm_s<2>(V, arith, Xj, ldx, Yj, ldy, st); break;
        case 3: launch_spmm_s<3>(V, arith, Xj, ldx, Yj, ldy, st); break;
        case 4: launch_spmm_s<4>(V, arith, Xj, ldx, Yj, ldy, st); break;
        case 5: launch_spmm_s<5>(V, arith, Xj, ldx, Yj, ldy, st); break;
        case 6: launch_spmm_s<6>(V, arith, Xj, ldx, Yj, ldy, st); break;
        case 7: launch_spmm_s<7>(V, arith, Xj, ldx, Yj, ldy, st); break;
        default: launch_spmm_s<8>(V, arith, Xj, ldx, Yj, ldy, st); break;
        }
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_bcsr4_spmm_dev(mi_bcsr4_t A, int s, const double* d_X, long long ldx, double* d_Y, long long ldy, int arith,
                                 mi_stream_t st)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(s >= 0, "negative column count");
    CHECK_ARG(arith == MI_ARITH_CHAIN || arith == MI_ARITH_BLOCKACC, "unknown arithmetic id");
    if (s == 0 || A->nbrows == 0) return MI_OK;
    CHECK_ARG(d_X && d_Y, "null matrix");
    CHECK_ARG(ldx >= 4LL * A->nbcols && ldy >= 4LL * A->nbrows, "leading dimension shorter than a column");
    CHECK_ARG((((uintptr_t)d_X) & 15) == 0 && (ldx & 1) == 0, "X columns must be 16-byte aligned (even ldx)");
    return launch_spmm(A, s, arith, d_X, ldx, d_Y, ldy, (hipStream_t)st, true);
}

extern "C" int mi_bcsr4_spmm(mi_bcsr4_t A, int s, const double* X, long long ldx, double* Y, long long ldy, int arith)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(s >= 0, "negative column count");
    if (s == 0 || A->nbrows == 0) return MI_OK;
    CHECK_ARG(X && Y, "null matrix");
    CHECK_ARG(ldx >= 4LL * A->nbcols && ldy >= 4LL * A->nbrows, "leading dimension shorter than a column");
    int rc = need_device();
    if (rc) return rc;
    Scratch S;
    double *dX = nullptr, *dY = nullptr;
    const size_t nx = 4 * (size_t)A->nbcols, ny = 4 * (size_t)A->nbrows;
    if ((rc = S.up(nullptr, nx * s, &dX)) || (rc = S.up(nullptr, ny * s, &dY))) return rc;
    for (int j = 0; j < s; j++) HIP_TRY(hipMemcpy(dX + nx * j, X + (size_t)ldx * j, sizeof(double) * nx, hipMemcpyHostToDevice));
    if ((rc = mi_bcsr4_spmm_dev(A, s, dX, (long long)nx, dY, (long long)ny, arith, nullptr))) return rc;
    for (int j = 0; j < s; j++) HIP_TRY(hipMemcpy(Y + (size_t)ldy * j, dY + ny * j, sizeof(double) * ny, hipMemcpyDeviceToHost));
    return MI_OK;
}

// CSR handle: through the blocked copy when the matrix has one (matrix read once for all columns; a BCSR chain visits
// the CSR row's terms in CSR order, so every column carries the bits of SpMV_CSR_FMA), else column by column.
extern "C" int mi_spmm_dev(mi_csr_t A, int s, const double* d_X, long long ldx, double* d_Y, long long ldy, mi_stream_t st_)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(s >= 0, "negative column count");
    CHECK_ARG(!A->mapped, "multi-vector products need an unmapped matrix");
    if (s == 0 || A->n == 0) return MI_OK;
    CHECK_ARG(d_X && d_Y, "null matrix");
    CHECK_ARG(ldx >= A->ncols && ldy >= A->n, "leading dimension shorter than a column");
    hipStream_t st = (hipStream_t)st_;
    const bool aligned = (((uintptr_t)d_X) & 15) == 0 && (ldx & 1) == 0;
    if (A->inner && A->inner->blocked && aligned) { // reordered: gather the s columns into the new numbering first
        const size_t n = (size_t)A->n;
        double* Xp = nullptr; // the gathered columns as one dense block, stream-ordered allocation
        HIP_TRY(hipMallocAsync((void**)&Xp, sizeof(double) * n * s, st));
        int rc = MI_OK;
        for (int j = 0; j < s && !rc; j++) rc = gather_perm(A, d_X + (size_t)j * ldx, Xp + n * j, st);
        if (!rc) rc = launch_spmm(A->inner->blocked, s, MI_ARITH_CHAIN, Xp, (long long)n, d_Y, ldy, st, true);
        (void)hipFreeAsync(Xp, st);
        return rc;
    }
    if (!A->inner && A->blocked && aligned) return launch_spmm(A->blocked, s, MI_ARITH_CHAIN, d_X, ldx, d_Y, ldy, st, true);
    for (int j = 0; j < s; j++) {
        int rc = launch_spmv(A, d_X + (size_t)j * ldx, d_Y + (size_t)j * ldy, st);
        if (rc) return rc;
    }
    return MI_OK;
}

// y[i] /= *d (IEEE division by a scalar read from device memory)
__global__ __launch_bounds__(256) void div_by_scalar_kernel(int n, const double* __restrict__ d, double* __restrict__ y)
{
    const double q = d[0];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = __ddiv_rn(y[i], q);
}

// V[:, 0] = v0, V[:, k+1] = A V[:, k] for k < s: BuildKrylovBasis_AVX2, src/kernels/spmm_avx2.c:112-168 (a dense n x (s+1)
// column-major V, each new column one product) — that is orth == 0, the monomial basis, bit-equal to the matrix-powers
// chain.  orth != 0 builds the ORTHONORMAL (Arnoldi) basis an s-step GMRES needs out of the reference's own pieces:
// V[:, 0] = v0 / ||v0||; each product is passed through orthonormalize_against_basis (mpk/2SpMV.cpp:13-28) against the
// columns before it and then divided by its norm2 (mpk/utils.cpp:131-136) — the normalisation that helper computes and
// drops (:23-26), without which its projections y -= (y.v) v are only meaningful for unit v.  Coefficients (the Hessenberg
// column of step k) go to d_coef[k * (s + 2) + j]: j <= k the dots in the order taken, j = k + 1 the norm;
// d_coef[s * (s + 2)] = ||v0||.  d_coef: s * (s + 2) + 1 doubles.
extern "C" int mi_krylov_basis_dev(mi_csr_t A, int s, const double* d_v0, double* d_V, long long ldv, int orth, double* d_coef,
                                   mi_stream_t st_)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(s >= 0 && s <= 64, "s must be in 0..64");
    CHECK_ARG(A->n == A->ncols && !A->mapped, "a Krylov basis needs a square, unmapped matrix");
    if (A->n == 0) return MI_OK;
    CHECK_ARG(d_v0 && d_V && ldv >= A->n, "bad argument");
    CHECK_ARG(!orth || d_coef, "null coefficient array");
    hipStream_t st = (hipStream_t)st_;
    const int n = A->n;
    int grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    int rc;
    if (d_V != d_v0) HIP_TRY(hipMemcpyAsync(d_V, d_v0, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    if (orth) {
        double* nrm0 = d_coef + (size_t)s * (s + 2);
        if ((rc = mi_norm2_dev(n, d_V, nrm0, st))) return rc;
        hipLaunchKernelGGL(div_by_scalar_kernel, dim3(grid), dim3(256), 0, st, n, nrm0, d_V);
    }
    std::vector<const double*> cols;
    for (int k = 0; k < s; k++) {
        double* next = d_V + (size_t)(k + 1) * ldv;
        if ((rc = launch_spmv(A, d_V + (size_t)k * ldv, next, st))) return rc;
        if (orth) {
            double* h = d_coef + (size_t)k * (s + 2);
            cols.push_back(d_V + (size_t)k * ldv);
            if ((rc = mi_orthonormalize_against_basis_dev(n, (int)cols.size(), cols.data(), next, h, st))) return rc;
            if ((rc = mi_norm2_dev(n, next, h + k + 1, st))) return rc;
            hipLaunchKernelGGL(div_by_scalar_kernel, dim3(grid), dim3(256), 0, st, n, h + k + 1, next);
        }
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

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
    dfree(P->d_wg_halo);
    mi_csr_destroy(P->piece_all);
    P->piece_all = nullptr;
    P->d_run_link = nullptr;
    P->d_wg_halo = nullptr;
    P->fused = P->fused_bcsr = false;
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
    if (pl.sends_contiguous) { // banded partitions: the neighbours' ghosts are slices of x, sent in place
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

// ---- peer-push exchange (push_exchange.hpp) ----------------------------------------------------------------------
static int part_need_timeouts(mi_part_s* P)
{
    if (P->h_timeouts) return MI_OK;
    HIP_TRY(hipHostMalloc((void**)&P->h_timeouts, sizeof(unsigned), hipHostMallocMapped));
    *P->h_timeouts = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&P->d_timeouts, P->h_timeouts, 0));
    return MI_OK;
}

extern "C" int mi_part_push_export(mi_part_t P, void* handle64, long long* layout)
{
    CHECK_ARG(P && handle64 && layout, "null argument");
    if (!P->finalized) return fail(MI_ERR_STATE, "partition not finalized");
    static_assert(sizeof(hipIpcMemHandle_t) == MI_IPC_HANDLE_BYTES, "IPC handle size");
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
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, P->win));
    memcpy(handle64, &h, sizeof h);
    {
        std::lock_guard<std::mutex> lock(g_mu);
        P->win_key.assign((const char*)&h, sizeof h);
        g_win_registry[P->win_key] = P->win;
        P->win_registered = true;
    }
    layout[0] = pl.n_halo;
    for (int p = 0; p < pl.nranks; p++) {
        layout[1 + p] = pl.recv_offsets[p];
        layout[1 + pl.nranks + p] = pl.recv_counts[p];
    }
    return MI_OK;
}

extern "C" int mi_part_push_connect(mi_part_t P, const void* handles, const long long* layouts)
{
    CHECK_ARG(P && handles && layouts, "null argument");
    if (!P->win) return fail(MI_ERR_STATE, "mi_part_push_export was not called");
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
        void* base = nullptr;
        const std::string key((const char*)handles + (size_t)p * MI_IPC_HANDLE_BYTES, MI_IPC_HANDLE_BYTES);
        {
            std::lock_guard<std::mutex> lock(g_mu);
            auto it = g_win_registry.find(key);
            if (it != g_win_registry.end()) base = it->second; // a rank of this very process
        }
        if (!base) {
            hipIpcMemHandle_t h;
            memcpy(&h, key.data(), sizeof h);
            HIP_TRY(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
            P->ipc_opened.push_back(base);
        }
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
    if (!(fe && !strcmp(fe, "0")) && pl.n_local > 0) {
        P->plan.build_combined();
        const LocalPiece& L = P->plan.all;
        rc = csr_create_impl(pl.n_local, pl.n_local + pl.n_halo, L.ptrow.data(), L.indcol.data(), L.coef.data(), nullptr, &P->piece_all,
                             P->plan.n_left, P->plan.n_left + pl.n_local); // ghosts: columns outside [n_left, n_left + n_local)
        if (rc) return rc;
        mi_csr_t A = P->piece_all;
        if (P->kernel != MI_KERNEL_AUTO && P->kernel != MI_KERNEL_RING) A->kernel = P->kernel;
        // Where the ring serves the combined piece it is taken even if another kernel measured a hair faster on this rank: the
        // one-launch step saves three launches, and it only happens if EVERY rank has it (mi_part_push_unfuse) — a rank whose
        // create-time measurement tipped the other way by noise would cost all of them the fused step.
        if (P->kernel == MI_KERNEL_AUTO && A->ring.d_plan && A->ring.d_run_halo && A->ring.ok_fraction >= 0.90 && !A->blocked)
            A->kernel = MI_KERNEL_RING;
        P->fused = resolve_kernel(A) == MI_KERNEL_RING && A->ring.d_run_halo;
        if (P->fused) { // push duty goes to the ghost-touching runs (short by construction): link l to the (l mod k)-th of them
            std::vector<int> link((size_t)A->ring.wgs, -1);
            int k = 0;
            for (int g = 0; g < A->ring.wgs && k < P->n_links; g++)
                if (A->ring.h_run_halo[g]) link[g] = k++;
            P->npush_runs = k; // 0: no ghost runs in the plan -> dedicated push workgroups in front of the grid
            HIP_TRY(hipMalloc(&P->d_run_link, sizeof(int) * link.size()));
            HIP_TRY(hipMemcpy(P->d_run_link, link.data(), sizeof(int) * link.size(), hipMemcpyHostToDevice));
        }
        // (only while the halo is small: the fused kernel reads ghosts straight from the UNCACHED window, every use of them, and
        // pushes through two workgroups — with an FE slab's boundary planes, 39 k ghosts for 163 k rows at N = 8, that made the
        // step 39 us where push + interior + wait-and-copy + boundary as separate launches cost less; sim_rank.py N 1 fe)
        if (!P->fused && resolve_kernel(A) == MI_KERNEL_BCSR4 && A->blocked && P->plan.n_left % 4 == 0 && pl.n_local % 4 == 0 &&
            pl.n_halo % 4 == 0 && pl.n_halo <= 16384) {
            // FE matrices: the blocked copy of the combined piece, one launch of spmv_bcsr4_fused per step.  Which workgroups
            // (kWG / 4 block rows each) touch a ghost node:
            const int nbr = pl.n_local / 4, per = kWG / 4, nwg = (nbr + per - 1) / per;
            const int nl0 = P->plan.n_left, nl1 = P->plan.n_left + pl.n_local;
            std::vector<int> wg_halo((size_t)nwg, 0);
            for (int w = 0; w < nwg; w++) {
                const int r0 = 4 * w * per, r1 = std::min(pl.n_local, 4 * (w + 1) * per);
                for (int k = L.ptrow[r0]; k < L.ptrow[r1] && !wg_halo[w]; k++) wg_halo[w] = L.indcol[k] < nl0 || L.indcol[k] >= nl1;
            }
            HIP_TRY(hipMalloc(&P->d_wg_halo, sizeof(int) * wg_halo.size()));
            HIP_TRY(hipMemcpy(P->d_wg_halo, wg_halo.data(), sizeof(int) * wg_halo.size(), hipMemcpyHostToDevice));
            P->fused = P->fused_bcsr = true;
        }
        if (!P->fused) {
            mi_csr_destroy(P->piece_all);
            P->piece_all = nullptr;
        }
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
    dfree(P->d_wg_halo);
    P->d_run_link = nullptr;
    P->d_wg_halo = nullptr;
    P->fused_bcsr = false;
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
    P->fused = P->fused_bcsr = false;
    dfree(P->d_run_link);
    dfree(P->d_wg_halo);
    P->d_run_link = nullptr;
    P->d_wg_halo = nullptr;
    mi_csr_destroy(P->piece_all);
    P->piece_all = nullptr;
    return MI_OK;
}

extern "C" int mi_part_push_info(mi_part_t P, int* ready, int* fused, int* neighbours)
{
    CHECK_ARG(P, "null handle");
    if (ready) *ready = P->push_ready ? 1 : 0;
    if (fused) *fused = P->fused ? 1 : 0;
    if (neighbours) *neighbours = P->n_nb;
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
    const unsigned step = ++P->push_step;
    static const unsigned spin_max = 1u << (getenv("MI355_PUSH_SPIN_LOG2") ? std::max(8, std::min(30, atoi(getenv("MI355_PUSH_SPIN_LOG2")))) : 20); // 2^20 polls: ~4 s
    if (P->fused) { // ONE launch: push workgroups first, then the ring kernel over all rows, ghost readers waiting in-kernel
        RingComm C;
        C.links = P->d_links;
        C.send_idx = P->d_send_idx;
        C.flags = P->win_flags;
        C.nb = P->d_nb;
        C.halo = P->win_data + (size_t)(step & 1u) * (size_t)(pl.n_halo > 0 ? pl.n_halo : 1);
        C.run_halo = P->piece_all->ring.d_run_halo;
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
        if (P->fused_bcsr) {
            if ((((uintptr_t)d_x_ext) & 15) != 0) return fail(MI_ERR_ARG, "the fused blocked step needs a 16-byte aligned x");
            mi_bcsr4_t B = P->piece_all->blocked;
            Bcsr4View V{B->nbrows, B->nbcols, B->d_ptrow, B->d_indcol, B->d_coef, nullptr};
            C.push_wgs = kNXCD; // dedicated push workgroups in front: the grid is many times the resident capacity
            C.npush_runs = 0;
            const int nwg = (4 * B->nbrows + kWG - 1) / kWG;
            hipLaunchKernelGGL((spmv_bcsr4_fused<kBcsrDepth, kWG>), dim3(nwg + C.push_wgs), dim3(kWG), 0, s, V, d_x_ext, d_y_local, C, P->d_wg_halo);
            HIP_TRY(hipGetLastError());
            return MI_OK;
        }
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
        LocalPiece& L = P->plan.all;
        for (size_t k = 0; k < L.coef.size(); k++) L.coef[k] = coef[k];
        int rc = mi_csr_update_values(P->piece_all, coef);
        if (rc) return rc;
    }
    return MI_OK;
}

extern "C" int mi_part_set_kernel(mi_part_t P, int kernel_id)
{
    CHECK_ARG(P, "null handle");
    CHECK_ARG(kernel_id >= MI_KERNEL_AUTO && kernel_id <= MI_KERNEL_ROWPAR, "unknown kernel id");
    P->kernel = kernel_id;
    for (int w = 0; w < 2; w++)
        if (P->piece[w]) P->piece[w]->kernel = kernel_id;
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
