// capi_internal.hpp — state shared by the translation units that implement include/mi355_spmv.h:
// error helpers, the handle structs, and the few functions one unit needs from another.  The library is
// built from capi_lib.hip (library / device / cache flush), capi_csr.hip (CSR handles: create, plans,
// autotuner, products), launch_csr.hip (the CSR kernels' instantiations and launch_spmv), capi_blas1.hip,
// capi_bcsr.hip (BCSR 4x4, multi-vector products, Krylov basis) and capi_part.hip (partition, RCCL and
// peer-push exchange); devtools.hip (mi355_devtools.h) is linked into libmi355spmv_dev.so only.
#pragma once
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

#include "partition.hpp"
#include "push_exchange.hpp"
#include "ring_plan.hpp"
#include "spmv_kernels.hpp"
#include "spmv_ring.hpp"
#include "spmv_sstream_mw.hpp"

using namespace mi355;

// ---------------------------------------------------------------- errors
extern thread_local std::string g_err; // capi_lib.hip

static inline void dfree(void* p)
{
    if (p) (void)hipFree(p);
}

static inline int fail(int code, const std::string& msg)
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


int need_device(); // capi_lib.hip

// true while stream s is being captured into a HIP graph.  Entry points whose kernels take a per-call counter as an ARGUMENT (the
// one-launch powers step's epoch, the push step's step number) or that measure on first use must not do either under capture:
// a replayed graph would present the same counter again and every in-kernel wait would pass at once.
static inline bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
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
    bool all_in_loop = false;  // every run is ring-served and holds no PLAIN block: every row is computed inside the counted loop
    std::vector<int> h_dep_ptr, h_dep_run; // one-launch powers step: per run the runs its columns name (ring_plan.hpp: build_run_deps); empty if not built
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

// the sliced copy of the sliced-stream kernel (spmv_sstream.hpp); valid iff dev.val != nullptr
struct SstreamTable {
    bool mw = false; // the cut-ring form (spmv_sstream_mw.hpp): several column neighbourhoods per row (3-D mesh operators)
    SsDevice dev;                 // values, slot stream, workgroup records, windows, slice tables (value refills), ghost marks
    int nwg = 0, rounds = 0;
    int shift = 0;                // the rows are planned one down (an odd y offset: row pairs stay 16-byte aligned)
    long long steps = 0;
    double padding = 0.0;         // padded places per nonzero
    int max_slice_nnz = 0;        // longest CSR segment of a slice (picks the refill kernel's LDS buffer)
    bool nt = true;               // non-temporal value loads (measured at create)
    bool deep = true;             // twelve steps of prefetch instead of eight (measured at create)
    bool fusable = false;         // a combined piece's plan that spmv_sstream_fused can run (every ghost-reading workgroup's columns fit its first fill)
    bool asked = false;           // built because the environment or the caller asked for it: never released for losing a measurement
    std::vector<int> h_wg_halo;   // a combined piece of the fused multi-GPU step: per workgroup, it reads ghost columns (empty otherwise)
    std::vector<SsWg> h_wg;       // ... and its workgroup records (capi_part.hip writes the push links into them at connect time)
    double tune_us[4] = {0, 0, 0, 0}; // D = 8 nt, D = 8 temporal, D = 12 nt, D = 12 temporal
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
    int ghost_lo = 0, ghost_hi = 0; // ghost_lo < ghost_hi: a partition's combined piece, columns outside [ghost_lo, ghost_hi) are ghosts
    std::vector<int> h_ptrow; // kept to (re)build row-block tables
    std::map<int, BlockTable> tables;
    RingTable ring;           // valid iff ring.d_plan != nullptr
    TileTable tile;           // valid iff tile.d_desc != nullptr
    double tune_us_tile = 0.0, tune_us_tile_nt = 0.0;
    SstreamTable ss;          // valid iff ss.dev.val != nullptr
    MringTable mring;         // valid iff mring.d_plan != nullptr
    double tune_us_mring = 0.0, tune_us_mring_nt = 0.0;
    int kernel = MI_KERNEL_AUTO;
    int auto_kernel = MI_KERNEL_STREAM;
    std::vector<double> place_us; // placement draws at create (capi_csr.hip): microseconds per launch, value array first ([0] = as first allocated) ...
    int place_draws_coef = 0;     // ... place_us[0 .. place_draws_coef) belong to the value array, the rest to the 16-bit column stream
    double *kept_x = nullptr, *kept_y = nullptr; // the scratch pair the placement draws were timed on, kept for mi_vec_alloc_placed (its first candidate)
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
    std::vector<int> h_iperm;   // the same on the host (mi_csr_perm)
    int* d_src_start = nullptr; // [n] offset of new row r' in the caller's coef (values refresh)
    double* d_xp = nullptr;     // x in the new numbering: the buffer of the FIRST stream that multiplies with this handle ...
    hipStream_t xp_stream = nullptr;
    bool xp_claimed = false;
    std::map<hipStream_t, double*> xp_more; // ... products enqueued on other streams get a gather buffer of their own (reorder_scratch)
    std::vector<double*> d_pp;  // powers in the new numbering (mi_spmk_dev: one k-step at a time per handle)
    double* d_vtmp = nullptr;   // staging for mi_csr_update_values (host values)
    double spread_before = 0.0, spread_after = 0.0, us_natural = 0.0, us_reordered = 0.0;
    int reorder_block = 0;      // 0: no reordering attempted
    // scratch for the host-pointer entry points
    double* d_x = nullptr;
    double* d_y = nullptr;
    std::vector<double*> d_pow;
    // the one-launch matrix-powers step (spmk_ring.hpp, launch_spmk.hip): run flags, dependency lists, the launch counter the
    // flags count from, give-ups (host-visible, sticky), and per k the measured choice between one launch and k launches
    unsigned* d_kflags = nullptr;
    int* d_kdep_ptr = nullptr;
    int* d_kdep_run = nullptr;
    unsigned kstep_epoch = 0;
    unsigned* h_ktimeouts = nullptr;
    unsigned* d_ktimeouts = nullptr;
    int kstep_setup = 0;                 // 0 not tried, 1 ready, -1 not eligible
    int kstep_choice[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // per k <= 8: 0 not measured, 1 one launch, -1 k launches
    double kstep_us[9][2] = {};          // measured microseconds per step: [k][0] k launches, [k][1] one launch
};

struct SpmmTilePlan {
    int rows = 0, umax = 0, ntiles = 0; // block rows per tile (at most); longest list; tiles
    double mean_list = 0.0;
    int* d_ptr = nullptr;
    unsigned* d_nodes = nullptr;
    unsigned short* d_slots = nullptr;
    int* d_rows = nullptr;              // [ntiles * rows]: block row per lane group, -1 - r for unused places
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
    // the sliced copy (spmv_bcsr_sell.hpp): 16 block rows per slice, one contiguous stream per persistent wave; null if not built
    double* d_sell_val = nullptr;
    unsigned* d_sell_col = nullptr;
    int* d_sell_sptr = nullptr;
    int* d_sell_wrng = nullptr;  // slice ranges of the waves: [sell_nwaves + 1] for ONE wave per SIMD (1024 waves) ...
    int* d_sell_wrng2 = nullptr; // ... and [sell_nwaves2 + 1] for two (2048)
    int sell_nslices = 0, sell_nwaves = 0, sell_nwaves2 = 0;
    long long sell_nsteps = 0;
    int max_slice_vals = 0;   // values of the longest slice of 16 block rows (the refresh kernel's LDS buffer)
    int sell_form = -1;       // -1: not in use; else the variant the create-time measurement kept (kSellForms, capi_bcsr.hip)
    double tune_us_sell[4] = {0, 0, 0, 0};
    // x tiles of the multi-vector product (spmm_tile.hpp): lists per group of 128 block rows (st) and of 64 (st64: the eight-column
    // form with two quads per block row), built at the first product
    SpmmTilePlan st, st64;
    int st_state = 0;         // 0 not tried, 1 built, -1 not possible
    int spmm_choice[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // per column count <= 8: 0 not measured, else 1 + the form that measured fastest
    double spmm_us[9][5] = {};                         // [s][form] microseconds per launch: 0 gather kernels, 1 tile (four lanes per block row),
                                                       // 2 / 3 tile with eight lanes per block row, temporal / non-temporal coefficient loads,
                                                       // 4 the sliced stream (spmm_bcsr4_sell)
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
    // the all-gather form of the RCCL exchange (wide halos): every rank contributes the M-entry slice of ITS entries that anybody
    // needs (the sorted union of its send lists, padded to the largest), one ncclAllGather, each ghost picked out of the N x M result
    std::vector<int> ag_union;   // local ids, ascending
    int ag_slice = 0;            // M
    int* d_ag_idx = nullptr;     // [M] my union, padded with entry 0
    int* d_ag_src = nullptr;     // [n_halo] where ghost h lies in the gathered buffer
    double* d_ag_send = nullptr; // [M]
    double* d_ag_recv = nullptr; // [nranks * M]
    bool ag_ready = false, ag_use = false;
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
    const int* d_run_halo = nullptr; // per run (ring) / workgroup (sliced stream) of piece_all: it reads ghosts (owned by piece_all)
    int npush_runs = 0;
    bool fused = false;
    bool ghost_readers = true; // some run / workgroup of the fused launch waits for the neighbours (false: the pushers wait)
    // the blocked (FE) form of it (spmv_bcsr4_ext.hpp): piece_all numbered [owned | halo], exchange workgroups in front of the grid, ghosts staged
    bool fused_ext = false;
    double* d_stage = nullptr;   // [n_halo] cached copy of the window's current parity
    unsigned* d_ready = nullptr; // exchange workgroups done (up by ext_wgs per step)
    int ext_wgs = 0;
    int2* d_ext_units = nullptr; // per workgroup behind the exchange: {first block row, mode}
    int n_ext_units = 0;         // plain units first, then the waiting ones
    int n_ext_plain = 0;
    bool peer_on_my_device = false; // a neighbour's window lives on this rank's device: ranks share a card
    bool ext_split = false;      // two launches (exchange + plain units, then the waiting units): no workgroup but the exchange's waits in-kernel
    bool ext_csr = false;        // the staged step on SCALAR rows (spmv_csr_fused_ext): d_ext_order instead of d_ext_units
    unsigned* d_ext_order = nullptr; // per workgroup behind the exchange: its row block of piece_all's 1024-nonzero table, bit 31: it waits
    int ext_debug = 0; // devtools only (mi_debug_part_ext_mode): parts of the exchange left out, for timing
};

static inline size_t win_data_offset(int nranks) { return ((size_t)nranks * kWinFlagStride * sizeof(unsigned) + 255) / 256 * 256; }

// g_mu guards every process-wide table of the library (reduction workspaces, flush buffers, window registry)
extern std::mutex g_mu; // capi_lib.hip
int get_ws(hipStream_t s, double** out); // capi_blas1.hip: reduction workspace of (device, stream)

// host-pointer helper: upload vectors, run on the device copies, download
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

// capi_csr.hip
int csr_create_impl(int n, int ncols, const int* ptrow, const int* indcol, const double* coef, const int* rowmap, mi_csr_t* out,
                    int ghost_lo = 0, int ghost_hi = 0, bool defer_placement = false);
int resolve_kernel(const mi_csr_s* A);
int get_table(mi_csr_t A, int nnzb, BlockTable** out);
// launch_csr.hip
int launch_spmv(mi_csr_t A, const double* d_x, double* d_y, hipStream_t s, bool use_map = true, const RingComm* comm = nullptr,
                const RingDot* dot = nullptr);
bool ring_dot_eligible(const mi_csr_s* A); // the next launch_spmv(A) can carry a dot epilogue (one partial per ring workgroup)
// launch_csr.hip: the x gather buffer of a relabelled handle for products enqueued on stream s (one per stream, so that products of
// one handle on different streams do not share scratch; allocated on first use — not under stream capture)
int reorder_scratch(mi_csr_t A, hipStream_t s, double** buf);
// launch_ring.hip
void launch_ring_cfg(const mi_csr_s* A, const CsrView& V, const double* d_x, double* d_y, hipStream_t s, const RingComm* comm, const RingDot* dot);
// launch_spmk.hip: the k powers on an unmapped view of H (its row map, if any, is not applied): one launch where that is
// eligible and measured faster, else k chained launches
int spmk_unmapped(mi_csr_t H, int k, const double* d_x, double* const* d_y, hipStream_t s);
void spmk_release(mi_csr_t H);
// capi_blas1.hip
int gather_perm(mi_csr_t A, const double* d_x, double* d_xp, hipStream_t s);
int ortho_update_from_parts(int n, int nparts, const double* parts, double alpha, const double* d_b, const double* d_x1, double* d_x3, double* d_beta_out,
                            hipStream_t s);
int scatter_perm(mi_csr_t A, const double* d_src, double* d_dst, hipStream_t s);
// launch_spmm_tile.hip: the multi-vector product's tile form (spmm_tile.hpp), up to four columns
constexpr size_t kLdsBytesPerCU = 160 * 1024;
static inline size_t spmm_tile_lds(const SpmmTilePlan* T, int s) { return T ? (size_t)T->umax * (4 * s + 2) * sizeof(double) : (size_t)-1; }
hipError_t spmm_tile_launch(const mi_bcsr4_s* A, const SpmmTilePlan* Pl, const Bcsr4View& V, int s, int arith, const double* X, long long ldx,
                            double* Y, long long ldy, hipStream_t st);
hipError_t spmm_otile_launch(const mi_bcsr4_s* A, const SpmmTilePlan* Pl, const Bcsr4View& V, int s, int arith, bool nt, const double* X, long long ldx,
                             double* Y, long long ldy, hipStream_t st);
// capi_part.hip: pieces of the peer-push set-up that capi_dist.hip (ranks of one process on different devices) uses directly
int part_push_window(mi_part_s* P);
void part_push_layout(const mi_part_s* P, long long* layout /* [2*nranks + 1] */);
int part_push_connect_bases(mi_part_s* P, void* const* bases /* [nranks] */, const long long* layouts);
// the staged one-launch step of a blocked rank (spmv_bcsr4_ext.hpp); trace: devtools only (3 stamps per workgroup of the grid)
int part_ext_launch(mi_part_s* P, const double* d_x_ext, double* d_y_local, unsigned step, unsigned spin_max, hipStream_t s, unsigned long long* trace, int* grid_out);
// capi_csr.hip: the sliced-stream kernel of a handle (spmv_sstream.hpp); d_y: where the handle's row 0 goes (unmapped) or the mapped vector's base;
// comm: the fused multi-GPU step (the handle is a partition's combined piece with ghost marks)
int launch_sstream(mi_csr_t A, const double* d_x, double* d_y, const int* rowmap, hipStream_t s, const RingComm* comm = nullptr);
// the sliced-stream kernel can write this y (row pairs as 16 bytes; a mapped handle stores row by row)
static inline bool sstream_y_ok(const mi_csr_s* A, const double* yy, const int* map) { return map || A->ss.mw || (((uintptr_t)(yy - A->ss.shift)) & 15) == 0; } // (the cut-ring form stores row by row)
// capi_bcsr.hip
// the blocked copy's values were rewritten on stream s (by whoever holds d_coef): the sliced copy follows at once, on the same stream
int bcsr4_values_changed(mi_bcsr4_s* A, hipStream_t s);
// give the sliced copy up (a handle whose products never run the sliced kernels: the partition's combined piece of a blocked one-launch step)
void bcsr4_drop_sliced(mi_bcsr4_s* A);
// a CSR handle's blocked copy (blocks + sliced values) from its CSR values d_src, which also go to d_csr_out when that is given: one pass
int bcsr4_refresh_from_csr(mi_bcsr4_s* A, const int* d_csr_ptrow, const double* d_src, double* d_csr_out, hipStream_t s);
int launch_bcsr4(mi_bcsr4_t A, const double* d_x, double* d_y, mi_stream_t s, bool use_map);
