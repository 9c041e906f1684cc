// capi_csr.hip: CSR handles — create (plans, autotuner, relabelling), update, info, products — part of libmi355spmv.so (see capi_internal.hpp for the layout of the library).
// Built for gfx950 only; no CPU fallback anywhere: every compute entry point needs a HIP device.
#include "capi_internal.hpp"
#include "reorder.hpp"
#include "mring_plan.hpp"
#include "tile_plan.hpp"

// ---------------------------------------------------------------- CSR create
int get_table(mi_csr_t A, int nnzb, BlockTable** out)
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

static void free_sstream(mi_csr_t A)
{
    ss_free(A->ss.dev);
    A->ss = SstreamTable();
}

constexpr double kSsMaxPadding = 0.12; // padded places per nonzero from which the sliced copy is not worth its bytes
// MI355_SSTREAM_MAX_PADDING=<places per nonzero> (tests: the reference-made goldens — a few hundred ragged rows — through the sliced kernels)
static double ss_max_padding()
{
    if (const char* e = getenv("MI355_SSTREAM_MAX_PADDING")) return std::max(0.0, atof(e));
    return kSsMaxPadding;
}

// The sliced copy of the sliced-stream kernel (spmv_sstream.hpp): plan + slot stream on the host, values filled on the device from the
// handle's CSR values (already uploaded).  MI_OK with A->ss left empty when the matrix is not eligible.
// A handle whose rows go to y[r + odd offset] (the interior rows of a rank) is planned one row down: its row pairs then start on
// even rows of y and leave as 16-byte stores like any other's.  ghost_lo < ghost_hi: a combined piece of the fused multi-GPU step.
static int build_sstream(mi_csr_t A, const int* ptrow, const int* indcol, int ghost_lo = 0, int ghost_hi = 0)
{
    if (A->ss.dev.val) return MI_OK;
    SsPlanHost P;
    const int shift = (A->y_offset & 1) && !A->d_rowmap ? 1 : 0;
    build_sstream_plan(A->n, A->ncols, ptrow, indcol, ss_max_padding(), P, true, shift, ghost_lo, ghost_hi);
    SstreamTable& T = A->ss;
    hipError_t e;
    if (!P.eligible) {
        // rows that name several column neighbourhoods (3-D mesh operators in natural order): the cut-ring form (spmv_sstream_mw.hpp).
        // Not for a combined piece with ghost columns (the fused step has no such form).  MI355_SSTREAM_MW=0: never.
        if (ghost_lo < ghost_hi || (getenv("MI355_SSTREAM_MW") && !strcmp(getenv("MI355_SSTREAM_MW"), "0"))) return MI_OK;
        SsMwPlanHost M;
        build_sstream_mw_plan(A->n, A->ncols, ptrow, indcol, ss_max_padding(), M, shift);
        if (!M.eligible) return MI_OK;
        e = ss_mw_upload(M, T.dev);
        T.mw = e == hipSuccess;
        P = M; // (the tables both forms share: rounds, steps, slices)
    } else e = ss_upload(P, T.dev, ghost_lo < ghost_hi);
    if (e != hipSuccess) {
        free_sstream(A);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return MI_OK; } // no room for a second copy of the matrix: the other kernels serve it
        return fail(MI_ERR_HIP, std::string("sliced copy: ") + hipGetErrorString(e));
    }
    T.nwg = P.nwg;
    T.rounds = P.rounds;
    T.shift = P.shift;
    T.steps = P.steps;
    T.padding = (double)P.pad_places / (double)A->nnz;
    T.max_slice_nnz = P.max_slice_nnz;
    if (ghost_lo < ghost_hi) {
        T.h_wg_halo = P.wg_halo;
        T.h_wg = P.wg;
        T.fusable = P.fusable;
    }
    // the sliced values are filled HERE and refilled where the CSR values change (mi_csr_update_values*), on that call's stream — never
    // lazily in front of a product: a product captured into a HIP graph holds only the product's node and must find the values in place
    sstream_fill_values(T.rounds, A->n, T.shift, A->d_ptrow, A->d_coef, nullptr, T.dev.slice_step, T.dev.slice_len, T.dev.val, T.max_slice_nnz, nullptr, T.mw ? 1 : 0);
    if ((e = hipGetLastError()) != hipSuccess) {
        free_sstream(A);
        return fail(MI_ERR_HIP, std::string("sliced copy fill: ") + hipGetErrorString(e));
    }
    T.nt = 10.0 * (double)A->nnz + 16.0 * (double)A->n > 0.75 * 256e6; // beyond the Infinity Cache: stream past it
    T.deep = T.nt;
    if (const char* fe = getenv("MI355_SSTREAM_FORM")) { // tests: one variant (0 D=8 nt, 1 D=8 temporal, 2 D=12 nt, 3 D=12 temporal)
        const int f = atoi(fe);
        T.nt = (f & 1) == 0;
        T.deep = f >= 2;
    }
    return MI_OK;
}

// y = A x through the sliced-stream kernel (the caller has checked sstream_y_ok); rowmap: nullptr or the handle's row map
int launch_sstream(mi_csr_t A, const double* d_x, double* d_y, const int* rowmap, hipStream_t s, const RingComm* comm)
{
    SstreamTable& T = A->ss;
    SsView S{T.dev.val, T.dev.slot, T.dev.wg, T.dev.win, T.nwg, A->n + T.shift, A->ncols, rowmap, T.shift};
    double* yy = rowmap ? d_y : d_y - T.shift;
    if (T.mw) {
        if (comm) return fail(MI_ERR_STATE, "the cut-ring sliced stream has no fused multi-GPU form");
        SsMwView V{S, T.dev.winK};
        if (T.deep) {
            if (T.nt) hipLaunchKernelGGL((spmv_sstream_mw<12, true>), dim3((unsigned)T.nwg), dim3(256), 0, s, V, d_x, yy);
            else hipLaunchKernelGGL((spmv_sstream_mw<12, false>), dim3((unsigned)T.nwg), dim3(256), 0, s, V, d_x, yy);
        } else {
            if (T.nt) hipLaunchKernelGGL((spmv_sstream_mw<8, true>), dim3((unsigned)T.nwg), dim3(256), 0, s, V, d_x, yy);
            else hipLaunchKernelGGL((spmv_sstream_mw<8, false>), dim3((unsigned)T.nwg), dim3(256), 0, s, V, d_x, yy);
        }
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
    if (comm) {
        const unsigned grid = (unsigned)(T.nwg + comm->push_wgs);
        if (T.deep) {
            if (T.nt) hipLaunchKernelGGL((spmv_sstream_fused<12, true>), dim3(grid), dim3(256), 0, s, S, d_x, yy, *comm);
            else hipLaunchKernelGGL((spmv_sstream_fused<12, false>), dim3(grid), dim3(256), 0, s, S, d_x, yy, *comm);
        } else {
            if (T.nt) hipLaunchKernelGGL((spmv_sstream_fused<8, true>), dim3(grid), dim3(256), 0, s, S, d_x, yy, *comm);
            else hipLaunchKernelGGL((spmv_sstream_fused<8, false>), dim3(grid), dim3(256), 0, s, S, d_x, yy, *comm);
        }
    } else if (T.deep) {
        if (T.nt) hipLaunchKernelGGL((spmv_sstream<12, true>), dim3((unsigned)T.nwg), dim3(256), 0, s, S, d_x, yy);
        else hipLaunchKernelGGL((spmv_sstream<12, false>), dim3((unsigned)T.nwg), dim3(256), 0, s, S, d_x, yy);
    } else {
        if (T.nt) hipLaunchKernelGGL((spmv_sstream<8, true>), dim3((unsigned)T.nwg), dim3(256), 0, s, S, d_x, yy);
        else hipLaunchKernelGGL((spmv_sstream<8, false>), dim3((unsigned)T.nwg), dim3(256), 0, s, S, d_x, yy);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// Plan of the multi-window ring kernel (host arrays of the caller, or nullptr: the handle's device copy is read back)
// Block shape of the ring / multi-window ring plans by matrix size (row_align argument of the planners; 0 = their default of 64).
// Whole waves of rows per block pay below ~20 M nonzeros (2-5 % on every box tried).  Beyond, their rate turned out to depend on
// WHERE THE CALLER'S x AND y lie in device memory — 133 / 144 / 153 us for three x / y pairs on one and the same C4 handle, and
// 149-154 us for sixteen pairs in another process — while blocks cut at the nonzero count alone run 138-146 us whatever the
// vectors (tools/state_probe.py, state_probe2.py; profiles/r03_state_probe.txt): their 1088-byte steps through y, x and the
// row pointers keep the 512 workgroups' streams out of phase with each other, the whole-wave plan's 1 KiB steps do not.  The
// create-time comparison of the two (rounds 2-3) timed them on the library's own scratch vectors and so could not tell what the
// caller would see; large matrices now take the unaligned blocks outright.
// Staging layout of the ring kernels' row chains (SKEW template argument): one pad slot per 32 staged nonzeros, or none.  A row
// chain reads its terms as 16-byte {coef, x} pairs (ring_stage_cx), lane l starting 16 * (length of the rows in front) bytes into the
// staging area: rows whose length is a multiple of 16 put every lane of a wave on the same four banks (16-way conflicts: uniform
// rows of 16 run 134-139 us plain and 92-102 us padded, rows of 32 132-136 against 107.5, rows of 8 89 against 79-85), and the pad
// slot breaks that up — at the price of index arithmetic per term, which is why every other shape is faster WITHOUT it: S15 140.5
// against 147.5 us, SVAR (lengths 8..22 mixed) 147 against 159-163, rows of 24 96.7 against 103.5, the FE shape (rows of 56) 162.6
// against 201.  (Measured with the merged staging of round 3's second session, tools/skew_ab.py; the two-array staging of
// rounds 1-2 wanted the pad for every multiple of 8.)  Mixed lengths do not line up whatever their share, so the rule asks for a
// MAJORITY of such rows.
static bool ring_wants_skew(int n, const int* ptrow)
{
    long long bad = 0;
    for (int i = 0; i < n; i++) {
        const int len = ptrow[i + 1] - ptrow[i];
        bad += len > 0 && (len % 16 == 0 || len == 8);
    }
    return 2 * bad > n;
}

constexpr long long kLargeNnz = 20000000;
// doubles in a vector handed out by mi_vec_alloc_placed (64 of slack: the kernels' clamped loads stay inside)
static size_t placed_vector_len(int n, int ncols) { return (size_t)std::max(n, ncols) + 64; }
static int large_row_align(long long nnz) { return nnz >= kLargeNnz ? 1 : 0; }

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
    M.skew = ring_wants_skew(A->n, A->h_ptrow.data());
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
static int fill_ring_table(RingTable& R, const RingPlanHost& best, int n, const int* ptrow, const int* indcol, long long nnz, bool ghosts,
                           const int* row_min = nullptr, const int* row_max = nullptr, bool square = false)
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
    R.all_in_loop = true;
    for (int g = 0; g < best.wgs; g++) R.all_in_loop = R.all_in_loop && best.run_ok[g] == 1;
    R.h_dep_ptr.clear();
    R.h_dep_run.clear();
    if (square && row_min && R.all_in_loop && R.lean && best.cfg.id == 4 && !ghosts) build_run_deps(best, n, R.h_dep_ptr, R.h_dep_run);
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
    R.skew = ring_wants_skew(n, ptrow);
    if (const char* e2 = getenv("MI355_RING_SKEW")) R.skew = atoi(e2) != 0;
    return MI_OK;
}

static int time_handle(mi_csr_t A, int warm, int timed, double* us);

// Placement draws (round 3, profiles/NOTES.md §4.12; FROZEN in round 4: no further work goes into it).  WHERE the value array lies in device
// memory moves a warm launch of the streaming kernels by up to 15 % on some boxes (tools/placement_lottery.py: 137-139 us against 158-166
// for handles of one and the same matrix and plan in one process; the coefficient array decides, the 16-bit column stream adds a few us,
// row pointers and plan records nothing; no allocation flag or address property found that predicts it).  So for matrices beyond the
// caches the chosen kernel is timed on a few fresh copies of those two arrays and the fastest copy is the one kept; every candidate
// stays allocated until the draws are over (a freed block would just be handed out again).
// Default since round 4: at most 4 draws of the value array (2 of the column stream), and none after the first when the first copy
// times within 2 % of the original (a box where placements are alike: nothing to find).  MI355_PLACEMENT_DRAWS=N (0..16) sets the
// number and disables the early stop — the 12-draw / 7 GB form of round 3 is MI355_PLACEMENT_DRAWS=12; =0 turns the draws off.
// tx / ty: the x / y pair to time on (null: a fresh pair is allocated; kept as A->kept_x/y when keep_pair and draws were made).
static int placement_draws(mi_csr_t A, double* tx, double* ty, bool keep_pair)
{
    const long long nnz = A->nnz;
    const char* pe = getenv("MI355_PLACEMENT_DRAWS");
    int draws = pe ? std::max(0, std::min(16, atoi(pe))) : 4;
    const bool early_stop = pe == nullptr;
    if (!pe) {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && nnz > 0)
            draws = (int)std::min<size_t>((size_t)draws, mem_free / 8 / (sizeof(double) * (size_t)nnz));
        else (void)hipGetLastError();
    }
    // (a blocked copy served by its SLICED form streams d_sell_val, never d_coef: drawing d_coef there timed noise and could swap the array on it — ADVICE r4)
    const bool blocked_choice = A->auto_kernel == MI_KERNEL_BCSR4 && A->blocked && A->blocked->d_coef && !(A->blocked->sell_form >= 0 && A->blocked->d_sell_val);
    const bool streams_coef = A->auto_kernel == MI_KERNEL_RING || A->auto_kernel == MI_KERNEL_MRING || A->auto_kernel == MI_KERNEL_STREAM || A->auto_kernel == MI_KERNEL_TILE || blocked_choice;
    if (!(draws > 0 && streams_coef && nnz >= kLargeNnz)) return MI_OK;
    struct Own {
        double *x = nullptr, *y = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Own()
        {
            dfree(x);
            dfree(y);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } own;
    if (!tx || !ty) {
        const size_t len = std::max(placed_vector_len(A->n, A->ncols), (size_t)(A->n_out > 0 ? A->n_out : 1));
        HIP_TRY(hipMalloc(&own.x, sizeof(double) * len));
        HIP_TRY(hipMalloc(&own.y, sizeof(double) * len));
        HIP_TRY(hipMemset(own.x, 0, sizeof(double) * len));
        tx = own.x;
        ty = own.y;
    }
    HIP_TRY(hipEventCreate(&own.e0));
    HIP_TRY(hipEventCreate(&own.e1));
    auto time_now = [&](int warm, int timed, double* us_out) -> int {
        int rc2;
        for (int w = 0; w < warm; w++)
            if ((rc2 = launch_spmv(A, tx, ty, nullptr))) return rc2;
        if (hipEventRecord(own.e0, nullptr) != hipSuccess) return MI_ERR_HIP;
        for (int w = 0; w < timed; w++)
            if ((rc2 = launch_spmv(A, tx, ty, nullptr))) return rc2;
        float ms = 0.f;
        if (hipEventRecord(own.e1, nullptr) != hipSuccess || hipEventSynchronize(own.e1) != hipSuccess || hipEventElapsedTime(&ms, own.e0, own.e1) != hipSuccess) return MI_ERR_HIP;
        *us_out = ms * 1e3 / timed;
        return MI_OK;
    };
    // pad_bytes: the zeroed tail the array was ALLOCATED with behind its payload — the ring family's unclamped loads read up to
    // kRingPadNnz values past the last nonzero, the blocked kernel one block past the last block; a copy must carry it too
    // (round 3's draws copied the payload into an exactly-sized buffer: every product on a redrawn array then read past its end)
    auto redraw = [&](void** slot, size_t bytes, size_t pad_bytes, int ndraws) {
        double best = 0.0;
        if (!*slot || bytes == 0 || time_now(3, 8, &best) != MI_OK) return;
        const double first = best;
        A->place_us.push_back(best);
        std::vector<void*> losers;
        for (int d = 0; d < ndraws; d++) {
            void* fresh = nullptr;
            if (hipMalloc(&fresh, bytes + pad_bytes) != hipSuccess) { (void)hipGetLastError(); break; } // no room for a copy: keep what there is
            if ((pad_bytes && hipMemset((char*)fresh + bytes, 0, pad_bytes) != hipSuccess) ||
                hipMemcpy(fresh, *slot, bytes, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipGetLastError(); dfree(fresh); break; }
            std::swap(*slot, fresh); // fresh = the previous holder from here
            double t = 0.0;
            const int rct = time_now(3, 8, &t);
            A->place_us.push_back(rct == MI_OK ? t : -1.0);
            if (rct == MI_OK && t < 0.96 * best) best = t; // the copy is clearly faster (two timings of ONE placement differ by 2-3 %): it holds the data from now on
            else std::swap(*slot, fresh);
            losers.push_back(fresh);
            if (early_stop && d == 0 && rct == MI_OK && std::fabs(t - first) <= 0.02 * first) break; // placements are alike on this box
        }
        for (void* l : losers) dfree(l);
    };
    // (a matrix that runs the blocked kernel streams the BLOCKED copy's values: 16 doubles per block, zero fill included)
    if (blocked_choice) redraw((void**)&A->blocked->d_coef, sizeof(double) * 16 * (size_t)A->blocked->nblocks, sizeof(double) * 16, draws); // capi_bcsr.hip: 16 * (nb + 1)
    else redraw((void**)&A->d_coef, sizeof(double) * (size_t)nnz, sizeof(double) * (size_t)kRingPadNnz, draws);
    A->place_draws_coef = (int)A->place_us.size();
    // (the slot streams are allocated at exactly nblk * nnzb entries: no tail to carry)
    if (A->auto_kernel == MI_KERNEL_RING) redraw((void**)&A->ring.d_slots, sizeof(unsigned short) * (size_t)A->ring.nblk * A->ring.cfg.nnzb, 0, (draws + 1) / 2);
    else if (A->auto_kernel == MI_KERNEL_MRING) redraw((void**)&A->mring.d_slots, sizeof(unsigned short) * (size_t)A->mring.nblk * kMringNnzb, 0, (draws + 1) / 2);
    // The draws chose copies that are fast WITH THIS x / y pair; a caller that asks the library for its vectors gets the pair
    // itself as the first candidate (user-facing square handles only: a partition's pieces and mapped views are not handed vectors)
    if (keep_pair && !A->kept_x) {
        A->kept_x = tx;
        A->kept_y = ty;
        if (tx == own.x) own.x = own.y = nullptr;
    }
    return MI_OK;
}

int csr_create_impl(int n, int ncols, const int* ptrow, const int* indcol, const double* coef,
                    const int* rowmap, mi_csr_t* out, int ghost_lo, int ghost_hi, bool defer_placement)
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
            build_ring_plan(kRingConfigs[id - 1], n, ptrow, row_min.data(), row_max.data(), P, ghost_lo, ghost_hi, large_row_align(nnz));
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
            const int rcr = fill_ring_table(A->ring, best, n, ptrow, indcol, nnz, ghost_lo < ghost_hi, row_min.data(), row_max.data(), n == ncols);
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
            const int rcm = build_mring(A, indcol, large_row_align(nnz));
            if (rcm != MI_OK) {
                mi_csr_destroy(A);
                return rcm;
            }
            if (!asked && A->mring.ok_fraction < 0.90) free_mring(A);
        }
    }
    // ... and the sliced copy of the sliced-stream kernel (spmv_sstream.hpp): unmapped matrices whose rounds of 512 rows fit the LDS
    // window and whose slices pad little; MI355_SSTREAM=0 never, =1 whatever the size
    {
        const char* se = getenv("MI355_SSTREAM");
        const char* ke = getenv("MI355_SPMV_KERNEL");
        const bool asked = (se && !strcmp(se, "1")) || (ke && !strcmp(ke, "sstream"));
        // (row-mapped handles — partition pieces, the relabelled twin — included since the store goes through the map; since round 5 also
        // the fused multi-GPU step's combined piece: spmv_sstream_fused reads its ghost columns from the receive window)
        A->ghost_lo = ghost_lo;
        A->ghost_hi = ghost_hi;
        if (n > 0 && nnz > 0 && !(se && !strcmp(se, "0")) && (asked || nnz >= 200000)) {
            const int rcs = build_sstream(A, ptrow, indcol, ghost_lo, ghost_hi);
            if (rcs != MI_OK) {
                mi_csr_destroy(A);
                return rcs;
            }
            A->ss.asked = asked;
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
        else if (!strcmp(e, "sstream") && A->ss.dev.val) A->auto_kernel = MI_KERNEL_SSTREAM;
        else forced_kernel = false;
    }
    const char* at = getenv("MI355_SPMV_AUTOTUNE");
    // Too small to be measured (< 200 000 nonzeros: a partition's boundary rows, a test matrix): the stream kernel — a launch of as
    // many workgroups as there are row blocks — not the ring kernel's persistent grid with its plan loads and window fill: a rank's
    // 2 000 boundary rows cost 11-17 us through the ring and 4.8 through the stream kernel (profiles/r05_sp2_trace_tail.txt).
    if (!forced_kernel && nnz < 200000 && A->auto_kernel == MI_KERNEL_RING && !(ghost_lo < ghost_hi)) A->auto_kernel = MI_KERNEL_STREAM;
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
        // (sized like the vectors mi_vec_alloc_placed hands out: for a large square matrix this very pair becomes its first candidate)
        const size_t scratch_len = placed_vector_len(n, ncols);
        TRY_OR_CLEAN(hipMalloc(&tx, sizeof(double) * std::max(scratch_len, (size_t)(ncols > 0 ? ncols : 1))));
        TRY_OR_CLEAN(hipMalloc(&ty, sizeof(double) * std::max(scratch_len, (size_t)(A->n_out > 0 ? A->n_out : 1))));
        TRY_OR_CLEAN(hipMemset(tx, 0, sizeof(double) * std::max(scratch_len, (size_t)(ncols > 0 ? ncols : 1))));
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
        if (A->ss.dev.val && !getenv("MI355_SSTREAM_FORM")) { // the sliced stream against whatever won so far: 8 / 12 steps of prefetch x non-temporal / temporal value loads
            const double best_so_far = A->auto_kernel == MI_KERNEL_BCSR4 ? A->tune_us_bcsr
                                       : A->auto_kernel == MI_KERNEL_RING ? best_ring : (A->auto_kernel == MI_KERNEL_TILE ? best_tile : (A->auto_kernel == MI_KERNEL_MRING ? best_mring : best_stream));
            A->kernel = MI_KERNEL_SSTREAM;
            double bs = 0.0;
            int bf = -1;
            for (int round = 0; round < 2; round++)
                for (int f = 0; f < 4; f++) {
                    A->ss.nt = (f & 1) == 0;
                    A->ss.deep = f >= 2;
                    for (int w = 0; w < 3; w++)
                        if (launch_spmv(A, tx, ty, nullptr) != MI_OK) break;
                    TRY_OR_CLEAN(hipEventRecord(e0, nullptr));
                    const int timed = nnz < 40000000 ? 12 : 6;
                    for (int w = 0; w < timed; w++)
                        if (launch_spmv(A, tx, ty, nullptr) != MI_OK) break;
                    TRY_OR_CLEAN(hipEventRecord(e1, nullptr));
                    TRY_OR_CLEAN(hipEventSynchronize(e1));
                    float ms = 0.f;
                    TRY_OR_CLEAN(hipEventElapsedTime(&ms, e0, e1));
                    const double t = ms * 1e3 / timed;
                    A->ss.tune_us[f] = A->ss.tune_us[f] > 0 ? std::min(A->ss.tune_us[f], t) : t;
                }
            for (int f = 0; f < 4; f++)
                if (A->ss.tune_us[f] > 0 && (bf < 0 || A->ss.tune_us[f] < bs)) { bs = A->ss.tune_us[f]; bf = f; }
            A->kernel = MI_KERNEL_AUTO;
            if (bf >= 0) {
                A->ss.nt = (bf & 1) == 0;
                A->ss.deep = bf >= 2;
                if (better(bs, best_so_far)) A->auto_kernel = MI_KERNEL_SSTREAM;
            }
            // A sliced copy that lost the measurement is released (+10 B per nonzero and its padding): mi_csr_set_kernel(MI_KERNEL_SSTREAM)
            // rebuilds it on demand.  Kept: one the environment asked for, and a combined piece's (the fused multi-GPU step takes the
            // sliced stream wherever EVERY rank can: its choice is collective, not this rank's measurement).
            if (A->auto_kernel != MI_KERNEL_SSTREAM && !A->ss.asked && !(ghost_lo < ghost_hi)) {
                double keep_us[4];
                for (int f = 0; f < 4; f++) keep_us[f] = A->ss.tune_us[f];
                free_sstream(A);
                for (int f = 0; f < 4; f++) A->ss.tune_us[f] = keep_us[f]; // (what it measured stays readable: mi_csr_sstream_info)
            }
        }
        // every further comparison on the SAME x / y scratch: where a vector lies in device memory moves a launch by a few per cent
        auto time_now = [&](int warm, int timed, double* us_out) -> int {
            int rc2;
            for (int w = 0; w < warm; w++)
                if ((rc2 = launch_spmv(A, tx, ty, nullptr))) return rc2;
            if (hipEventRecord(e0, nullptr) != hipSuccess) return MI_ERR_HIP;
            for (int w = 0; w < timed; w++)
                if ((rc2 = launch_spmv(A, tx, ty, nullptr))) return rc2;
            float ms = 0.f;
            if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return MI_ERR_HIP;
            *us_out = ms * 1e3 / timed;
            return MI_OK;
        };
        // Large ring-served matrices take UNALIGNED blocks (large_row_align()).  MI355_RING_SHAPE_COMPARE=1 also builds the plan of
        // whole-wave blocks and times both on the create-time scratch vectors (mi_csr_ring_shape_info; the shape in use stays unless
        // the other is 2 % faster) — a comparison that says little about the caller's vectors, which is why it is no longer the rule.
        // (a rank's combined piece of the fused multi-GPU step included: timed here without the exchange, as a plain product)
        const bool shape_compare = getenv("MI355_RING_SHAPE_COMPARE") && !strcmp(getenv("MI355_RING_SHAPE_COMPARE"), "1") && !getenv("MI355_RING_ROW_ALIGN");
        if (shape_compare && A->auto_kernel == MI_KERNEL_RING && A->ring.cfg.id == 4 && nnz >= kLargeNnz) {
            RingPlanHost alt;
            build_ring_plan(kRingConfigs[3], n, ptrow, row_min.data(), row_max.data(), alt, ghost_lo, ghost_hi, 64);
            RingTable T2;
            if (nnz > 0 && 1.0 - (double)alt.bad_nnz / (double)nnz >= 0.90 &&
                fill_ring_table(T2, alt, n, ptrow, indcol, nnz, ghost_lo < ghost_hi, row_min.data(), row_max.data(), n == ncols) == MI_OK) {
                T2.nt = A->ring.nt;
                double us64 = 0.0, us1 = 0.0;
                A->kernel = MI_KERNEL_RING;
                int rct = time_now(3, 8, &us1);
                std::swap(A->ring, T2);
                if (rct == MI_OK) rct = time_now(3, 8, &us64);
                A->kernel = MI_KERNEL_AUTO;
                A->tune_us_ring_aligned = us64;
                A->tune_us_ring_unaligned = us1;
                if (rct != MI_OK || !(us64 < 0.98 * us1)) std::swap(A->ring, T2); // keep the default unless the other is clearly faster
                free_ring_table(T2);
            }
        }
        if (shape_compare && A->auto_kernel == MI_KERNEL_MRING && nnz >= kLargeNnz) { // the same for the multi-window ring
            MringTable keep = A->mring;
            A->mring = MringTable();
            if (build_mring(A, indcol, 64) == MI_OK && A->mring.d_plan && A->mring.ok_fraction >= 0.90) {
                A->mring.nt = keep.nt;
                double us64 = 0.0, us1 = 0.0;
                A->kernel = MI_KERNEL_MRING;
                int rct = time_now(3, 8, &us64);
                std::swap(A->mring, keep);
                if (rct == MI_OK) rct = time_now(3, 8, &us1);
                A->kernel = MI_KERNEL_AUTO;
                A->tune_us_ring_aligned = us64;
                A->tune_us_ring_unaligned = us1;
                if (rct == MI_OK && us64 < 0.98 * us1) std::swap(A->mring, keep); // the whole-wave plan is clearly faster here
            } else {
                std::swap(A->mring, keep);
            }
            MringTable loser = keep; // release the plan not kept
            keep = A->mring;
            A->mring = loser;
            free_mring(A);
            A->mring = keep;
        }
        // placement draws (placement_draws() below): on the scratch pair the comparisons above ran on; deferred when the caller
        // (mi_csr_create) may still replace this handle's arrays by a relabelled twin
        if (!defer_placement) {
            const int rcp = placement_draws(A, tx, ty, !rowmap && n == ncols && n == A->n_out && !(ghost_lo < ghost_hi));
            if (rcp == MI_OK && A->kept_x == tx) tx = ty = nullptr; // the handle keeps the pair (mi_vec_alloc_placed's first candidate)
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
    free_sstream(A);
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
    if (!force && (n < 100000 || resolve_kernel(A) == MI_KERNEL_RING || resolve_kernel(A) == MI_KERNEL_SSTREAM)) return MI_OK; // (a band served by its window: nothing to relabel)
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
    A->h_iperm = R.iperm; // host copy for mi_csr_perm (callers that keep their vectors in the library's numbering)
    A->xp_claimed = false; // (the measurement above ran on the default stream: the gather buffer goes to the caller's first stream)
    release_natural_arrays(A);
    return MI_OK;
}

extern "C" int mi_csr_create(int n, int ncols, const int* ptrow, const int* indcol, const double* coef, mi_csr_t* out)
{
    // (placement draws deferred: a handle that maybe_reorder replaces by its relabelled twin would draw for arrays it is about to release)
    int rc = csr_create_impl(n, ncols, ptrow, indcol, coef, nullptr, out, 0, 0, true);
    if (rc) return rc;
    if ((rc = maybe_reorder(*out, ptrow, indcol, coef)) || (!(*out)->inner && (rc = placement_draws(*out, nullptr, nullptr, n == ncols)))) {
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

// ---- the library's own numbering, for callers that own the loop (a Krylov solve: many products per matrix) ----------------
extern "C" int mi_csr_perm(mi_csr_t A, int* reordered, int* perm)
{
    CHECK_ARG(A, "null handle");
    if (reordered) *reordered = A->inner ? 1 : 0;
    if (perm) {
        if (A->inner) {
            CHECK_ARG((int)A->h_iperm.size() == A->n, "permutation not kept on this handle");
            for (int r = 0; r < A->n; r++) perm[A->h_iperm[r]] = r; // perm[old] = new
        } else {
            for (int i = 0; i < A->n; i++) perm[i] = i;
        }
    }
    return MI_OK;
}

extern "C" int mi_vec_to_internal_dev(mi_csr_t A, const double* d_x, double* d_x_int, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    if (A->n == 0) return MI_OK;
    CHECK_ARG(d_x && d_x_int && d_x != d_x_int, "null vector, or in place");
    if (A->inner) return gather_perm(A, d_x, d_x_int, (hipStream_t)s);
    HIP_TRY(hipMemcpyAsync(d_x_int, d_x, sizeof(double) * (size_t)A->n, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return MI_OK;
}

extern "C" int mi_vec_from_internal_dev(mi_csr_t A, const double* d_x_int, double* d_x, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    if (A->n == 0) return MI_OK;
    CHECK_ARG(d_x && d_x_int && d_x != d_x_int, "null vector, or in place");
    if (A->inner) return scatter_perm(A, d_x_int, d_x, (hipStream_t)s);
    HIP_TRY(hipMemcpyAsync(d_x, d_x_int, sizeof(double) * (size_t)A->n, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return MI_OK;
}

extern "C" int mi_spmv_internal_dev(mi_csr_t A, const double* d_x_int, double* d_y_int, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(A->n == A->ncols && !A->mapped, "the internal numbering is defined for square, unmapped matrices");
    CHECK_ARG(A->n == 0 || (d_x_int && d_y_int), "null vector");
    // a relabelled handle: the twin alone, reading and writing its own numbering — no gather, no row map, no per-handle
    // scratch (so products on several streams may overlap); any other handle: its numbering IS the caller's
    if (A->inner) return launch_spmv(A->inner, d_x_int, d_y_int, (hipStream_t)s, false);
    return launch_spmv(A, d_x_int, d_y_int, (hipStream_t)s);
}

extern "C" int mi_spmk_internal_dev(mi_csr_t A, int k, const double* d_x_int, double* const* d_y_int_out, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    if (k < 1 || k > MI_MAX_POWERS) return fail(MI_ERR_UNSUPPORTED, "k must be in 1..MI_MAX_POWERS");
    CHECK_ARG(d_y_int_out, "null output array");
    CHECK_ARG(A->n == A->ncols && !A->mapped, "the internal numbering is defined for square, unmapped matrices");
    if (A->n == 0) return MI_OK;
    CHECK_ARG(d_x_int, "null vector");
    for (int p = 0; p < k; p++) CHECK_ARG(d_y_int_out[p], "null output vector");
    return spmk_unmapped(A->inner ? A->inner : A, k, d_x_int, d_y_int_out, (hipStream_t)s);
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
    dfree(A->kept_x);
    dfree(A->kept_y);
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
    free_sstream(A);
    mi_bcsr4_destroy(A->blocked);
    mi_csr_destroy(A->inner);
    dfree(A->d_iperm);
    dfree(A->d_src_start);
    dfree(A->d_xp);
    for (auto& kv : A->xp_more) dfree(kv.second);
    dfree(A->d_vtmp);
    for (double* p : A->d_pp) dfree(p);
    spmk_release(A);
    delete A;
    return MI_OK;
}

// every value refresh comes by here, on the stream the new CSR values were written on: the copies derived from them follow AT ONCE on
// that stream (the sliced values of spmv_sstream unless the caller filled them in the same pass as the CSR values; the blocked copy and
// its sliced values), so that stream order — and a HIP graph captured earlier — see them like the CSR values themselves
static int refresh_blocked_values(mi_csr_t A, hipStream_t s, bool sliced_done = false)
{
    if (A->ss.dev.val && !sliced_done) {
        SstreamTable& T = A->ss;
        sstream_fill_values(T.rounds, A->n, T.shift, A->d_ptrow, A->d_coef, nullptr, T.dev.slice_step, T.dev.slice_len, T.dev.val, T.max_slice_nnz, s, T.mw ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    if (!A->blocked || A->blocked->nbrows == 0) return MI_OK;
    return bcsr4_refresh_from_csr(A->blocked, A->d_ptrow, A->d_coef, nullptr, s); // blocks and sliced values from the CSR values, one pass
}

// New coefficients for an unchanged sparsity pattern (what a Newton loop does to its Jacobian every
// iteration, src/solve_newton.c:1245-1247): only the value array is replaced.  Row-block tables, the
// ring plan, the 16-bit column stream and the kernel choice depend on the pattern alone and are kept;
// the blocked copy's values are regenerated on the device.
// values of a reordered twin from the caller's (original order) values: new row r' copies its segment
__global__ __launch_bounds__(256) void permute_values_kernel(int n, const int* __restrict__ new_ptrow, const int* __restrict__ src_start,
                                                             const double* __restrict__ src, double* __restrict__ dst)
{
    // (round 5) 64 new rows per workgroup and turn: their destination is ONE contiguous segment, written coalesced; every element finds
    // its row by bisection over the 65 row pointers in LDS and reads from that row's (contiguous) source segment.  The first form — a
    // thread per row walking it — took 2.0-2.8 ms per value update of the relabelled mesh / FE matrices.
    __shared__ int s_p[65], s_a[64];
    const int tid = threadIdx.x;
    for (int r0 = blockIdx.x * 64; r0 < n; r0 += gridDim.x * 64) {
        const int nr = min(64, n - r0);
        if (tid <= nr) s_p[tid] = new_ptrow[r0 + tid];
        if (tid < nr) s_a[tid] = src_start[r0 + tid];
        __syncthreads();
        const int b0 = s_p[0], seg = s_p[nr] - b0;
        for (int k = tid; k < seg; k += 256) {
            const int p = b0 + k;
            int lo = 0, hi = nr; // s_p[lo] <= p < s_p[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_p[mid] <= p) lo = mid;
                else hi = mid;
            }
            dst[p] = src[s_a[lo] + (p - s_p[lo])];
        }
        __syncthreads();
    }
}

extern "C" int mi_csr_update_values_dev(mi_csr_t A, const double* d_coef, mi_stream_t s_)
{
    CHECK_ARG(A, "null handle");
    if (A->nnz == 0) return MI_OK;
    CHECK_ARG(d_coef, "null coef");
    hipStream_t s = (hipStream_t)s_;
    if (A->inner) {
        mi_csr_t I = A->inner;
        hipLaunchKernelGGL(permute_values_kernel, dim3((unsigned)std::max(1, std::min((A->n + 63) / 64, 4096))), dim3(256), 0, s, A->n, I->d_ptrow, A->d_src_start, d_coef, I->d_coef);
        HIP_TRY(hipGetLastError());
        return refresh_blocked_values(I, s);
    }
    if (A->ss.dev.val && d_coef != A->d_coef) { // ONE pass over the caller's values: the CSR values and the sliced values leave together
        SstreamTable& T = A->ss;
        sstream_fill_values(T.rounds, A->n, T.shift, A->d_ptrow, d_coef, A->d_coef, T.dev.slice_step, T.dev.slice_len, T.dev.val, T.max_slice_nnz, s, T.mw ? 1 : 0);
        HIP_TRY(hipGetLastError());
        return refresh_blocked_values(A, s, true);
    }
    if (A->blocked && A->blocked->nbrows > 0 && d_coef != A->d_coef) // the same pass for a blocked copy: CSR values, blocks and sliced values leave together
        return bcsr4_refresh_from_csr(A->blocked, A->d_ptrow, d_coef, A->d_coef, s);
    if (d_coef != A->d_coef) HIP_TRY(hipMemcpyAsync(A->d_coef, d_coef, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToDevice, s));
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

int resolve_kernel(const mi_csr_s* A)
{
    int k = A->kernel != MI_KERNEL_AUTO ? A->kernel : A->auto_kernel;
    if (k == MI_KERNEL_RING && !A->ring.d_plan) k = MI_KERNEL_STREAM; // empty matrix: nothing to plan
    if (k == MI_KERNEL_BCSR4 && !A->blocked) k = MI_KERNEL_STREAM;
    if (k == MI_KERNEL_TILE && !A->tile.d_desc) k = MI_KERNEL_STREAM;
    if (k == MI_KERNEL_MRING && !A->mring.d_plan) k = MI_KERNEL_STREAM;
    if (k == MI_KERNEL_SSTREAM && !A->ss.dev.val) k = A->ring.d_plan && A->ring.ok_fraction >= 0.90 ? MI_KERNEL_RING : MI_KERNEL_STREAM;
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
            const int slot = slots[(size_t)b * c.nnzb + (size_t)ring_slot_pos(T, per, k)];
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

// host-only: the run dependencies of the one-launch powers step exactly as mi_csr_create derives them (ring configuration 4),
// checked against the matrix: every column a run's rows NAME, and every column its lanes LOAD when the plan is replayed (window
// fills, the T lanes behind a block's new columns, the empty blocks behind the run), lies in the rows of a run on its list.
extern "C" int mi_spmk_plan_probe(int n, const int* ptrow, const int* indcol, int* eligible, int* runs, int* max_deps)
{
    CHECK_ARG(n >= 0 && ptrow && ptrow[0] == 0 && eligible, "bad argument");
    CHECK_ARG(ptrow[n] == 0 || indcol, "indcol is null");
    *eligible = 0;
    if (runs) *runs = 0;
    if (max_deps) *max_deps = 0;
    if (n == 0 || ptrow[n] == 0) return MI_OK;
    std::vector<int> row_min((size_t)n), row_max((size_t)n);
    for (int i = 0; i < n; i++) {
        int lo = 0x7fffffff, hi = -1;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k++) {
            CHECK_ARG(indcol[k] >= 0 && indcol[k] < n, "column index outside [0, n)");
            lo = std::min(lo, indcol[k]);
            hi = std::max(hi, indcol[k]);
        }
        row_min[i] = lo;
        row_max[i] = hi;
    }
    RingPlanHost P;
    build_ring_plan(kRingConfigs[3], n, ptrow, row_min.data(), row_max.data(), P);
    bool all_in_loop = P.lean && P.nblk > 0;
    for (int g = 0; g < P.wgs; g++) all_in_loop = all_in_loop && P.run_ok[g] == 1;
    if (!all_in_loop) return MI_OK; // such a handle runs k launches
    std::vector<int> dp, dr;
    build_run_deps(P, n, dp, dr);
    const int W = P.wgs, T = P.cfg.threads;
    std::vector<int> owner((size_t)n, -1);
    for (int g = 0; g < W; g++)
        for (int b = P.run_rng[2 * g]; b < P.run_rng[2 * g + 1]; b++)
            for (int r = P.plan[(size_t)8 * b]; r < P.plan[(size_t)8 * b] + P.plan[(size_t)8 * b + 2]; r++) owner[r] = g;
    for (int i = 0; i < n; i++)
        if (owner[i] < 0) return fail(MI_ERR_STATE, "a row belongs to no run");
    int md = 0;
    std::vector<char> on_list((size_t)W);
    for (int g = 0; g < W; g++) {
        md = std::max(md, dp[g + 1] - dp[g]);
        std::fill(on_list.begin(), on_list.end(), 0);
        for (int j = dp[g]; j < dp[g + 1]; j++) on_list[dr[j]] = 1;
        const int b0 = P.run_rng[2 * g], b1 = P.run_rng[2 * g + 1];
        if (b0 >= b1) continue;
        for (int b = b0; b < b1; b++) {
            const int* Q = &P.plan[(size_t)8 * b];
            for (int k = Q[1]; k < Q[1] + Q[3]; k++)
                if (!on_list[owner[indcol[k]]]) return fail(MI_ERR_STATE, "a run names a column of a run that is not on its list");
            const int load_hi = std::min(n - 1, Q[4] + std::max(Q[5], T) - 1);
            for (int col = Q[4]; col <= load_hi; col++)
                if (!on_list[owner[col]]) return fail(MI_ERR_STATE, "a run loads a column of a run that is not on its list");
        }
    }
    *eligible = md <= 64 ? 1 : 0;
    if (runs) *runs = W;
    if (max_deps) *max_deps = md;
    return MI_OK;
}

extern "C" int mi_csr_sstream_info(mi_csr_t A, int* built, int* rounds, long long* steps, double* padding, double us[4], int* form)
{
    CHECK_ARG(A, "null handle");
    if (A->inner) A = A->inner;
    if (built) *built = A->ss.dev.val != nullptr;
    if (rounds) *rounds = A->ss.rounds;
    if (steps) *steps = A->ss.steps;
    if (padding) *padding = A->ss.padding;
    if (us)
        for (int f = 0; f < 4; f++) us[f] = A->ss.tune_us[f];
    if (form) *form = A->ss.dev.val ? (A->ss.deep ? 2 : 0) + (A->ss.nt ? 0 : 1) : -1;
    return MI_OK;
}

extern "C" int mi_sstream_plan_probe(int n, int ncols, const int* ptrow, const int* indcol, int* eligible, int* rounds, long long* steps, double* padding)
{
    return mi_sstream_plan_probe_ex(n, ncols, ptrow, indcol, 0, 0, 0, eligible, rounds, steps, padding, nullptr, nullptr);
}

extern "C" int mi_sstream_plan_probe_ex(int n, int ncols, const int* ptrow, const int* indcol, int shift, int ghost_lo, int ghost_hi, int* eligible, int* rounds,
                                        long long* steps, double* padding, int* ghost_workgroups, int* rounds_min_max)
{
    CHECK_ARG(n >= 0 && ncols >= 0 && ptrow && ptrow[0] == 0 && eligible, "bad argument");
    CHECK_ARG(shift == 0 || shift == 1, "shift must be 0 or 1");
    CHECK_ARG(ghost_lo >= 0 && ghost_hi <= ncols, "ghost range outside the columns");
    CHECK_ARG(ptrow[n] == 0 || indcol, "indcol is null");
    for (int i = 0; i < n; i++) CHECK_ARG(ptrow[i] <= ptrow[i + 1], "ptrow must be non-decreasing");
    for (int k = 0; k < ptrow[n]; k++) CHECK_ARG(indcol[k] >= 0 && indcol[k] < ncols, "column index outside [0, ncols)");
    SsPlanHost P;
    build_sstream_plan(n, ncols, ptrow, indcol, ss_max_padding(), P, true, shift, ghost_lo, ghost_hi);
    *eligible = P.eligible ? 1 : 0;
    if (rounds) *rounds = P.rounds;
    if (steps) *steps = P.steps;
    if (padding) *padding = ptrow[n] > 0 ? (double)P.pad_places / (double)ptrow[n] : 0.0;
    if (ghost_workgroups) *ghost_workgroups = 0;
    if (rounds_min_max) rounds_min_max[0] = rounds_min_max[1] = 0;
    if (!P.eligible) {
        g_err = std::string("not eligible: ") + P.why;
        return MI_OK;
    }
    if (const char* bad = check_sstream_plan(P, n, ptrow, indcol, ghost_lo, ghost_hi)) return fail(MI_ERR_STATE, std::string("sliced-stream plan: ") + bad);
    int gw = 0, rmin = 0x7fffffff, rmax = 0, rmax_ghost = 0, pmin = 0x7fffffff, pmax = 0;
    for (int g = 0; g < P.nwg; g++) {
        const int c = P.rptr[g + 1] - P.rptr[g];
        gw += P.wg_halo[g];
        rmin = std::min(rmin, c);
        rmax = std::max(rmax, c);
        if (P.wg_halo[g]) rmax_ghost = std::max(rmax_ghost, c);
        else { pmin = std::min(pmin, c); pmax = std::max(pmax, c); }
    }
    // the dealing's promises: the shares of the workgroups that read no ghost differ by at most one round, a ghost-reading workgroup
    // never carries the longest share when there is more than one round per workgroup to deal, and (checked in the replay above) takes
    // all its columns in with its first fill
    if (pmax > 0 && pmax - pmin > 1) return fail(MI_ERR_STATE, "sliced-stream plan: the workgroups' shares differ by more than a round");
    if (gw && gw < P.nwg && rmax > 1 && rmax_ghost >= rmax && P.rounds >= 2 * P.nwg) return fail(MI_ERR_STATE, "sliced-stream plan: a ghost-reading workgroup carries the longest share");
    if (ghost_lo < ghost_hi && !P.fusable) g_err = "eligible, but a ghost-reading workgroup's columns do not fit its first fill: the fused step keeps the ring kernel";
    if (ghost_workgroups) *ghost_workgroups = gw;
    if (rounds_min_max) { rounds_min_max[0] = rmin; rounds_min_max[1] = rmax; }
    return MI_OK;
}

// the cut-ring form's plan (spmv_sstream_mw.hpp) built as mi_csr_create would and replayed against the matrix (host only: no device needed)
extern "C" int mi_sstream_mw_plan_probe(int n, int ncols, const int* ptrow, const int* indcol, int shift, int* eligible, int* rounds, long long* steps, double* padding)
{
    CHECK_ARG(n >= 0 && ncols >= 0 && ptrow && ptrow[0] == 0 && eligible, "bad argument");
    CHECK_ARG(shift == 0 || shift == 1, "shift must be 0 or 1");
    CHECK_ARG(ptrow[n] == 0 || indcol, "indcol is null");
    for (int i = 0; i < n; i++) CHECK_ARG(ptrow[i] <= ptrow[i + 1], "ptrow must be non-decreasing");
    for (int k = 0; k < ptrow[n]; k++) CHECK_ARG(indcol[k] >= 0 && indcol[k] < ncols, "column index outside [0, ncols)");
    SsMwPlanHost P;
    build_sstream_mw_plan(n, ncols, ptrow, indcol, ss_max_padding(), P, shift);
    *eligible = P.eligible ? 1 : 0;
    if (rounds) *rounds = P.rounds;
    if (steps) *steps = P.steps;
    if (padding) *padding = ptrow[n] > 0 ? (double)P.pad_places / (double)ptrow[n] : 0.0;
    if (!P.eligible) {
        g_err = std::string("not eligible: ") + P.why;
        return MI_OK;
    }
    if (const char* bad = check_sstream_mw_plan(P, n, ptrow, indcol)) return fail(MI_ERR_STATE, std::string("cut-ring sliced-stream plan: ") + bad);
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

extern "C" int mi_csr_placement_info(mi_csr_t A, int* n_values, int* n_total, double* us, int cap)
{
    CHECK_ARG(A && cap >= 0 && (cap == 0 || us), "bad argument");
    if (A->inner) A = A->inner;
    if (n_values) *n_values = A->place_draws_coef;
    if (n_total) *n_total = (int)A->place_us.size();
    for (int i = 0; i < cap && i < (int)A->place_us.size(); i++) us[i] = A->place_us[i];
    return MI_OK;
}

// Placement draws for the caller's vectors (include/mi355_spmv.h; profiles/NOTES.md §4.12)
extern "C" int mi_vec_alloc_placed(mi_csr_t A, int nvec, int draws, double** d_vecs, double* us, int cap, int* n_us)
{
    CHECK_ARG(A && nvec >= 1 && d_vecs && cap >= 0 && (cap == 0 || us), "bad argument");
    int rc = need_device();
    if (rc) return rc;
    if (n_us) *n_us = 0;
    const size_t bytes = sizeof(double) * placed_vector_len(A->n, A->ncols);
    const int npairs = (nvec + 1) / 2;
    if (draws > 64) draws = 64;
    const bool timed = draws > npairs && A->nnz >= kLargeNnz && A->n == A->ncols;
    const int cand = timed ? draws : npairs;
    std::vector<double*> xs((size_t)cand, nullptr), ys((size_t)cand, nullptr);
    auto release = [&]() {
        for (double* q : xs) dfree(q);
        for (double* q : ys) dfree(q);
    };
    for (int c = 0; c < cand; c++) { // pairs one after the other: the levels come in windows of consecutive allocations
        hipError_t e = hipSuccess;
        if (c == 0 && A->kept_x && A->kept_y) { // the pair the create-time placement draws were timed on (ownership leaves the handle)
            xs[0] = A->kept_x;
            ys[0] = A->kept_y;
            A->kept_x = A->kept_y = nullptr;
        } else {
            e = hipMalloc(&xs[c], bytes);
            if (e == hipSuccess) e = hipMalloc(&ys[c], bytes);
        }
        // (the kept pair too: its y holds what the draws' products left there)
        if (e == hipSuccess) e = hipMemset(xs[c], 0, bytes);
        if (e == hipSuccess) e = hipMemset(ys[c], 0, bytes);
        if (e != hipSuccess) {
            release();
            return fail(e == hipErrorOutOfMemory ? MI_ERR_ALLOC : MI_ERR_HIP, std::string("mi_vec_alloc_placed: ") + hipGetErrorString(e));
        }
    }
    std::vector<int> order((size_t)cand);
    for (int c = 0; c < cand; c++) order[c] = c;
    if (timed) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        std::vector<double> t((size_t)cand, 0.0);
        bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
        for (int c = 0; c < cand && ok; c++) {
            for (int w = 0; w < 4 && ok; w++) ok = mi_spmv_dev(A, xs[c], ys[c], nullptr) == MI_OK;
            ok = ok && hipEventRecord(e0, nullptr) == hipSuccess;
            for (int w = 0; w < 10 && ok; w++) ok = mi_spmv_dev(A, xs[c], ys[c], nullptr) == MI_OK;
            float ms = 0.f;
            ok = ok && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            t[c] = ms * 1e3 / 10;
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (!ok) {
            release();
            return fail(MI_ERR_HIP, "mi_vec_alloc_placed: timing a candidate pair failed");
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return t[a] < t[b]; });
        for (int c = 0; c < cand && c < cap; c++) us[c] = t[c];
        if (n_us) *n_us = std::min(cand, cap);
        // y = A * 0 left zeros in the y candidates (or NaN-free garbage if A holds infinities): hand out zero-filled vectors
        for (int c = 0; c < cand; c++) (void)hipMemset(ys[c], 0, bytes);
        (void)hipDeviceSynchronize();
    }
    for (int v = 0; v < nvec; v++) { // vector 2k and 2k + 1 = the x and the y of the k-th fastest pair
        double*& slot = (v & 1) ? ys[order[v >> 1]] : xs[order[v >> 1]];
        d_vecs[v] = slot;
        slot = nullptr;
    }
    release(); // everything not handed out
    return MI_OK;
}

extern "C" int mi_vec_free_placed(double* d_vec)
{
    if (d_vec) HIP_TRY(hipFree(d_vec));
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

extern "C" int mi_mring_plan_deal_probe(int n, const int* ptrow, const int* indcol, int* table_len, int* run_blocks, int cap)
{
    CHECK_ARG(n >= 0 && ptrow && ptrow[0] == 0 && table_len && cap >= 0 && (cap == 0 || run_blocks), "bad argument");
    CHECK_ARG(ptrow[n] == 0 || indcol, "indcol is null");
    MringPlanHost P;
    build_mring_plan(n, ptrow, indcol, P);
    *table_len = P.wgs;
    for (int g = 0; g < P.wgs && g < cap; g++) run_blocks[g] = P.run_rng[2 * g + 1] - P.run_rng[2 * g];
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
    CHECK_ARG(kernel_id >= MI_KERNEL_AUTO && kernel_id <= MI_KERNEL_SSTREAM, "unknown kernel id");
    if (kernel_id == MI_KERNEL_SSTREAM && !A->ss.dev.val && A->nnz > 0 && A->d_indcol && !(getenv("MI355_SSTREAM") && !strcmp(getenv("MI355_SSTREAM"), "0"))) {
        // not built at create (a small matrix) or released after losing the create-time measurement: built on request
        int rc = need_device();
        if (rc) return rc;
        std::vector<int> back((size_t)A->nnz);
        HIP_TRY(hipMemcpy(back.data(), A->d_indcol, sizeof(int) * (size_t)A->nnz, hipMemcpyDeviceToHost));
        if ((rc = build_sstream(A, A->h_ptrow.data(), back.data(), A->ghost_lo, A->ghost_hi))) return rc;
        A->ss.asked = true;
        HIP_TRY(hipStreamSynchronize(nullptr)); // (the sliced values were filled on the null stream)
    }
    if (kernel_id == MI_KERNEL_SSTREAM && !A->ss.dev.val)
        return fail(MI_ERR_UNSUPPORTED, "MI_KERNEL_SSTREAM: the handle holds no sliced copy (the matrix pads too much or its rows do "
                                        "not fit the sliding LDS window: mi_sstream_plan_probe says which)");
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
        if ((rc = build_mring(A, nullptr, large_row_align(A->nnz)))) return rc;
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
        static thread_local char nm[128];
        const RingConfig& c = A->ring.cfg;
        snprintf(nm, sizeof nm, "spmv_csr_ring<%d, %d, %d, %d, %d, %s, %s, %s, false, %s, false>", c.threads, c.nnzb, c.ring, c.depth, kRingMaxB,
                 A->d_rowmap ? "true" : "false", A->ring.nt ? "true" : "false", A->ring.skew ? "true" : "false",
                 A->ring.lean && c.threads == 256 && c.depth != 3 ? "true" : "false");
        return nm;
    }
    case MI_KERNEL_ROWPAR: return "spmv_csr_rowpar";
    case MI_KERNEL_BCSR4: {
        const mi_bcsr4_s* B = A->blocked;
        static const char* const sell_names[4] = {"spmv_bcsr4_sell<8, true, 0, 2, 4>", "spmv_bcsr4_sell<8, false, 0, 2, 4>", "spmv_bcsr4_sell<4, true, 0, 2, 8>", "spmv_bcsr4_sell<12, true, 0, 2, 4>"};
        if (B && B->sell_form >= 0 && B->d_sell_val) return sell_names[B->sell_form & 3];
        return B && B->use_tile && B->d_tl_ptr ? "spmv_bcsr4_tile<2>" : "spmv_bcsr4<2>";
    }
    case MI_KERNEL_MRING: {
        static thread_local char nm[96];
        snprintf(nm, sizeof nm, "spmv_csr_mring<%d, %d, %d, %d, %s, %s, %s>", kMringThreads, kMringNnzb, A->mring.depth, kMringMaxB,
                 A->d_rowmap ? "true" : "false", A->mring.nt ? "true" : "false", A->mring.skew ? "true" : "false");
        return nm;
    }
    case MI_KERNEL_SSTREAM: {
        static thread_local char nm[64];
        if (A->ss.mw) snprintf(nm, sizeof nm, "spmv_sstream_mw<%d, %s>", A->ss.deep ? 12 : 8, A->ss.nt ? "true" : "false");
        else snprintf(nm, sizeof nm, "spmv_sstream<%d, %s, 0>", A->ss.deep ? 12 : 8, A->ss.nt ? "true" : "false");
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
    if (A->n == 0) return MI_OK;
    for (int p = 0; p < k; p++) CHECK_ARG(d_y_out[p], "null output vector");
    if (A->inner) {
        // the whole chain in the new numbering (each power feeds the next without leaving it), then every power scattered to
        // the caller's numbering
        while ((int)A->d_pp.size() < k) {
            double* p = nullptr;
            HIP_TRY(hipMalloc(&p, sizeof(double) * (size_t)A->n));
            A->d_pp.push_back(p);
        }
        double* xp = nullptr;
        int rc = reorder_scratch(A, (hipStream_t)s, &xp);
        if (rc) return rc;
        if ((rc = gather_perm(A, d_x, xp, (hipStream_t)s))) return rc;
        if ((rc = spmk_unmapped(A->inner, k, xp, A->d_pp.data(), (hipStream_t)s))) return rc;
        for (int p = 0; p < k; p++)
            if ((rc = scatter_perm(A, A->d_pp[p], d_y_out[p], (hipStream_t)s))) return rc;
        return MI_OK;
    }
    return spmk_unmapped(A, k, d_x, d_y_out, (hipStream_t)s);
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
