// launch_csr.hip: instantiations of the CSR kernels and launch_spmv — part of libmi355spmv.so (see capi_internal.hpp for the layout of the library).
// Built for gfx950 only; no CPU fallback anywhere: every compute entry point needs a HIP device.
#include "capi_internal.hpp"
#include "spmv_rowpar.hpp"
#include "spmv_tile.hpp"
#include "spmv_mring.hpp"

// ---------------------------------------------------------------- SpMV launch
int reorder_scratch(mi_csr_t A, hipStream_t s, double** buf)
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (!A->xp_claimed || A->xp_stream == s) {
        A->xp_claimed = true;
        A->xp_stream = s;
        *buf = A->d_xp;
        return MI_OK;
    }
    double*& b = A->xp_more[s];
    if (!b) {
        if (stream_is_capturing(s)) return fail(MI_ERR_STATE, "first product of a relabelled handle on this stream: run one outside stream capture first (it allocates a gather buffer)");
        HIP_TRY(hipMalloc(&b, sizeof(double) * (size_t)A->n));
    }
    *buf = b;
    return MI_OK;
}

bool ring_dot_eligible(const mi_csr_s* A)
{
    if (A->inner || A->n == 0 || A->d_rowmap || A->y_offset) return false;
    if (resolve_kernel(A) != MI_KERNEL_RING) return false;
    const RingTable& R = A->ring;
    int depth = R.cfg.depth;
    if (const char* e = getenv("MI355_RING_DEPTH")) depth = atoi(e);
    return R.cfg.id == 4 && R.lean && depth != 3 && R.all_in_loop && R.wgs >= 1 && R.wgs <= 1024 /* kMaxPartials */ &&
           !(getenv("MI355_SPMV_DOT_EPILOGUE") && !strcmp(getenv("MI355_SPMV_DOT_EPILOGUE"), "0"));
}

int launch_spmv(mi_csr_t A, const double* d_x, double* d_y, hipStream_t s, bool use_map, const RingComm* comm, const RingDot* dot)
{
    if (dot && !ring_dot_eligible(A)) return fail(MI_ERR_STATE, "dot epilogue requested on a handle whose launch cannot carry it");
    if (A->n == 0) return MI_OK;
    // The fused multi-GPU step hands over a RingComm: its piece is numbered [ghosts | owned | ghosts], which only the ring
    // kernel's FUSED instantiation understands.  Any other launch would index x with that numbering — refuse, never drop it.
    const bool ss_fused = comm && resolve_kernel(A) == MI_KERNEL_SSTREAM && A->ss.fusable;
    if (comm && (A->inner || A->d_rowmap || (resolve_kernel(A) != MI_KERNEL_RING && !ss_fused)))
        return fail(MI_ERR_STATE, "fused multi-GPU step: the combined piece must be an unmapped, unreordered handle served by the ring or the sliced-stream kernel");
    if (A->inner) { // reordered: x into the new numbering, then the twin writes y through its row map
        double* xp = nullptr;
        int rc = reorder_scratch(A, s, &xp);
        if (rc) return rc;
        if ((rc = gather_perm(A, d_x, xp, s))) return rc;
        // (y through the twin's row map: scattered 8-byte stores.  Round 4 measured the alternative — the twin writes its own numbering,
        // whole lines, and a gather y[i] = scratch[perm[i]] follows: c2_perm 54.7 -> 59.6 us, fe_perm 123 -> 129, mesh_perm 304 -> 359.)
        return launch_spmv(A->inner, xp, d_y, s, true);
    }
    int kid = resolve_kernel(A);
    if (kid == MI_KERNEL_SSTREAM && !dot && (!comm || ss_fused)) {
        const int* map = use_map ? A->d_rowmap : nullptr;
        double* yy = d_y + (use_map ? A->y_offset : 0);
        if (sstream_y_ok(A, yy, map)) return launch_sstream(A, d_x, yy, map, s, comm);
        if (comm) return fail(MI_ERR_ARG, "fused multi-GPU step: y must be 16-byte aligned");
        kid = A->ring.d_plan && A->ring.ok_fraction >= 0.90 ? MI_KERNEL_RING : MI_KERNEL_STREAM; // y's row pairs are not 16-byte aligned: the sliced kernel stores them whole
    } else if (kid == MI_KERNEL_SSTREAM) {
        kid = A->ring.d_plan && A->ring.ok_fraction >= 0.90 ? MI_KERNEL_RING : MI_KERNEL_STREAM; // (the dot epilogue lives in the ring kernel; the sliced stream's own epilogue measured
        // slower than product + separate dot: profiles/NOTES.md R4.4)
    }
    if (use_map) d_y += A->y_offset;
    CsrView V{};
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
#define MRING_L(D_, MP_, NT_, SK_) hipLaunchKernelGGL((spmv_csr_mring<kMringThreads, kMringNnzb, D_, kMringMaxB, MP_, NT_, SK_>), dim3(M.wgs), dim3(kMringThreads), 0, s, V, plan, reinterpret_cast<const int4*>(M.d_first), M.d_ok, M.d_slots, d_x, d_y, rng, M.wgs)
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
        launch_ring_cfg(A, V, d_x, d_y, s, comm, dot); // launch_ring.hip
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
