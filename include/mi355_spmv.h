/*
 * mi355_spmv.h — the C-ABI boundary of the MI355X-native CSR SpMV /
 * matrix-powers path (libmi355spmv.so).
 *
 * Plain C: opaque handles, raw pointers and sizes, int status codes.  These are
 * the entry points a binding of the reference's hot path attaches to; the
 * reference interface each one stands behind is cited as file:line relative to
 * aantoine890/navierstokes.  The C++ drop-in with the reference's exact
 * mpk/SpMV.h names and signatures (include/SpMV.h, libmpk_mi355.so) is a thin
 * layer over this header; INTEGRATION.md shows both bindings.
 *
 * Conventions
 *   - every function returns MI_OK (0) or an MI_ERR_* code; mi_last_error()
 *     gives the detail of the last failure on the calling thread.  The
 *     reference's functions are void and unchecked (mpk/SpMV.h:52-66); the
 *     C++ shim aborts with the message on a non-zero status.
 *   - indices are 0-based int32, values are IEEE double (mpk/SpMV.h:18-24).
 *   - the caller owns every buffer it passes; the library copies the matrix to
 *     the device at create time and never retains caller memory.  Output
 *     vectors are fully overwritten (as mpk/SpMV.cpp:13, mpk/SpM2V.cpp:85-86).
 *   - "*_dev" entry points take DEVICE pointers (HBM-resident vectors) and a
 *     hipStream_t passed as void*; they enqueue work and return without
 *     synchronising.  The others take HOST pointers, copy in/out and
 *     synchronise — the semantics of the reference's CPU functions.
 *   - arithmetic: each row of y = A x is ONE sequential fma chain in CSR order,
 *     bit-identical to the reference's SpMV_CSR_OPT / SpMV_CSR_FMA
 *     (mpk/SpMV.cpp:23-56) for every matrix and every kernel id.
 *   - there is no CPU fallback: without a usable HIP device the compute entry
 *     points fail with MI_ERR_NODEVICE.
 */
#ifndef MI355_SPMV_H
#define MI355_SPMV_H

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_SPMV_VERSION 501 /* 0.5.1: mi_bcsr4_spmm_info writes us[5] since 0.4 (it was us[4] in 0.3: a caller built against 0.3 must be rebuilt); 0.5 adds
                                * mi_sstream_plan_probe_ex, mi_part_kernel_name, mi_part_sends_contiguous and refills sliced copies inside mi_*_update_values*;
                                * 0.5.1 adds mi_sstream_mw_plan_probe (nothing changed for a caller built against 0.5.0) */

enum {
    MI_OK = 0,
    MI_ERR_ARG = 1,         /* null pointer, negative size, inconsistent CSR */
    MI_ERR_NODEVICE = 2,    /* no HIP device / driver */
    MI_ERR_HIP = 3,         /* a HIP runtime call failed (see mi_last_error) */
    MI_ERR_ALLOC = 4,       /* host or device allocation failed */
    MI_ERR_UNSUPPORTED = 5, /* e.g. k outside 1..MI_MAX_POWERS */
    MI_ERR_STATE = 6        /* handle used before it was finalised / wrong device */
};

#define MI_MAX_POWERS 16

typedef struct mi_csr_s* mi_csr_t;   /* device-resident CSR matrix (struct csrmatrix, mpk/SpMV.h:18-24) */
typedef struct mi_bcsr4_s* mi_bcsr4_t; /* device-resident 4x4 BCSR matrix (struct bcsr4x4_matrix, mpk/SpMV.h:26-33) */
typedef struct mi_part_s* mi_part_t; /* one rank's share of a row-partitioned matrix */
typedef void* mi_stream_t;           /* hipStream_t; NULL = the default stream */

/* kernel ids for mi_csr_set_kernel (all produce the same bits) */
enum {
    MI_KERNEL_AUTO = 0,    /* the fastest eligible kernel, measured on the handle at mi_csr_create (mi_csr_tune_detail) */
    MI_KERNEL_STREAM = 1,  /* row-block CSR-stream, x gathered through L1/L2 (any matrix) */
    MI_KERNEL_RING = 2,    /* persistent workgroups, sliding x window in LDS, pipelined matrix stream */
    MI_KERNEL_ROWPAR = 3,  /* one thread per row straight from global memory (reference shape; slow) */
    MI_KERNEL_BCSR4 = 4,   /* the BCSR 4x4 kernel on a blocked copy made at mi_csr_create; available only when the
                            * CSR matrix has exact 4x4 node-block structure (FE matrices), where it returns the
                            * same bits from 8.25 instead of 12 matrix bytes per nonzero */
    MI_KERNEL_MRING = 6,   /* the ring kernel with five independent sliding windows: 3-D mesh operators, whose rows reach into a
                            * few narrow column clusters whole mesh planes apart (any mesh size) */
    MI_KERNEL_SSTREAM = 7, /* (round 4) the matrix as ONE contiguous stream per wave: a sliced copy (128-row slices, 16 bytes of values per
                            * lane and step, a 32-bit word of LDS slots and flags), a lane per row pair, x in an LDS ring that slides with
                            * the rows, y parked in LDS — banded matrices with near-uniform row lengths (spmv_sstream.hpp) */
    MI_KERNEL_TILE = 5     /* per row block the DISTINCT columns are gathered once into an LDS tile, nonzeros address it
                            * through a 16-bit stream: wide-band matrices whose neighbouring rows share columns (P1
                            * operators on unstructured 3-D meshes), which the ring's contiguous window cannot hold */
};

/* ---- library / device ------------------------------------------------- */
int mi_version(void);
const char* mi_strerror(int status);
const char* mi_last_error(void);
int mi_device_count(int* count);
int mi_set_device(int device);
int mi_device_synchronize(void);
/* Evict the GPU's L2s and 256 MiB Infinity Cache: the device analogue of flush_cache(), mpk/utils.cpp:146-154, which the
 * reference calls before every timed kernel.  A 512 MiB device fill (the reference writes its buffer too) followed by a
 * 512 MiB READ sweep of a second buffer and a synchronise: the fill alone leaves the Infinity Cache full of dirty lines
 * whose write-back the next kernel pays for (≈11 us on a 1 GB product); behind the read sweep the caches hold clean lines
 * of a buffer nobody uses.  MI355_FLUSH_FILL_ONLY=1 restores the fill-only form. */
int mi_flush_cache(void);
/* The same eviction enqueued on stream s WITHOUT the closing synchronise: a product enqueued right behind it starts on cold
 * caches but on a GPU that never went idle.  (Behind the synchronous form the device idles until the host has returned and
 * launched; a kernel started on an idle MI355X measures 8-26 us longer than the same kernel behind another one, whatever
 * the caches hold — tools/cold_probe.py.  The reference's CPU protocol has no such effect to separate.) */
int mi_flush_cache_async(mi_stream_t s);

/* ---- CSR matrix handles ------------------------------------------------ */
/* Upload a csrmatrix (mpk/SpMV.h:18-24: n, ptrow[n+1], indcol, coef; nnz taken
 * from ptrow[n], not from the possibly stale csrmatrix::nnz — mpk/utils.cpp:100).
 * ncols = length of the x vectors (n for the reference's square matrices).
 * Host pointers. */
int mi_csr_create(int n, int ncols, const int* ptrow, const int* indcol, const double* coef, mi_csr_t* out);
/* Same, for a row subset whose results are scattered: row r of this matrix
 * writes y[rowmap[r]] (rowmap == NULL: y[r]).  Used for the interior/boundary
 * split of a partitioned matrix. */
int mi_csr_create_mapped(int n, int ncols, const int* ptrow, const int* indcol, const double* coef,
                         const int* rowmap, mi_csr_t* out);
int mi_csr_destroy(mi_csr_t A);
/* New coefficients for an UNCHANGED sparsity pattern: coef has the nnz values in the order of the
 * arrays given to mi_csr_create.  The reference's functions read the caller's live arrays on every
 * call; a handle holds a device copy, so a caller that rewrites its coefficients in place — the
 * Newton loop does that to the Jacobian every iteration, src/solve_newton.c:1245-1247 (MatCopy +
 * add_nonlinear_jacobian_terms + MatZeroRows), before KSPSolve :1265 — refreshes the copy with this
 * call: one transfer of the values; row blocks, ring plan, column stream and kernel choice depend on
 * the pattern only and are kept, the blocked copy's values are regenerated on the device.  The
 * mpk/SpMV.h shim (libmpk_mi355.so) detects such changes by itself (full content hash) and calls this. */
int mi_csr_update_values(mi_csr_t A, const double* coef);                          /* host values; synchronous */
int mi_csr_update_values_dev(mi_csr_t A, const double* d_coef, mi_stream_t s);     /* device values; asynchronous on s */
int mi_csr_dims(mi_csr_t A, int* n, int* ncols, long long* nnz);
/* Locality reordering.  The reference's matrices come from unstructured gmsh meshes (src/solve_newton.c:91-197,
 * src/benchmark_spmv.c:76-123) whose node numbering scatters a row's columns over all of x.  For a square matrix
 * that is not already served by the ring kernel and whose nonzeros lie far from the diagonal for its size,
 * mi_csr_create computes a reverse Cuthill-McKee relabelling of the NODE graph (4x4 node blocks stay together),
 * builds A' = P A P^T with every row's nonzeros in the caller's order, and keeps it if it measures faster.  The
 * permutation never shows: x is gathered into the new numbering before a product and y is written back through a
 * row map, and since a row's terms keep their order every bit of y equals the unreordered kernels' (and the
 * reference's SpMV_CSR_FMA).  A reordered handle keeps one gather buffer per stream it has multiplied on (products of one handle on
 * different streams do not share scratch; mi_spmk_dev's power buffers are per handle: one k-step at a time).
 * MI355_REORDER=0 disables, =1 forces.  *reordered = 1 if the handle computes through the relabelled twin; block = 4
 * if nodes of four rows were moved, 1 if single rows; spread = mean |column - row| in nodes before / after;
 * us_* = measured launch time of the natural-order choice and of the twin incl. its gather (0 if not measured). */
int mi_csr_reorder_info(mi_csr_t A, int* reordered, int* block, double* spread_before, double* spread_after,
                        double* us_natural, double* us_reordered);
/* ---- staying in the library's numbering (callers that own the loop) -----------------------------------------------
 * A relabelled handle pays an x gather and a mapped y store on EVERY mi_spmv_dev (random 8-byte accesses: 92 + 45 us of
 * 326 us at 5 M rows).  A caller that runs many products per matrix — a Krylov solve: src/solve_newton.c:1265 KSPSolve, one
 * MatMult per GMRES iteration on the mesh numbering of :91-197 — permutes its vectors ONCE per solve instead:
 *   mi_csr_perm              perm[old] = new (identity and *reordered = 0 for a handle that was not relabelled)
 *   mi_vec_to_internal_dev   x_int[perm[i]] = x[i]     (one gather; out of place)
 *   mi_vec_from_internal_dev x[i] = x_int[perm[i]]
 *   mi_spmv_internal_dev     y_int = A' x_int: the twin alone — no gather, no row map, no per-handle scratch
 *   mi_spmk_internal_dev     the powers chain, every power left in the internal numbering
 * The BLAS-1 entry points (mi_dot_dev, mi_axpy_dev, mi_orthogonalize_dev, mi_norm2_dev, ...) are numbering-agnostic:
 * element-wise updates commute with a permutation bit for bit; a reduction sums in index order of whatever numbering it is
 * given, so its last bits may differ between numberings (inside the bound documented at mi_dot).  Every row of y_int is the
 * same fma chain as the caller-numbering product's: un-permuting y_int gives mi_spmv_dev's bits. */
int mi_csr_perm(mi_csr_t A, int* reordered, int* perm /* [n] or NULL */);
int mi_vec_to_internal_dev(mi_csr_t A, const double* d_x, double* d_x_int, mi_stream_t s);
int mi_vec_from_internal_dev(mi_csr_t A, const double* d_x_int, double* d_x, mi_stream_t s);
int mi_spmv_internal_dev(mi_csr_t A, const double* d_x_int, double* d_y_int, mi_stream_t s);
int mi_spmk_internal_dev(mi_csr_t A, int k, const double* d_x_int, double* const* d_y_int_out, mi_stream_t s);
/* host-only: the relabelling mi_csr_create would compute; perm[old] = new (n entries) */
int mi_reorder_probe(int n, const int* ptrow, const int* indcol, int* block, int* perm, double* spread_before,
                     double* spread_after);
/* MI_KERNEL_RING on a matrix whose window does not fit is still correct (its runs take the
 * per-block path); mi_csr_ring_info reports how much of the matrix the ring serves. */
int mi_csr_set_kernel(mi_csr_t A, int kernel_id);
int mi_csr_ring_info(mi_csr_t A, int* config_id, int* runs, int* runs_not_ringable, double* nnz_fraction_ringable);
/* Shape of the ring plan in use: row blocks, whether the LEAN instantiation runs, blocks of prefetch.  Matrices of >= 20 M
 * nonzeros take row blocks cut at the nonzero count, smaller ones blocks ending on multiples of 64 rows (the first shape's
 * rate does not depend on where the caller's x and y lie in device memory, the second's does: profiles/NOTES.md §4.1).  With
 * MI355_RING_SHAPE_COMPARE=1 in the environment mi_csr_create also times the other shape on its scratch vectors
 * (microseconds per launch; 0 = not compared). */
int mi_csr_ring_shape_info(mi_csr_t A, int* blocks, int* lean, int* depth, double* us_aligned, double* us_unaligned);
/* MI_KERNEL_AUTO is decided by measurement: mi_csr_create times the candidate kernels (ring if
 * >= 90 % of the nonzeros are ring-served, stream, tile if a plan was kept, BCSR 4x4 if a blocked copy exists) on the new
 * handle, two interleaved rounds of a few launches each, and keeps the fastest.  All kernels produce
 * the same bits, so the choice never changes a result.  Reports the measured microseconds per launch
 * of the chosen temporal / non-temporal form (0 = candidate not eligible / not timed).
 * MI355_SPMV_KERNEL=ring|stream|rowpar|bcsr4|tile|mring or MI355_SPMV_AUTOTUNE=0 skip the measurement (then: ring
 * if eligible else stream, BCSR 4x4 if blocked; non-temporal loads for matrices beyond the Infinity Cache). */
int mi_csr_tune_info(mi_csr_t A, double* us_ring, double* us_stream);
/* Each candidate is timed twice, with temporal and with non-temporal loads of the matrix (a matrix
 * that fits the 256 MB Infinity Cache is faster temporal across repeated products, a larger one
 * non-temporal); us[0..4] = ring, ring non-temporal, stream, stream non-temporal, BCSR 4x4 on the
 * blocked copy (0 = not eligible / not timed); *ring_nt / *stream_nt = 1 if that kernel of this
 * handle uses non-temporal loads.  MI355_RING_NT / MI355_STREAM_NT = 0|1 force the choice,
 * MI355_AUTO_BCSR=0 disables the blocked copy. */
int mi_csr_tune_detail(mi_csr_t A, double us[5], int* ring_nt, int* stream_nt);
/* host-only: build the ring kernel's window plan and 16-bit column stream for configuration config_id (1..4,
 * ring_plan.hpp) exactly as mi_csr_create would, verify their invariants (MI_ERR_STATE names the first
 * violation) and report what the ring would serve.  Lets the planner be tested without a GPU. */
int mi_ring_plan_probe(int n, const int* ptrow, const int* indcol, int config_id, int* nblk, int* runs,
                       int* runs_not_ringable, double* nnz_fraction_ringable, int* max_slot);
/* Tile kernel (MI_KERNEL_TILE): mi_csr_create builds its plan for matrices the ring does not serve and keeps it when the
 * row blocks average at most 0.6 distinct columns per nonzero (MI355_TILE=0 never, =1 always); mi_csr_set_kernel(A,
 * MI_KERNEL_TILE) builds it on request.  *built = 1 if the handle holds a plan; unique_per_nnz = distinct columns per
 * nonzero over all row blocks; us[0..1] = measured microseconds per launch with temporal / non-temporal value loads
 * (0 = not timed); *nt = 1 if the handle's tile kernel uses non-temporal loads (MI355_TILE_NT=0|1 forces). */
int mi_csr_tile_info(mi_csr_t A, int* built, int* nblk, double* unique_per_nnz, double us[2], int* nt);
/* host-only: build the tile kernel's plan (tile_plan.hpp) exactly as mi_csr_create would, verify its invariants
 * (MI_ERR_STATE names the first violation: every slot names its nonzero's column, lists strictly ascending, 16-byte
 * aligned slot segments, over-long rows unlisted) and report its size.  threads = 0: as many as the library would use. */
int mi_tile_plan_probe(int n, const int* ptrow, const int* indcol, int threads, int* nblk, long long* distinct_total,
                       int* max_distinct, long long* nnz_listed);
/* Placement draws (profiles/NOTES.md §4.12): for matrices beyond the caches (>= 20 M nonzeros, a CSR kernel chosen) mi_csr_create times the
 * chosen kernel on a few fresh device copies of the value array, then of the 16-bit column stream, and keeps the fastest copy of
 * each — where these arrays lie in device memory moves a warm launch by up to 15 %.  us[0 .. *n_values) = microseconds per launch
 * with the value array as first allocated ([0]) and after each draw; us[*n_values .. *n_total) the same for the column stream
 * (cap = length of us; both counts 0: no draws were made).  MI355_PLACEMENT_DRAWS=0 turns the draws off, =N sets their number
 * (default 12 for the value array, capped so that the copies fit an eighth of the free device memory; half as many for the stream). */
int mi_csr_placement_info(mi_csr_t A, int* n_values, int* n_total, double* us, int cap);
/* The same for the CALLER's vectors (round 3, profiles/NOTES.md §4.12): on some MI355X boxes a product with A runs 126 or 141-143 us by which
 * physical memory its x and y were handed — in windows that follow the order of allocation — whatever the kernel does.  A solver that
 * keeps its vectors for many products (src/solve_newton.c:1265: one KSPSolve, hundreds of MatMults) can let the library place them:
 * mi_vec_alloc_placed allocates nvec device vectors of max(rows, columns) doubles each (zero-filled, 256-byte aligned) for use with A,
 * by allocating `draws` candidate PAIRS one after the other (the first: the scratch pair mi_csr_create's own placement draws were timed
 * on, kept by the handle for this call), timing y = A x on each pair (a few launches on stream 0, synchronous),
 * keeping the nvec vectors of the fastest pairs and freeing the others.  draws <= 1 (or a matrix of < 20 M nonzeros, whose products
 * live in the caches): plain allocations, nothing timed.  us[0 .. *n_us) = microseconds per launch of each candidate pair in the
 * order drawn (cap = length of us).  Vectors are released with mi_vec_free_placed (any order, any time after the handle's last use).
 * Results do not depend on where a vector lies: placement is a matter of speed only. */
int mi_vec_alloc_placed(mi_csr_t A, int nvec, int draws, double** d_vecs, double* us, int cap, int* n_us);
int mi_vec_free_placed(double* d_vec);
/* Sliced-stream kernel (MI_KERNEL_SSTREAM, spmv_sstream.hpp): mi_csr_create plans it for unmapped matrices of >= 200 000 nonzeros whose
 * rounds of 512 rows fit an 8192-column LDS window that slides forward by at most 1024 columns per round and whose 128-row slices pad by
 * at most 12 % (MI355_SSTREAM=0 never, =1 plan whatever the size), builds the sliced copy (10 bytes per nonzero beside the CSR arrays)
 * and times it — 8 / 12 steps of prefetch, non-temporal / temporal value loads — against the other candidates.  *built = 1 if the handle
 * holds the copy; *padding = padded places per nonzero; us[0..3] = microseconds per launch for D = 8 nt, D = 8 temporal, D = 12 nt,
 * D = 12 temporal (0 = not timed); *form = the variant in use (index into us).  y must be 16-byte aligned (else the handle's next-best
 * kernel runs that product).  A copy that loses the create-time measurement is released again (mi_csr_set_kernel(MI_KERNEL_SSTREAM) rebuilds
 * it on request); the sliced values follow every mi_csr_update_values* at once, on that call's stream (HIP-graph replays included).
 * A matrix whose rows name SEVERAL column neighbourhoods further apart than that window holds — a 3-D mesh operator in natural node
 * order: a node's plane and the two next to it — gets the CUT-RING form of the same kernel (spmv_sstream_mw.hpp; mi_csr_kernel_name says
 * spmv_sstream_mw<...>): four sub-rings of 2048 columns, each following one neighbourhood; eligible with at most four neighbourhoods per
 * 512-row round (gaps of more than 512 columns separate them), none wider than 2048 columns.  Same id, same info call; any y alignment.
 * MI355_SSTREAM_MW=0 disables the form. */
int mi_csr_sstream_info(mi_csr_t A, int* built, int* rounds, long long* steps, double* padding, double us[4], int* form);
/* host-only: build that plan exactly as mi_csr_create would and REPLAY it against the matrix (MI_ERR_STATE names the first violation:
 * every nonzero's slot is its column's ring slot and the column lies inside the window when its round runs; padding places are flagged).
 * *eligible = 0 with the reason in mi_last_error() when the matrix does not qualify. */
int mi_sstream_plan_probe(int n, int ncols, const int* ptrow, const int* indcol, int* eligible, int* rounds, long long* steps, double* padding);
/* the same for the two forms round 5 added: shift = 1 plans the rows one down (what mi_csr_create_mapped does for a piece whose rows go to
 * y[r + odd offset]: row pairs stay 16-byte aligned); ghost_lo < ghost_hi: columns outside [ghost_lo, ghost_hi) are ghosts (a partition's
 * combined piece, numbered [lower ghosts | owned | upper ghosts]): a workgroup whose windows hold one is marked (*ghost_workgroups counts
 * them; the fused multi-GPU step makes them push and wait first), takes its whole column range in with its first fill and gets three rounds
 * less than its share (fewer still where that range would not fit the 8192-column ring).  rounds_min_max[2] = the
 * shortest and the longest share of rounds.  The replay also checks the marks and the dealing. */
int mi_sstream_plan_probe_ex(int n, int ncols, const int* ptrow, const int* indcol, int shift, int ghost_lo, int ghost_hi, int* eligible, int* rounds,
                             long long* steps, double* padding, int* ghost_workgroups, int* rounds_min_max);
/* ... and of its cut-ring form for rows that name several column neighbourhoods (3-D mesh operators in natural node order;
 * navierstokes_amd/csrc/spmv_sstream_mw.hpp): built as mi_csr_create builds it when the one-window plan is not eligible, and replayed —
 * every nonzero must find its column at its slot of the cut ring when its round runs.  Host only. */
int mi_sstream_mw_plan_probe(int n, int ncols, const int* ptrow, const int* indcol, int shift, int* eligible, int* rounds, long long* steps, double* padding);
/* Multi-window ring kernel (MI_KERNEL_MRING, mring_plan.hpp): mi_csr_create plans it for matrices the single ring does not serve
 * and keeps the plan when it serves >= 90 % of the nonzeros (MI355_MRING=0 never, =1 always keep); mi_csr_set_kernel builds it
 * on request.  us[0..1] = measured microseconds per launch, temporal / non-temporal value loads (MI355_MRING_NT=0|1 forces). */
int mi_csr_mring_info(mi_csr_t A, int* built, int* runs, int* runs_not_served, double* nnz_fraction_served, double us[2], int* nt);
/* host-only: build that plan exactly as mi_csr_create would and REPLAY it (MI_ERR_STATE names the first violation: every
 * nonzero's 16-bit slot must hold its column when its block runs, runs cover every block once, records are consistent). */
int mi_mring_plan_probe(int n, const int* ptrow, const int* indcol, int* nblk, int* runs, int* runs_not_served,
                        double* nnz_fraction_served, long long* window_restarts);
/* host-only: the DISPATCH ORDER of that plan's runs (mring_plan.hpp: workgroup i of the grid goes to XCD i % 8, an XCD takes its
 * workgroups in order, 64 resident at a time): *table_len = the kernel's grid = 8 * per_xcd; run_blocks[x * per_xcd + j] = blocks of
 * the run the j-th workgroup of XCD x executes (0: none).  Long runs come first in every XCD's share, the short ones behind them. */
int mi_mring_plan_deal_probe(int n, const int* ptrow, const int* indcol, int* table_len, int* run_blocks, int cap);
/* host-only: would the ring plan of this pattern (configuration config_id) run the LEAN instantiation of the kernel — no block
 * inside a run bringing more than T new columns, none holding more than T rows (spmv_ring.hpp)?  The planner cuts its runs
 * at window restarts to make it so; only configuration 4 has the instantiation. */
int mi_ring_plan_lean(int n, const int* ptrow, const int* indcol, int config_id, int* lean);
/* host-only: does this CSR pattern have the exact 4x4 node-block structure mi_csr_create looks for (n % 4 == 0,
 * the four rows of a block row hold the same columns, in aligned groups {4j..4j+3}) — i.e. will a blocked copy be
 * built and the BCSR kernel become an AUTO candidate?  *nblocks = number of 4x4 blocks if so. */
int mi_csr_block4_structure(int n, const int* ptrow, const int* indcol, int* is_blocked, long long* nblocks);
/* override the measured choice: 1 = non-temporal matrix loads, 0 = temporal, -1 = leave as is */
int mi_csr_set_nontemporal(mi_csr_t A, int ring_nt, int stream_nt);
/* calibration: microseconds per launch of a plain non-temporal read sweep over `bytes` of device memory (16-byte
 * loads, 2048 workgroups, back to back): the rate THIS GPU streams from HBM at.  MI355X boxes of one pool differ by 10-20 %
 * here; bench.py prints it beside the kernel's rate so that a roofline fraction can be read against the box it was taken on. */
int mi_stream_read_probe(long long bytes, int launches, double* us_per_launch);
int mi_csr_get_kernel(mi_csr_t A, int* kernel_id);
/* name of the HIP kernel the next mi_spmv*(A) launches (for matching rocprof rows) */
const char* mi_csr_kernel_name(mi_csr_t A);

/* ---- SpMV: y = A x  (SpMV_CSR{,_OPT,_FMA,_AVX2}, mpk/SpMV.cpp:6-85) ---- */
int mi_spmv(mi_csr_t A, const double* x, double* y);                           /* host vectors */
int mi_spmv_dev(mi_csr_t A, const double* d_x, double* d_y, mi_stream_t s);    /* device vectors */

/* ---- matrix powers: y_out[p] = A^(p+1) x, p = 0..k-1 --------------------
 * SpM2V_CSR (mpk/SpM2V.cpp:79-112: y_out[0]=y, y_out[1]=z), SpM3V / SpM4V
 * (mpk/SpMVmulti0.cpp:132-155, :189-221: all intermediate powers returned).
 * Unlike the CPU first-touch traversal, rows of A^p x that no row references
 * as a column are computed too (SURVEY.md §8a-10 caveat).
 * mi_spmk_dev is asynchronous on s — EXCEPT the first k-step of a handle at a given k (2 <= k <= 8) on an eligible ring-served
 * matrix: that call times the one-launch form against k launches (a few dozen launches on s, then a stream synchronise) and keeps
 * the faster (mi_csr_spmk_info says which).  Under stream capture nothing is measured and k plain launches are recorded. */
int mi_spmk(mi_csr_t A, int k, const double* x, double* const* y_out);                      /* host */
int mi_spmk_dev(mi_csr_t A, int k, const double* d_x, double* const* d_y_out, mi_stream_t s); /* device; d_y_out is a HOST array of k device pointers */

/* The powers step runs as ONE launch where that is possible and faster (spmk_ring.hpp: every persistent workgroup keeps its run
 * of row blocks for all k powers; a run's power p is published write-through + flag, power p + 1 waits for the flags of the runs
 * its columns name) — k <= 8, ring-served square matrices whose whole grid is resident at once; the first k-step of a handle at a
 * given k times both forms (same bits) and keeps the faster.  MI355_SPMK_FUSED=0 never, =1 always where eligible.  The handle
 * carries the step's flags: one k-step at a time per handle.  That FIRST k-step at a given k is therefore synchronous: it runs
 * 2 rounds x 2 forms x (2 warm-up + 5 timed) extra k-steps on the caller's stream and vectors and blocks the host on an event (a few
 * milliseconds at 1 M rows); later calls are asynchronous as documented.  If a wait of the one-launch form gives up during that
 * measurement, the measurement stops at once and the handle runs k launches.  Under HIP stream capture the step is recorded as k launches (the flags'
 * epoch is a kernel argument; a replayed graph would present it again).  Reports what the handle does at this k (after its first k-step). */
int mi_csr_spmk_info(mi_csr_t A, int k, int* eligible, int* one_launch, double* us_k_launches, double* us_one_launch);

/* host-only: derive the one-launch step's run dependencies as mi_csr_create would and CHECK them against the matrix (every column
 * a run names or loads belongs to a run on its list; MI_ERR_STATE names the first violation).  *eligible = 0 when the plan cannot
 * carry the step (rows outside the ring loop, or a band so wide that a run would wait for more than 64 others). */
int mi_spmk_plan_probe(int n, const int* ptrow, const int* indcol, int* eligible, int* runs, int* max_deps);

/* ---- BLAS-1 between SpMVs ---------------------------------------------- */
/* out = sum x_i y_i  (std::inner_product, mpk/SpMVmulti.cpp:147).  Fixed
 * two-stage reduction tree: deterministic run to run, not the CPU's order. */
int mi_dot(int n, const double* x, const double* y, double* out);
int mi_dot_dev(int n, const double* d_x, const double* d_y, double* d_out, mi_stream_t s);
/* y += a x  (VecAXPY at src/solve_newton.c:1269; the AXPY half of orthogonalize) */
int mi_axpy(int n, double a, const double* x, double* y);
int mi_axpy_dev(int n, double a, const double* d_x, double* d_y, mi_stream_t s);
/* x3 = x1 - alpha * (b . x1) * b ; *beta_out = b . x1
 * (orthogonalize, mpk/SpMVmulti.cpp:146-151; in-place twin mpk/2SpMV.cpp:3-11: pass x3 == x1).
 * The update is evaluated as the reference's object code does it: x3[i] = fma(-(alpha*beta), b[i], x1[i]).
 * Reductions enqueued on different streams use separate workspaces and may run concurrently. */
int mi_orthogonalize(int n, const double* b, const double* x1, double* x3, double alpha, double* beta_out);
int mi_orthogonalize_dev(int n, const double* d_b, const double* d_x1, double* d_x3, double alpha,
                         double* d_beta_out, mi_stream_t s);
/* The Krylov-step pipeline of mpk/SpMVmulti.cpp:563-569 — SpMV_CSR(x1, b, a); orthogonalize(nrow, b, x1, x3); SpMV_CSR(x2, x3, a)
 * — with the dot folded into the product (SURVEY.md §8 f-4 "fuse orthogonalisation with the k-step kernel"):
 *   mi_spmv_dot_dev            y = A x and *beta = b . y; b . y is accumulated in the product's epilogue, while each row's
 *                              value is still in its thread's register (ring kernel, LEAN form), so y and b are not read again
 *                              by a dot kernel of their own.  Handles whose launch cannot carry the epilogue (other kernels,
 *                              relabelled or blocked matrices) run product and dot as separate launches: same call, same bound.
 *   mi_spmv_orthogonalize_dev  x1 = A x; beta = b . x1 (epilogue); x3 = fma(-(alpha*beta), b, x1): two launches instead of three.
 * beta is a fixed-tree reduction like mi_dot's (deterministic run to run; its last bits differ from mi_dot's and from the CPU's
 * left-to-right sum: |beta - b.y| <= 1e-13 * sum |b_i y_i| is what the tests assert); every row of y / x1 and, given beta, every
 * element of x3 are the reference's bits.  Square or rectangular, unmapped matrices; device vectors. */
int mi_spmv_dot_dev(mi_csr_t A, const double* d_x, double* d_y, const double* d_b, double* d_beta_out, mi_stream_t s);
/* *in_epilogue = 1 if the next mi_spmv_dot_dev / mi_spmv_orthogonalize_dev on this handle carries the dot in the product's launch */
int mi_csr_dot_epilogue_info(mi_csr_t A, int* in_epilogue);
int mi_spmv_orthogonalize_dev(mi_csr_t A, const double* d_x, double* d_x1, const double* d_b, double* d_x3, double alpha,
                              double* d_beta_out, mi_stream_t s);
/* orthonormalize_against_basis(nrow, basis, y), mpk/2SpMV.cpp:13-28: for each of the m basis vectors IN TURN
 * dots[j] = y . v_j (on the y updated so far), y <- fma(-dots[j], v_j, y).  Like the reference nothing is
 * normalised (its norm is computed and dropped).  basis: HOST array of m pointers (host resp. device vectors).
 * m + 1 launches: each update also accumulates the next vector's dot. */
int mi_orthonormalize_against_basis(int n, int m, const double* const* basis, double* y, double* dots_out /* [m] or NULL */);
int mi_orthonormalize_against_basis_dev(int n, int m, const double* const* d_basis, double* d_y, double* d_dots /* [m] */,
                                        mi_stream_t s);
/* sqrt(sum x^2) (norm2, mpk/utils.cpp:131-136) and ||ref-test||/||ref|| (rel_error, :138-143) */
int mi_norm2(int n, const double* x, double* out);
int mi_norm2_dev(int n, const double* d_x, double* d_out, mi_stream_t s);
int mi_rel_error(int n, const double* ref, const double* test, double* out);
int mi_rel_error_dev(int n, const double* d_ref, const double* d_test, double* d_out, mi_stream_t s);
/* dst[i] = src[idx[i]] — halo pack */
int mi_gather_dev(int m, const int* d_idx, const double* d_src, double* d_dst, mi_stream_t s);

/* ---- BCSR 4x4 (SpMV_BCSR*, mpk/SpMV.cpp:90-219; row-major blocks) ------- */
int mi_bcsr4_create(int nbrows, int nbcols, const int* ptrow, const int* indcol, const double* coef,
                    mi_bcsr4_t* out);
int mi_bcsr4_destroy(mi_bcsr4_t A);
/* Block layout of the caller's coefficient array.  mpk/ stores a 4x4 block ROW-major (mpk/SpMV.cpp:112: blk[4*i + j]); PETSc's
 * MATSEQBAIJ — the matrix the reference's solver multiplies with (MatSetOperation(..., MATOP_MULT, ...), src/solve_newton.c:864-879;
 * kernels src/kernels/baij4_{mad,fma,avx2}.c) — stores it COLUMN-major (baij4_mad.c:73-76: row 0 is v[0], v[4], v[8], v[12]).
 * The *_layout entry points take either; column-major blocks are transposed on the way to the device (at create and at every
 * value refresh, never per product), after which the handle is an ordinary one: y is bit-equal to the row-major handle of the
 * transposed blocks, i.e. each row is the fma chain of SpMV_BCSR_FMA (mpk/SpMV.cpp:150-178).  (That is NOT the summation order of
 * MatMult_SeqBAIJ_4_AVX2, which keeps four per-column accumulators and adds them at the end, src/kernels/baij4_avx2.c:42-66; that
 * kernel needs PETSc to run, so against it parity is unpinned here — see INTEGRATION.md §4.) */
enum { MI_BLOCK_ROWMAJOR = 0, MI_BLOCK_COLMAJOR = 1 };
int mi_bcsr4_create_layout(int nbrows, int nbcols, const int* ptrow, const int* indcol, const double* coef, int layout, mi_bcsr4_t* out);
int mi_bcsr4_update_values_layout(mi_bcsr4_t A, const double* coef, int layout);                          /* host values */
int mi_bcsr4_update_values_layout_dev(mi_bcsr4_t A, const double* d_coef, int layout, mi_stream_t s);     /* device values */
/* The blocked kernel exists in two forms: x blocks gathered through L1/L2 per block (spmv_bcsr4), or each workgroup's distinct
 * block columns gathered once into an LDS tile and addressed through a 16-bit stream (spmv_bcsr4_tile; built when no group of 64
 * block rows touches more than 1024 block columns).  mi_bcsr4_create times both and keeps the faster; same bits.
 * MI355_BCSR_TILE=0|1 forces. */
int mi_bcsr4_tile_info(mi_bcsr4_t A, int* built, int* in_use, double* us_plain, double* us_tile);
/* Third form (round 4, spmv_bcsr_sell.hpp): a SLICED copy of the block values made at mi_bcsr4_create for matrices of >= 100 000 blocks
 * — 16 consecutive block rows per slice, padded to the slice's longest row, step j of a slice = the j-th block of each of its rows as
 * two contiguous kilobytes — streamed by persistent waves, each over one contiguous range of slices, with unconditional counted loads
 * (non-temporal where that measures faster) that never look at a row boundary; a slice's sums are parked in LDS and stored behind the
 * wave's last load (a store issued among the loads costs the read stream many times its bytes).  Same lanes per row, same fma order,
 * same bits.  Four variants are timed against the two forms above at create and the fastest of all runs (MI355_BCSR_SELL=0 never builds
 * it, =1 takes variant 0 unmeasured, MI355_BCSR_SELL_FORM=0..3 a given one).  The blocked copy of a relabelled matrix runs it too (a node's
 * four sums leave through the block-row map).  The sliced values follow mi_bcsr4_update_values* at once, on that call's stream.  *form_in_use: -1 none; 0 one wave per SIMD, eight steps of prefetch, non-temporal value loads; 1 the
 * same with temporal loads; 2 two waves per SIMD (one workgroup of eight waves per CU), four steps, non-temporal; 3 as 0 with twelve steps;
 * *padding = padded places / blocks - 1; us[f] = microseconds per launch measured for variant f (0: not measured). */
int mi_bcsr4_sell_info(mi_bcsr4_t A, int* built, int* form_in_use, long long* steps, double* padding, double us[4]);
/* new block values (16 per block, row-major) for an unchanged block pattern; see mi_csr_update_values */
int mi_bcsr4_update_values(mi_bcsr4_t A, const double* coef);
int mi_bcsr4_update_values_dev(mi_bcsr4_t A, const double* d_coef, mi_stream_t s);
int mi_bcsr4_spmv(mi_bcsr4_t A, const double* x, double* y);
int mi_bcsr4_spmv_dev(mi_bcsr4_t A, const double* d_x, double* d_y, mi_stream_t s);
/* y_out[p] = A^(p+1) x for p < k on the blocked matrix: what SpM2V_BCSR{,_OPT,_FMA,_AVX2}(z, y, x, A, ptrowend1)
 * (mpk/SpM2V.cpp:375-801) return for k = 2 (y_out = {y, z}); k chained launches, each row one fma chain. */
int mi_bcsr4_spmk(mi_bcsr4_t A, int k, const double* x, double* const* y_out);
int mi_bcsr4_spmk_dev(mi_bcsr4_t A, int k, const double* d_x, double* const* d_y_out, mi_stream_t s);

/* ---- multi-vector products and the s-step Krylov basis ----
 * Y[:, j] = A X[:, j] for j < s with the matrix read ONCE for the s columns: MatMatMult_SeqBAIJ_4_AVX2(A, X, Y, s_step),
 * src/kernels/spmm_avx2.c:7-109.  X, Y dense column-major, column j at X + j*ldx (the reference's MatDense layout,
 * lda = 4 * mbs at :23); ldx >= 4*nbcols even, ldy >= 4*nbrows.  arith selects the association of a row's sum:
 *   MI_ARITH_CHAIN     one continuous fma chain over the row's blocks — every column bit-equal to SpMV_BCSR_FMA
 *                      (mpk/SpMV.cpp:150-178), i.e. to the CSR fma chain;
 *   MI_ARITH_BLOCKACC  per block the four products chained from zero, the partial added to the row value — what
 *                      spmm_avx2.c:77-88 and SpM2V_BCSR_OPT (mpk/SpM2V.cpp:502-507) do.  The reference's final
 *                      horizontal add (:96-101) sums four identical broadcast lanes and returns 4 A X; not reproduced. */
enum { MI_ARITH_CHAIN = 0, MI_ARITH_BLOCKACC = 1 };
int mi_bcsr4_spmm(mi_bcsr4_t A, int s, const double* X, long long ldx, double* Y, long long ldy, int arith);          /* host */
int mi_bcsr4_spmm_dev(mi_bcsr4_t A, int s, const double* d_X, long long ldx, double* d_Y, long long ldy, int arith,
                      mi_stream_t st);
/* The product exists in five forms with the same bits.  0: x blocks gathered through L1/L2 per block (spmm_bcsr4 / spmm_bcsr4_quad).
 * 1: tiles of up to 128 block rows — clusters of the block graph — gather the S columns of their distinct block columns ONCE into LDS
 * and the loop waits for coefficients only (spmm_bcsr4_tile, up to four columns).  2 / 3: the same with EIGHT lanes per block row
 * (64-row tiles; one 16-byte coefficient load per lane and block, the two halves of a row swapped through DPP; even column counts),
 * coefficients loaded temporally / non-temporally (spmm_bcsr4_otile).  The lists are built at the first product; the first product of
 * a handle at a column count times the possible forms and keeps the fastest (MI355_SPMM_TILE=0..4 forces one).  4 (round 4): the
 * SLICED stream of mi_bcsr4_sell_info with S sums per lane (spmm_bcsr4_sell; four or eight columns, unmapped products): the x blocks
 * shared inside the quad through DPP, a finished slice's sums parked in LDS and stored behind the loads.  *form_in_use = what
 * the next product runs; us[f] = microseconds per launch measured for form f (0: not possible / not yet measured). */
int mi_bcsr4_spmm_info(mi_bcsr4_t A, int s, int* tile_built, int* form_in_use, int* longest_list, double us[5]);
/* the same for a CSR handle (MI_ARITH_CHAIN bits = SpMV_CSR_FMA per column): one launch over the blocked copy when the
 * matrix has exact 4x4 node-block structure, else s single-vector launches */
int mi_spmm_dev(mi_csr_t A, int s, const double* d_X, long long ldx, double* d_Y, long long ldy, mi_stream_t st);
/* orth == 0: V[:, 0] = v0, V[:, k+1] = A V[:, k], k < s — BuildKrylovBasis_AVX2, src/kernels/spmm_avx2.c:112-168 (dense
 * column-major n x (s+1) V, ldv >= n): the monomial basis, bit-equal to the matrix-powers chain.
 * orth != 0: the orthonormal (Arnoldi) basis of the same space, from the reference's own pieces: V[:, 0] = v0/||v0||, each
 * product passed through orthonormalize_against_basis (mpk/2SpMV.cpp:13-28) against the earlier columns, then divided by
 * its norm2 — the normalisation that helper computes and drops.  d_coef (s*(s+2)+1 doubles): [k*(s+2) + j] = dot of step
 * k with column j (j <= k), [k*(s+2) + k+1] = the norm, [s*(s+2)] = ||v0||. */
int mi_krylov_basis_dev(mi_csr_t A, int s, const double* d_v0, double* d_V, long long ldv, int orth, double* d_coef,
                        mi_stream_t st);

/* ---- row-range partition of one matrix over the GPUs of a node ----------
 * New design (the reference has no distributed code).  Rank r
 * owns global rows [row_starts[r], row_starts[r+1]) and the matching slice of
 * x and y.  Columns are relabelled to [0,n_local) owned | [n_local,
 * n_local+n_halo) ghosts (ascending global id, hence contiguous per owner);
 * rows are split into interior (no ghost column) and boundary rows so the
 * interior SpMV overlaps the halo exchange.  The exchange itself is done by
 * the caller (torch.distributed over RCCL in bench.py) between
 * mi_part_pack_dev and mi_part_spmv_boundary_dev.
 *
 * Planning functions are host-only and work without a GPU. */
int mi_part_create(int nranks, int rank, const long long* row_starts, const int* ptrow,
                   const int* indcol_global, const double* coef, mi_part_t* out);
int mi_part_destroy(mi_part_t P);
int mi_part_sizes(mi_part_t P, int* n_local, int* n_halo, int* n_interior_rows, int* n_boundary_rows);
int mi_part_recv_counts(mi_part_t P, int* counts /* [nranks] */);
int mi_part_recv_ids(mi_part_t P, int peer, long long* ids /* [recv_counts[peer]] global, ascending */);
int mi_part_set_send_ids(mi_part_t P, int peer, int count, const long long* ids /* global ids peer needs from me */);
int mi_part_send_counts(mi_part_t P, int* counts /* [nranks] */);
/* local pieces on the host, for CPU checks: which = 0 interior, 1 boundary, 2 = the one-launch step's piece: ALL local rows
 * in natural order (rowmap NULL), columns numbered [ghosts of lower ranks | owned | ghosts of higher ranks];
 * mi_part_combined_info gives the number of ghosts in front */
int mi_part_local_csr(mi_part_t P, int which, int* nrows, const int** ptrow, const int** indcol_local,
                      const double** coef, const int** rowmap);
int mi_part_combined_info(mi_part_t P, int* n_left);
/* *contiguous = 1 when every send list of this rank is a run of consecutive local ids (banded partitions with whole-range halos): the
 * RCCL step (mi_part_spmv_dev) then sends slices of x in place and launches no pack kernel. */
int mi_part_sends_contiguous(mi_part_t P, int* contiguous);
int mi_part_send_index(mi_part_t P, int* total, const int** local_idx /* packed by peer, ascending */);
/* upload the two pieces and the send index to the current device */
int mi_part_finalize(mi_part_t P);
int mi_part_set_kernel(mi_part_t P, int kernel_id);
/* new coefficients for an unchanged pattern: this rank's values in the order of the arrays given to mi_part_create
 * (host pointer; after mi_part_finalize).  Every piece of the handle is refreshed, plans and exchange set-up are kept. */
int mi_part_update_values(mi_part_t P, const double* coef);
/* per step, on device: x_ext = [x_local | halo], sendbuf packed by peer */
int mi_part_pack_dev(mi_part_t P, const double* d_x_ext, double* d_sendbuf, mi_stream_t s);
int mi_part_spmv_interior_dev(mi_part_t P, const double* d_x_ext, double* d_y_local, mi_stream_t s);
int mi_part_spmv_boundary_dev(mi_part_t P, const double* d_x_ext, double* d_y_local, mi_stream_t s);

/* ---- native halo exchange: RCCL point-to-point over xGMI, driven from C++ ----
 * Optional fast path for the whole step (pack, exchange on the partition's own comm
 * stream, interior overlapped, boundary) with no per-step host work outside this
 * library.  librccl is resolved at run time (dlopen); when it is missing these
 * return MI_ERR_UNSUPPORTED and the caller keeps exchanging the halos itself.
 * Bootstrap: rank 0 obtains a 128-byte id and hands it to every rank by any
 * side channel (bench.py: torch.distributed broadcast); then ALL ranks call
 * mi_part_comm_init collectively. */
#define MI_COMM_ID_BYTES 128
int mi_comm_available(void); /* MI_OK iff librccl could be resolved in this process */
int mi_comm_unique_id(void* id128);
int mi_part_comm_init(mi_part_t P, const void* id128);
/* what the communicator made by mi_part_comm_init says about itself (ncclCommCount / ncclCommUserRank): *comm_ranks = 0 and
 * *comm_rank = -1 when the handle has none.  bench.py prints it as `rccl_ranks`, so that a multi-GPU line shows how many ranks
 * RCCL really connected. */
int mi_part_comm_info(mi_part_t P, int* comm_ranks, int* comm_rank);
/* y_local = (A x)_local: d_x_ext = [x_local | halo], the halo part is overwritten.
 * Cross-stream hand-offs are HIP events; MI355_PART_HANDOFF=flags selects one-wave flag kernels instead
 * (cheaper, but a wait on a stalled peer then spins on the GPU: it gives up after minutes and the give-up is
 * sticky — this call, mi_part_status and mi_part_destroy return MI_ERR_HIP from then on). */
int mi_part_spmv_dev(mi_part_t P, double* d_x_ext, double* d_y_local, mi_stream_t s);
/* MI_OK, or MI_ERR_HIP once a hand-off wait has given up (call after synchronising; no GPU work, no copy) */
int mi_part_status(mi_part_t P);
/* The ALL-GATHER form of that step, for wide halos (an FE slab partition: a whole mesh plane per neighbour, 38 648 ghosts of 163 048
 * rows at N = 8): instead of one grouped ncclSend / ncclRecv pair per neighbour, every rank contributes ONE fixed-size slice — the
 * entries anybody needs from it, i.e. the sorted union of its send lists, padded to the largest such union M — to one ncclAllGather
 * on the comm stream, and every ghost is picked out of the gathered nranks x M buffer (BASELINE north_star: "RCCL all-gather of halo x
 * entries over xGMI overlapped with interior SpMV").  Set-up (collective, after mi_part_comm_init): mi_part_send_union on every rank,
 * all-gather the counts and the GLOBAL ids (local id + the rank's first row) by any side channel, mi_part_allgather_setup with all of
 * them; mi_part_set_allgather(P, 1) on EVERY rank (or on none) selects the form for mi_part_spmv_dev.  Same bits either way. */
int mi_part_send_union(mi_part_t P, int* count, const int** local_idx /* ascending local ids; valid until the handle is destroyed */);
int mi_part_allgather_setup(mi_part_t P, const int* counts /* [nranks] */, const long long* ids /* every rank's union as global ids, rank after rank */);
int mi_part_set_allgather(mi_part_t P, int on);
int mi_part_allgather_info(mi_part_t P, int* ready, int* in_use, int* slice /* M */);
/* ---- halo exchange by peer push (no RCCL, one stream) -------------------------
 * Each rank's kernel writes the x entries a neighbour needs straight into a receive window in the NEIGHBOUR's memory
 * (HIP IPC mapping; xGMI between GPUs) and raises a flag there; the receiver's kernel waits for its neighbours' flags and
 * moves the window behind x_local (navierstokes_amd/csrc/push_exchange.hpp has the protocol).  The step is four launches
 * on the caller's stream — push, interior rows, wait + copy, boundary rows — with no RCCL call, no second stream and no
 * cross-stream hand-off.  Set-up (collective, once): every rank calls mi_part_push_export, the 64-byte handles and the
 * (2*nranks+1)-entry layouts are all-gathered by any side channel, every rank calls mi_part_push_connect with all of
 * them.  One PROCESS per rank (ranks as threads of one process share hardware queues and can deadlock in the wait).
 * All ranks must then call mi_part_spmv_push_dev the same number of times (the flags carry the step number; for the same reason the
 * call refuses to be captured into a HIP graph: MI_ERR_UNSUPPORTED).
 * A wait on a stalled neighbour gives up after minutes; that is sticky and reported like a hand-off time-out. */
#define MI_IPC_HANDLE_BYTES 64
int mi_part_push_export(mi_part_t P, void* handle64, long long* layout /* [2*nranks + 1] */);
int mi_part_push_connect(mi_part_t P, const void* handles /* nranks x 64 B */, const long long* layouts /* nranks x (2*nranks+1) */);
int mi_part_spmv_push_dev(mi_part_t P, double* d_x_ext, double* d_y_local, mi_stream_t s);
/* *fused = 1: the step is ONE launch — when all local rows form a piece the sliced-stream kernel (round 5: spmv_sstream_fused) or the
 * ring kernel (spmv_ring.hpp, FUSED) serves, the push duty and the ghost reads (straight from the window, behind an in-kernel wait)
 * live inside that kernel, in the few workgroups / runs whose rows touch ghosts; these get a shorter share of the rows, push first and wait
 * second (mi_part_kernel_name(P, 2) names the kernel; MI355_PUSH_FUSED=0 disables the form, MI355_PUSH_FUSED_KERNEL=ring|sstream|csr_ext forces
 * one).  A rank whose rows have the 4x4 node structure runs the blocked kernel's one-launch form, spmv_bcsr4_fused_ext
 * (spmv_bcsr4_ext.hpp): the launch's first workgroups push, wait and copy the window once into a cached buffer of the handle, the
 * workgroups whose rows name ghosts wait for THEM and read that buffer (MI355_PUSH_FUSED_EXT=0 keeps the four launches for such
 * ranks; ranks that share a device — a neighbour's window lives on this rank's device — run it as two launches, so that only the
 * exchange's few workgroups wait in-kernel: MI355_PUSH_EXT_SPLIT=0|1 forces).  A rank that gets none of these forms (scalar rows, a halo
 * too wide for the sliced stream's and the ring's fused forms: a 3-D mesh operator over ranks) runs the same staged step on the stream
 * kernel's row blocks, spmv_csr_fused_ext (MI355_PUSH_FUSED_CSR_EXT=0 keeps the four launches).  In the fused forms the halo part of d_x_ext is
 * neither read nor written (d_x_ext must still hold n_local + n_halo entries for the four-launch form the ranks may have to
 * agree on). */
int mi_part_push_info(mi_part_t P, int* ready, int* fused, int* neighbours);
/* the kernel a piece's products launch (as rocprofv3 names it): which = 0 the interior rows' piece, 1 the boundary rows', 2 the combined piece
 * of the one-launch push step — spmv_sstream_fused<...> (round 5) wherever that piece holds a sliced copy, else the ring kernel's FUSED
 * form, the blocked one (spmv_bcsr4_fused_ext) or the staged scalar one (spmv_csr_fused_ext); "" when the step is not fused.  The string lives until the thread's next call. */
const char* mi_part_kernel_name(mi_part_t P, int which);
/* Step down from the one-launch form to the four-launch form (push, interior rows, wait + copy, boundary rows).  All ranks must
 * drive the step the same way; the caller compares mi_part_push_info's `fused` across ranks and calls this where they differ. */
int mi_part_push_unfuse(mi_part_t P);
/* give the push exchange up again (e.g. after a failed collective self-check): windows released, sticky give-ups cleared */
int mi_part_push_disable(mi_part_t P);
/* development check of the RCCL plumbing on ONE GPU: a communicator of size 1 sends
 * `count` doubles to itself through the same send/recv/stream/event code path
 * (the per-step cost measurements of this path live in tools/comm_timing.hip) */
int mi_comm_selftest(int count, double* max_abs_err);

/* ---- ONE process, N GPUs: the row-range partition behind a single handle -------------------------------------------
 * "mi_dist_create(ndev, ...)"; "Threading: one host thread drives all GPUs (or one per GPU internally), hidden
 * behind the ABI".  The reference's callers are single-process C++ programs that call SpMV_CSR(y, x, A) from one thread
 * (mpk/2SpMV.cpp:128-141, mpk/SpM2V.cpp:885-889, mpk/SpMVmulti0.cpp:369-411 behind the seam mpk/SpMV.h:52-66): this is the form
 * of the multi-GPU path they can reach (the mpk/SpMV.h shim routes to it with MI355_NGPUS=N).  The
 * mi_part_* entry points above are its per-rank building blocks and stay what a process-per-GPU host (bench.py, torch.distributed)
 * drives itself.
 *
 * mi_dist_create cuts the n rows into ndev contiguous ranges of (nearly) equal nonzero count — at node boundaries for a matrix
 * with exact 4x4 node-block structure —, plans every rank's share (partition.hpp: columns relabelled to owned | ghosts, each
 * row's nonzeros kept in the caller's order, so every row of y is the CSR-ordered fma chain of SpMV_CSR_FMA for every ndev),
 * uploads rank r's pieces to device r mod (devices present) — MI355_DIST_DEVICES="0,2,..." overrides the map — and starts one
 * worker thread per rank, which enqueues that rank's launches from then on.  Host pointers; the caller's arrays are not retained.
 * The halo exchange of a step is chosen at create (mi_dist_exchange_name / _note say which and why):
 *   "push"   ranks on distinct devices with peer access: the peer-push step of mi_part_spmv_push_dev (one launch per rank in its
 *            fused form) with the neighbours' windows reached through peer pointers;
 *   "rccl"   mi_part_spmv_dev per rank (grouped ncclSend / ncclRecv on the rank's comm stream);
 *   "event"  any rank-to-device map, N ranks on ONE device included: halo entries copied straight into the peers' vectors with
 *            hipMemcpyPeerAsync on the sender's stream, ordered by HIP events.
 * A candidate is used only after one product through it equalled one through the event exchange bit for bit on every rank.
 * MI355_DIST_EXCHANGE=event|push|rccl forces one (create fails if it is not usable). */
typedef struct mi_dist_s* mi_dist_t;
typedef struct mi_dist_vec_s* mi_dist_vec_t; /* a vector distributed like the rows: per rank [owned | halo] on the rank's device */
int mi_dist_create(int ndev, int n, const int* ptrow, const int* indcol, const double* coef, mi_dist_t* out);
int mi_dist_destroy(mi_dist_t D);
/* *exchange: 0 event, 1 push, 2 rccl; *fused = 1: the push step is one launch per rank; halo_*: ghost entries over all ranks / of the largest */
int mi_dist_info(mi_dist_t D, int* nranks, int* distinct_devices, int* exchange, int* fused, long long* halo_total, long long* halo_max);
const char* mi_dist_exchange_name(mi_dist_t D);
const char* mi_dist_exchange_note(mi_dist_t D); /* what was tried at create, in order, and why a candidate was dropped */
int mi_dist_rank_info(mi_dist_t D, int rank, int* device, long long* row_start, int* n_local, int* n_halo, long long* nnz_local,
                      mi_stream_t* stream /* the stream the rank's work is enqueued on */);
/* new coefficients for an unchanged pattern, in the order of the arrays given to mi_dist_create (see mi_csr_update_values) */
int mi_dist_update_values(mi_dist_t D, const double* coef);
/* Host vectors of length n — the reference's calling convention: scatter, compute, gather, synchronise.
 *   mi_dist_spmv          y = A x                                  SpMV_CSR{,_OPT,_FMA,_AVX2}, mpk/SpMV.cpp:6-85
 *   mi_dist_spmk          y_out[p] = A^(p+1) x, one exchange per power   SpM2V_CSR mpk/SpM2V.cpp:79-112, SpM3V / SpM4V mpk/SpMVmulti0.cpp:132-221
 *   mi_dist_dot           every rank's fixed-tree partial (mi_dot_dev), summed on the host in rank order: deterministic for a given
 *                         ndev, inside the bound documented at mi_dot
 *   mi_dist_orthogonalize x3 = fma(-(alpha * beta), b, x1) with beta = that dot   orthogonalize, mpk/SpMVmulti.cpp:146-151
 *                         (x3 == x1 allowed: the in-place form of mpk/2SpMV.cpp:3-11); given beta the update is the reference's bits */
int mi_dist_spmv(mi_dist_t D, const double* x, double* y);
int mi_dist_spmk(mi_dist_t D, int k, const double* x, double* const* y_out);
int mi_dist_dot(mi_dist_t D, const double* x, const double* y, double* out);
int mi_dist_orthogonalize(mi_dist_t D, const double* b, const double* x1, double* x3, double alpha, double* beta_out);
/* Device-resident vectors (a solver that keeps its vectors on the GPUs between products).  mi_dist_vec_set / _get move a full
 * host vector in / out (synchronous); mi_dist_vec_ptr gives rank r's buffer (n_local + n_halo doubles on the rank's device; the
 * owned entries first) for the caller's own kernels, to be enqueued on the rank's stream (mi_dist_rank_info). */
int mi_dist_vec_create(mi_dist_t D, mi_dist_vec_t* out);
int mi_dist_vec_destroy(mi_dist_vec_t v);
int mi_dist_vec_set(mi_dist_vec_t v, const double* host /* [n] */);
int mi_dist_vec_get(mi_dist_vec_t v, double* host /* [n] */);
int mi_dist_vec_ptr(mi_dist_vec_t v, int rank, double** d_ptr);
/* Asynchronous: the step is enqueued on every rank's stream when the call returns; mi_dist_synchronize waits for all ranks and
 * reports an in-kernel wait that gave up (push) or a hand-off time-out (rccl).  x and y (and the k outputs) must be distinct vectors. */
int mi_dist_spmv_dev(mi_dist_t D, mi_dist_vec_t x, mi_dist_vec_t y);
int mi_dist_spmk_dev(mi_dist_t D, int k, mi_dist_vec_t x, const mi_dist_vec_t* y_out /* host array of k vectors */);
int mi_dist_synchronize(mi_dist_t D);
/* the reductions return their scalar on the host and therefore synchronise */
int mi_dist_dot_dev(mi_dist_t D, mi_dist_vec_t a, mi_dist_vec_t b, double* out);
int mi_dist_orthogonalize_dev(mi_dist_t D, mi_dist_vec_t b, mi_dist_vec_t x1, mi_dist_vec_t x3, double alpha, double* beta_out);

#ifdef __cplusplus
}
#endif
#endif /* MI355_SPMV_H */
