// capi_bcsr.hip: BCSR 4x4 handles and products, multi-vector products, Krylov basis — part of libmi355spmv.so (see capi_internal.hpp for the layout of the library).
// Built for gfx950 only; no CPU fallback anywhere: every compute entry point needs a HIP device.
#include "capi_internal.hpp"
#include "spmv_bcsr_sell.hpp"

static int sell_fill(mi_bcsr4_t A, hipStream_t s);

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
    for (int s0 = 0; s0 < nbrows; s0 += kSellRows) // values of the longest slice of 16 block rows (bcsr4_refresh_kernel picks its LDS buffer by it)
        A->max_slice_vals = std::max(A->max_slice_vals, 16 * (ptrow[std::min(nbrows, s0 + kSellRows)] - ptrow[s0]));
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
    // The sliced copy (spmv_bcsr_sell.hpp): built for matrices large enough to stream (MI355_BCSR_SELL=0 never, =1 always and
    // unmeasured with D = 4, non-temporal); its four variants are timed against the row-per-quad kernels above and the fastest of
    // all is what mi_bcsr4_spmv* launches.  Costs a second copy of the block values on the device (+0.9 % padding on the FE matrix).
    {
        const char* se = getenv("MI355_BCSR_SELL");
        const char* at = getenv("MI355_SPMV_AUTOTUNE");
        const bool forced = se && !strcmp(se, "1");
        if (!(se && !strcmp(se, "0")) && (forced || nb >= 100000) && nbcols < (1 << 30)) {
            SellPlanHost P, P2;
            build_sell_plan(nbrows, ptrow, indcol, 1024, P);
            build_sell_wave_ranges(P, 2048, P2.wrng, P2.nwaves);
            const size_t vbytes = sizeof(double) * (size_t)(P.nsteps + kSellPadSteps) * kSellStepDoubles;
            if ((e = hipMalloc(&A->d_sell_val, vbytes)) != hipSuccess || (e = hipMalloc(&A->d_sell_col, sizeof(unsigned) * P.col.size())) != hipSuccess ||
                (e = hipMalloc(&A->d_sell_sptr, sizeof(int) * P.sptr.size())) != hipSuccess || (e = hipMalloc(&A->d_sell_wrng, sizeof(int) * P.wrng.size())) != hipSuccess ||
                (e = hipMalloc(&A->d_sell_wrng2, sizeof(int) * P2.wrng.size())) != hipSuccess ||
                (e = hipMemcpy(A->d_sell_wrng2, P2.wrng.data(), sizeof(int) * P2.wrng.size(), hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemset(A->d_sell_val + (size_t)P.nsteps * kSellStepDoubles, 0, sizeof(double) * (size_t)kSellPadSteps * kSellStepDoubles)) != hipSuccess ||
                (e = hipMemcpy(A->d_sell_col, P.col.data(), sizeof(unsigned) * P.col.size(), hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(A->d_sell_sptr, P.sptr.data(), sizeof(int) * P.sptr.size(), hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(A->d_sell_wrng, P.wrng.data(), sizeof(int) * P.wrng.size(), hipMemcpyHostToDevice)) != hipSuccess) {
                // no room for a second copy of the block values (or a failed upload): the row-per-quad kernels serve the handle, as
                // build_sstream does for the scalar format — a create that worked without the sliced copy must not fail because of it
                (void)hipGetLastError();
                dfree(A->d_sell_val); dfree(A->d_sell_col); dfree(A->d_sell_sptr); dfree(A->d_sell_wrng); dfree(A->d_sell_wrng2);
                A->d_sell_val = nullptr;
                A->d_sell_col = nullptr;
                A->d_sell_sptr = A->d_sell_wrng = A->d_sell_wrng2 = nullptr;
                A->sell_form = -1;
                if (e != hipErrorOutOfMemory) {
                    mi_bcsr4_destroy(A);
                    return fail(MI_ERR_HIP, std::string("bcsr4 sliced copy: ") + hipGetErrorString(e));
                }
            }
            if (A->d_sell_val) {
            A->sell_nslices = P.nslices;
            A->sell_nwaves = P.nwaves;
            A->sell_nwaves2 = P2.nwaves;
            A->sell_nsteps = P.nsteps;
            // filled HERE and refilled where the block values change (mi_bcsr4_update_values*), on that call's stream — never lazily in
            // front of a product: a product captured into a HIP graph holds only the product's node
            if ((rc = sell_fill(A, nullptr)) != MI_OK) {
                mi_bcsr4_destroy(A);
                return rc;
            }
            if (const char* fe = getenv("MI355_BCSR_SELL_FORM")) A->sell_form = std::max(0, std::min(3, atoi(fe))); // tests: one variant, unmeasured
            else if (forced) A->sell_form = 0;
            else if (!(at && !strcmp(at, "0"))) {
                double *tx = nullptr, *ty = nullptr;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                const size_t nx = 4 * (size_t)std::max(nbcols, 1), ny = 4 * (size_t)std::max(nbrows, 1);
                if (hipMalloc(&tx, sizeof(double) * nx) == hipSuccess && hipMalloc(&ty, sizeof(double) * ny) == hipSuccess &&
                    hipMemset(tx, 0, sizeof(double) * nx) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                    double us[5] = {0, 0, 0, 0, 0}; // [0] the row-per-quad choice made above, [1..4] the sliced variants
                    for (int round = 0; round < 2; round++)
                        for (int c = 0; c < 5; c++) {
                            A->sell_form = c - 1;
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
                    int best = 0;
                    for (int c = 1; c < 5; c++) {
                        A->tune_us_sell[c - 1] = us[c];
                        if (us[c] > 0 && us[c] < us[best]) best = c;
                    }
                    if (A->tune_us_plain <= 0 && A->tune_us_tile <= 0) A->tune_us_plain = us[0];
                    A->sell_form = best - 1;
                }
                dfree(tx);
                dfree(ty);
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
            } else A->sell_form = 0; // no measurement: the sliced, non-temporal form for matrices of this size
            }
        }
    }
    *out = A;
    return MI_OK;
}

// (re)fill the sliced copy from the row-major blocks, on stream s
static int sell_fill(mi_bcsr4_t A, hipStream_t s)
{
    const int grid = std::max(1, std::min(A->sell_nslices, 8192));
    hipLaunchKernelGGL(bcsr4_to_sell_kernel, dim3((unsigned)grid), dim3(64), 0, s, A->sell_nslices, A->nbrows, A->d_ptrow, A->d_coef, A->d_sell_sptr, A->d_sell_val);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// the blocked copy of a CSR handle from that handle's (new) CSR values, d_src — and, when d_csr_out is given, the handle's CSR value
// array — in one pass (bcsr4_refresh_kernel); every copy the blocked kernels read is current when this returns (in stream order)
int bcsr4_refresh_from_csr(mi_bcsr4_s* A, const int* d_csr_ptrow, const double* d_src, double* d_csr_out, hipStream_t s)
{
    if (!A || A->nbrows == 0) return MI_OK;
    const int nslices = (A->nbrows + kSellRows - 1) / kSellRows;
    const int grid = std::max(1, std::min(nslices, 2048));
    const int* sptr = A->d_sell_val ? A->d_sell_sptr : nullptr;
    double* sv = A->d_sell_val;
    if (A->max_slice_vals <= 4096) // 32 KB of LDS: four workgroups per CU (the FE rows of 56-60: 3 584-3 840 values per slice)
        hipLaunchKernelGGL((bcsr4_refresh_kernel<4096>), dim3((unsigned)grid), dim3(256), 0, s, nslices, A->nbrows, d_csr_ptrow, d_src, d_csr_out, A->d_ptrow, A->d_coef, sptr, sv);
    else if (A->max_slice_vals <= 16384)
        hipLaunchKernelGGL((bcsr4_refresh_kernel<16384>), dim3((unsigned)std::min(grid, 256)), dim3(256), 0, s, nslices, A->nbrows, d_csr_ptrow, d_src, d_csr_out, A->d_ptrow, A->d_coef, sptr, sv);
    else
        hipLaunchKernelGGL((bcsr4_refresh_kernel<0>), dim3((unsigned)grid), dim3(256), 0, s, nslices, A->nbrows, d_csr_ptrow, d_src, d_csr_out, A->d_ptrow, A->d_coef, sptr, sv);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// new block values from a device array in the caller's layout: the handle's row-major blocks and the sliced values in one pass
static int bcsr4_refresh_from_blocks(mi_bcsr4_s* A, const double* d_src, bool colmajor, hipStream_t s)
{
    if (!A || A->nbrows == 0 || A->nblocks == 0) return MI_OK;
    const int nslices = (A->nbrows + kSellRows - 1) / kSellRows;
    const int grid = std::max(1, std::min(nslices, 2048));
    const int* sptr = A->d_sell_val ? A->d_sell_sptr : nullptr;
    double* out = d_src == A->d_coef ? nullptr : A->d_coef;
    if (colmajor && !out) return fail(MI_ERR_ARG, "column-major blocks cannot be transposed in place");
    if (!out && !A->d_sell_val) return MI_OK;
#define BR_LAUNCH(CAP_, CM_, G_) hipLaunchKernelGGL((bcsr4_blocks_refresh_kernel<CAP_, CM_>), dim3((unsigned)(G_)), dim3(256), 0, s, nslices, A->nbrows, A->d_ptrow, d_src, out, sptr, A->d_sell_val)
    if (A->max_slice_vals <= 4096) { if (colmajor) BR_LAUNCH(4096, true, grid); else BR_LAUNCH(4096, false, grid); }
    else if (A->max_slice_vals <= 16384) { if (colmajor) BR_LAUNCH(16384, true, std::min(grid, 256)); else BR_LAUNCH(16384, false, std::min(grid, 256)); }
    else { if (colmajor) BR_LAUNCH(0, true, grid); else BR_LAUNCH(0, false, grid); }
#undef BR_LAUNCH
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

int bcsr4_values_changed(mi_bcsr4_s* A, hipStream_t s)
{
    if (A && A->d_sell_val && A->nblocks > 0) return sell_fill(A, s);
    return MI_OK;
}

extern "C" int mi_bcsr4_sell_info(mi_bcsr4_t A, int* built, int* form_in_use, long long* steps, double* padding, double us[4])
{
    CHECK_ARG(A, "null handle");
    if (built) *built = A->d_sell_val != nullptr;
    if (form_in_use) *form_in_use = A->d_sell_val ? A->sell_form : -1;
    if (steps) *steps = A->sell_nsteps;
    if (padding) *padding = A->nblocks > 0 && A->d_sell_val ? (double)A->sell_nsteps * kSellRows / (double)A->nblocks - 1.0 : 0.0;
    if (us)
        for (int i = 0; i < 4; i++) us[i] = A->tune_us_sell[i];
    return MI_OK;
}

// ---- PETSc BAIJ block layout (column-major 4x4 blocks: v[0], v[4], v[8], v[12] are row 0 — src/kernels/baij4_mad.c:73-76) ------
// The kernels read row-major blocks (mpk/SpMV.cpp:112).  A PETSc-side caller hands its Mat_SeqBAIJ arrays over as they are and
// says so; the blocks are transposed on the way to the device (setup-time / value-refresh traffic, never per product).
static void transpose_blocks_host(long long nb, const double* src, double* dst)
{
    for (long long k = 0; k < nb; k++)
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) dst[16 * k + 4 * r + c] = src[16 * k + 4 * c + r];
}

extern "C" int mi_bcsr4_create_layout(int nbrows, int nbcols, const int* ptrow, const int* indcol, const double* coef, int layout,
                                      mi_bcsr4_t* out)
{
    CHECK_ARG(layout == MI_BLOCK_ROWMAJOR || layout == MI_BLOCK_COLMAJOR, "unknown block layout");
    if (layout == MI_BLOCK_ROWMAJOR) return mi_bcsr4_create(nbrows, nbcols, ptrow, indcol, coef, out);
    CHECK_ARG(out, "out is null");
    *out = nullptr;
    CHECK_ARG(nbrows >= 0 && ptrow && ptrow[0] == 0 && ptrow[nbrows] >= 0, "bad argument");
    const long long nb = ptrow[nbrows];
    CHECK_ARG(nb == 0 || coef, "coef is null");
    std::vector<double> t((size_t)nb * 16);
    transpose_blocks_host(nb, coef, t.data());
    return mi_bcsr4_create(nbrows, nbcols, ptrow, indcol, t.data(), out);
}

extern "C" int mi_bcsr4_update_values_layout(mi_bcsr4_t A, const double* coef, int layout)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(layout == MI_BLOCK_ROWMAJOR || layout == MI_BLOCK_COLMAJOR, "unknown block layout");
    if (layout == MI_BLOCK_ROWMAJOR || A->nblocks == 0) return mi_bcsr4_update_values(A, coef);
    CHECK_ARG(coef, "null coef");
    std::vector<double> t((size_t)A->nblocks * 16);
    transpose_blocks_host(A->nblocks, coef, t.data());
    return mi_bcsr4_update_values(A, t.data());
}

extern "C" int mi_bcsr4_update_values_layout_dev(mi_bcsr4_t A, const double* d_coef, int layout, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    CHECK_ARG(layout == MI_BLOCK_ROWMAJOR || layout == MI_BLOCK_COLMAJOR, "unknown block layout");
    if (layout == MI_BLOCK_ROWMAJOR || A->nblocks == 0) return mi_bcsr4_update_values_dev(A, d_coef, s);
    CHECK_ARG(d_coef && d_coef != A->d_coef, "null coef, or the handle's own array");
    return bcsr4_refresh_from_blocks(A, d_coef, true, (hipStream_t)s); // transposed on the way, blocks and sliced values in one pass
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
    int rc = bcsr4_values_changed(A, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    return MI_OK;
}

extern "C" int mi_bcsr4_update_values_dev(mi_bcsr4_t A, const double* d_coef, mi_stream_t s)
{
    CHECK_ARG(A, "null handle");
    if (A->nblocks == 0) return MI_OK;
    CHECK_ARG(d_coef, "null coef");
    return bcsr4_refresh_from_blocks(A, d_coef, false, (hipStream_t)s);
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
    dfree(A->d_sell_val);
    dfree(A->d_sell_col);
    dfree(A->d_sell_sptr);
    dfree(A->d_sell_wrng);
    dfree(A->d_sell_wrng2);
    dfree(A->st.d_ptr);
    dfree(A->st.d_nodes);
    dfree(A->st.d_slots);
    dfree(A->st.d_rows);
    dfree(A->st64.d_ptr);
    dfree(A->st64.d_nodes);
    dfree(A->st64.d_slots);
    dfree(A->st64.d_rows);
    dfree(A->d_x);
    dfree(A->d_y);
    for (double* p : A->d_pow) dfree(p);
    delete A;
    return MI_OK;
}

int launch_bcsr4(mi_bcsr4_t A, const double* d_x, double* d_y, mi_stream_t s, bool use_map)
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
    // the sliced form (a relabelled matrix's blocked copy stores through its block-row map)
    if (A->sell_form >= 0 && A->d_sell_val) {
        // forms (mi_bcsr4_sell_info): 0 / 1 / 3 one wave per SIMD — one workgroup of four waves per CU, 8 (12) steps of prefetch,
        // non-temporal / temporal / non-temporal; 2 one workgroup of eight waves per CU, 4 steps, non-temporal.  All park y in LDS.
        const bool two = A->sell_form == 2;
        SellView S{A->d_sell_val, A->d_sell_col, A->d_sell_sptr, two ? A->d_sell_wrng2 : A->d_sell_wrng, A->sell_nslices, A->nbrows, V.browmap};
        const int swg = two ? A->sell_nwaves2 / 8 : A->sell_nwaves / 4;
        switch (A->sell_form) {
        case 0: hipLaunchKernelGGL((spmv_bcsr4_sell<8, true, 0, 2, 4>), dim3((unsigned)swg), dim3(256), 0, (hipStream_t)s, S, d_x, d_y, swg); break;
        case 1: hipLaunchKernelGGL((spmv_bcsr4_sell<8, false, 0, 2, 4>), dim3((unsigned)swg), dim3(256), 0, (hipStream_t)s, S, d_x, d_y, swg); break;
        case 2: hipLaunchKernelGGL((spmv_bcsr4_sell<4, true, 0, 2, 8>), dim3((unsigned)swg), dim3(512), 0, (hipStream_t)s, S, d_x, d_y, swg); break;
        default: hipLaunchKernelGGL((spmv_bcsr4_sell<12, true, 0, 2, 4>), dim3((unsigned)swg), dim3(256), 0, (hipStream_t)s, S, d_x, d_y, swg); break;
        }
        HIP_TRY(hipGetLastError());
        return MI_OK;
    }
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

void bcsr4_drop_sliced(mi_bcsr4_s* A)
{
    if (!A || !A->d_sell_val) return;
    (void)hipDeviceSynchronize();
    dfree(A->d_sell_val);
    dfree(A->d_sell_col);
    dfree(A->d_sell_sptr);
    dfree(A->d_sell_wrng);
    dfree(A->d_sell_wrng2);
    A->d_sell_val = nullptr;
    A->d_sell_col = nullptr;
    A->d_sell_sptr = A->d_sell_wrng = A->d_sell_wrng2 = nullptr;
    A->sell_form = -1;
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

// ---- the tile form (spmm_tile.hpp) ------------------------------------------------------------------------------------------

// Tiles of the multi-vector product: groups of at most `per` block rows, the list of distinct block columns each group touches, and
// every block's position in its group's list.  The groups are CLUSTERS of the block graph, not runs of consecutive rows: grown
// breadth-first from the lowest unassigned row over unassigned rows (a "ball" of the mesh), because the tile's cost — its gather, its
// LDS — is the number of distinct columns per row, and a ball of 128 nodes of a 3-D mesh touches ~2.5 per row where 128
// consecutive nodes (1.9 mesh lines) touch 5.1 (FE matrix, 68^3 cells: lists of 321 against 654 entries on average).  A cluster
// whose list would exceed `ucap` entries is cut in halves (in growth order) until it fits: the LDS footprint, hence the
// workgroups per CU, is set by the LONGEST list.  rows[t * per + i] = block row of lane group i of tile t, or -1 - (a valid row of
// the tile) for unused places (those lanes shadow that row and store nothing).
static int build_spmm_tile_plan(mi_bcsr4_t A, int per, int ucap, const std::vector<int>& ptrow, const std::vector<int>& indcol, SpmmTilePlan& T)
{
    const int nbr = A->nbrows;
    std::vector<int> order;           // block rows in cluster growth order
    std::vector<int> cuts;            // first position of every cluster, then cut further below
    order.reserve((size_t)nbr);
    {
        std::vector<char> assigned((size_t)nbr, 0);
        std::vector<int> stamp((size_t)nbr, 0), queue;
        int seed = 0, tid = 0, in_cur = 0;
        while (true) {
            while (seed < nbr && assigned[seed]) seed++;
            if (seed >= nbr) break;
            if (in_cur == 0) {
                tid++;
                cuts.push_back((int)order.size());
            }
            queue.clear();
            queue.push_back(seed);
            stamp[seed] = tid;
            for (size_t qh = 0; qh < queue.size() && in_cur < per; qh++) {
                const int r = queue[qh];
                order.push_back(r);
                assigned[r] = 1;
                in_cur++;
                for (int k = ptrow[r]; k < ptrow[r + 1]; k++) {
                    const int nb = indcol[k];
                    if (nb < nbr && !assigned[nb] && stamp[nb] != tid) { // (columns beyond the rows: a rectangular matrix has no such node)
                        stamp[nb] = tid;
                        queue.push_back(nb);
                    }
                }
            }
            if (in_cur == per) in_cur = 0; // full; else the component ran dry: the next seed continues this cluster
        }
        cuts.push_back((int)order.size());
    }
    // lists; clusters over the cap are halved
    std::vector<int> wg_ptr(1, 0), rows;
    std::vector<unsigned> nodes, u;
    std::vector<unsigned short> slots((size_t)A->nblocks + 1, 0);
    int umax = 0;
    std::vector<std::pair<int, int>> work; // [first, end) positions in `order`, processed in order (a stack keeps the order)
    for (size_t t = cuts.size() - 1; t-- > 0;) work.push_back({cuts[t], cuts[t + 1]});
    while (!work.empty()) {
        const std::pair<int, int> w = work.back();
        work.pop_back();
        if (w.first >= w.second) continue;
        u.clear();
        for (int i = w.first; i < w.second; i++) u.insert(u.end(), indcol.begin() + ptrow[order[i]], indcol.begin() + ptrow[order[i] + 1]);
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        if ((int)u.size() > ucap && w.second - w.first > 1) {
            const int mid = (w.first + w.second) / 2;
            work.push_back({mid, w.second});
            work.push_back({w.first, mid});
            continue;
        }
        if (u.size() > 65535) return -1;
        umax = std::max(umax, (int)u.size());
        for (int i = w.first; i < w.second; i++)
            for (int k = ptrow[order[i]]; k < ptrow[order[i] + 1]; k++)
                slots[k] = (unsigned short)(std::lower_bound(u.begin(), u.end(), (unsigned)indcol[k]) - u.begin());
        nodes.insert(nodes.end(), u.begin(), u.end());
        wg_ptr.push_back((int)nodes.size());
        // (round 5) the tile's rows in ASCENDING order, not in the order the cluster grew: neighbouring lane groups then stream
        // neighbouring rows' blocks (one run of the coefficient array per run of consecutive rows) and store neighbouring pieces of Y.
        // MI355_SPMM_TILE_SORT=0: growth order (A/B).
        static const bool sort_rows = !(getenv("MI355_SPMM_TILE_SORT") && !strcmp(getenv("MI355_SPMM_TILE_SORT"), "0"));
        if (sort_rows) std::sort(order.begin() + w.first, order.begin() + w.second);
        for (int i = 0; i < per; i++) rows.push_back(w.first + i < w.second ? order[w.first + i] : -1 - order[w.first]);
    }
    const int ntiles = (int)wg_ptr.size() - 1;
    if (umax < 1 || ntiles < 1) return -1;
    nodes.push_back(0);
    hipError_t er;
    if ((er = hipMalloc(&T.d_ptr, sizeof(int) * wg_ptr.size())) != hipSuccess ||
        (er = hipMalloc(&T.d_nodes, sizeof(unsigned) * nodes.size())) != hipSuccess ||
        (er = hipMalloc(&T.d_slots, sizeof(unsigned short) * slots.size())) != hipSuccess ||
        (er = hipMalloc(&T.d_rows, sizeof(int) * rows.size())) != hipSuccess ||
        (er = hipMemcpy(T.d_ptr, wg_ptr.data(), sizeof(int) * wg_ptr.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (er = hipMemcpy(T.d_nodes, nodes.data(), sizeof(unsigned) * nodes.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (er = hipMemcpy(T.d_slots, slots.data(), sizeof(unsigned short) * slots.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (er = hipMemcpy(T.d_rows, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice)) != hipSuccess) {
        (void)hipGetLastError();
        dfree(T.d_ptr);
        dfree(T.d_nodes);
        dfree(T.d_slots);
        dfree(T.d_rows);
        T = SpmmTilePlan();
        return -1;
    }
    T.rows = per;
    T.umax = umax;
    T.ntiles = ntiles;
    T.mean_list = (double)(nodes.size() - 1) / ntiles;
    return 1;
}

static int build_spmm_tile(mi_bcsr4_t A)
{
    if (A->st_state) return A->st_state;
    A->st_state = -1;
    const char* e = getenv("MI355_SPMM_TILE");
    if ((e && !strcmp(e, "0")) || A->nblocks < 4096) return -1;
    std::vector<int> ptrow((size_t)A->nbrows + 1), indcol((size_t)A->nblocks);
    if (hipMemcpy(ptrow.data(), A->d_ptrow, sizeof(int) * ptrow.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(indcol.data(), A->d_indcol, sizeof(int) * indcol.size(), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    // list caps: what lets two workgroups share a CU at four columns (128-row tiles: 368 x 144 B = 53 KB) and at eight (64-row tiles,
    // two quads per block row: 256 x 272 B = 70 KB); MI355_SPMM_TILE_UCAP=<128-row cap>,<64-row cap> for A/B
    int cap128 = 368, cap64 = 256;
    if (const char* ce = getenv("MI355_SPMM_TILE_UCAP")) (void)sscanf(ce, "%d,%d", &cap128, &cap64);
    A->st_state = build_spmm_tile_plan(A, 128, cap128, ptrow, indcol, A->st);
    (void)build_spmm_tile_plan(A, 64, cap64, ptrow, indcol, A->st64);
    return A->st_state;
}

static const SpmmTilePlan* spmm_plan_of(const mi_bcsr4_s* A, int s)
{
    if (s < 1 || s > 4) return nullptr;
    const SpmmTilePlan* Pl = A->st.d_ptr ? &A->st : nullptr;
    if (!Pl || spmm_tile_lds(Pl, s) > kLdsBytesPerCU) return nullptr;
    return Pl;
}

enum { kSpmmGather = 0, kSpmmTile = 1, kSpmmOct = 2, kSpmmOctNt = 3, kSpmmSell = 4, kSpmmForms = 5 };

static int sell_fill(mi_bcsr4_t A, hipStream_t s);

static bool spmm_form_possible(const mi_bcsr4_s* A, int s, int form, bool mapped = false)
{
    if (form == kSpmmGather) return true;
    // the sliced stream (spmm_bcsr4_sell): four or eight columns, unmapped products of a handle that holds the sliced copy
    if (form == kSpmmSell) return (s == 4 || s == 8) && A->d_sell_val; // (mapped products store through the block-row map)
    if (form == kSpmmTile) return spmm_plan_of(A, s) != nullptr;
    return s % 2 == 0 && A->st64.d_ptr && spmm_tile_lds(&A->st64, s) <= kLdsBytesPerCU;
}

static void launch_spmm_gather(const Bcsr4View& V, int m, int arith, const double* Xj, long long ldx, double* Yj, long long ldy, hipStream_t st)
{
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

static int launch_spmm(mi_bcsr4_t A, int s, int arith, const double* X, long long ldx, double* Y, long long ldy, hipStream_t st,
                       bool use_map)
{
    Bcsr4View V{A->nbrows, A->nbcols, A->d_ptrow, A->d_indcol, A->d_coef, use_map ? A->d_browmap : nullptr};
    for (int j0 = 0; j0 < s; j0 += 8) { // more than eight columns: batches of eight (the matrix is read once per batch)
        const int m = std::min(8, s - j0);
        const double* Xj = X + (size_t)j0 * ldx;
        double* Yj = Y + (size_t)j0 * ldy;
        // four forms, same bits: gather kernels; LDS tile with four lanes per block row (up to four columns); LDS tile with eight
        // lanes per block row (even column counts), coefficients loaded temporally or non-temporally.  The first product of a handle
        // at a column count times the possible ones and keeps the fastest (MI355_SPMM_TILE=0..3 forces a form)
        const bool mapped = V.browmap != nullptr;
        auto run = [&](int form) -> hipError_t {
            if (form == kSpmmSell) {
                SellView Sv{A->d_sell_val, A->d_sell_col, A->d_sell_sptr, A->d_sell_wrng, A->sell_nslices, A->nbrows, V.browmap};
                const int swg = A->sell_nwaves / 4;
                if (m == 4) {
                    if (arith == MI_ARITH_CHAIN) hipLaunchKernelGGL((spmm_bcsr4_sell<4, 0, 6, true>), dim3((unsigned)swg), dim3(256), 0, st, Sv, Xj, ldx, Yj, ldy, swg);
                    else hipLaunchKernelGGL((spmm_bcsr4_sell<4, 1, 6, true>), dim3((unsigned)swg), dim3(256), 0, st, Sv, Xj, ldx, Yj, ldy, swg);
                } else {
                    if (arith == MI_ARITH_CHAIN) hipLaunchKernelGGL((spmm_bcsr4_sell<8, 0, 6, true>), dim3((unsigned)swg), dim3(256), 0, st, Sv, Xj, ldx, Yj, ldy, swg);
                    else hipLaunchKernelGGL((spmm_bcsr4_sell<8, 1, 6, true>), dim3((unsigned)swg), dim3(256), 0, st, Sv, Xj, ldx, Yj, ldy, swg);
                }
                return hipGetLastError();
            }
            if (form == kSpmmTile) return spmm_tile_launch(A, spmm_plan_of(A, m), V, m, arith, Xj, ldx, Yj, ldy, st);
            if (form == kSpmmOct || form == kSpmmOctNt) return spmm_otile_launch(A, &A->st64, V, m, arith, form == kSpmmOctNt, Xj, ldx, Yj, ldy, st);
            launch_spmm_gather(V, m, arith, Xj, ldx, Yj, ldy, st);
            return hipGetLastError();
        };
        int form = kSpmmGather;
        const bool capturing = stream_is_capturing(st); // no first-use measurement (it synchronises) and no plan upload under capture
        if (capturing && A->spmm_choice[m] > 0 && (A->st_state == 1 || A->spmm_choice[m] - 1 == kSpmmSell) &&
            !(A->spmm_choice[m] - 1 == kSpmmSell && mapped))
            form = A->spmm_choice[m] - 1;
        const bool tiles = !capturing && build_spmm_tile(A) == 1;
        if (!capturing && (tiles || spmm_form_possible(A, m, kSpmmSell, mapped))) {
            const char* e = getenv("MI355_SPMM_TILE");
            const int forced = e ? atoi(e) : -1;
            auto possible = [&](int f) { return (f == kSpmmGather || f == kSpmmSell || tiles) && spmm_form_possible(A, m, f, mapped); };
            if (forced >= 0 && forced < kSpmmForms) form = possible(forced) ? forced : kSpmmGather;
            else {
                if (A->spmm_choice[m] == 0) {
                    hipEvent_t e0 = nullptr, e1 = nullptr;
                    int best = kSpmmGather;
                    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                        double us[kSpmmForms] = {0, 0, 0, 0, 0};
                        bool ok = true;
                        for (int round = 0; round < 2 && ok; round++)
                            for (int f = 0; f < kSpmmForms && ok; f++) {
                                if (!possible(f)) continue;
                                for (int i = 0; i < 2 && ok; i++) ok = run(f) == hipSuccess;
                                (void)hipEventRecord(e0, st);
                                for (int i = 0; i < 5 && ok; i++) ok = run(f) == hipSuccess;
                                (void)hipEventRecord(e1, st);
                                (void)hipEventSynchronize(e1);
                                float ms = 0.f;
                                (void)hipEventElapsedTime(&ms, e0, e1);
                                const double t = ms * 1e3 / 5;
                                us[f] = us[f] > 0 ? std::min(us[f], t) : t;
                            }
                        for (int f = 0; f < kSpmmForms; f++) {
                            A->spmm_us[m][f] = us[f];
                            if (ok && us[f] > 0 && us[f] < us[best]) best = f;
                        }
                        if (!ok) { (void)hipGetLastError(); best = kSpmmGather; }
                    }
                    if (e0) (void)hipEventDestroy(e0);
                    if (e1) (void)hipEventDestroy(e1);
                    A->spmm_choice[m] = 1 + best;
                }
                form = A->spmm_choice[m] - 1;
                if (!possible(form)) form = kSpmmGather; // (e.g. the measured choice was the sliced form and this product is a mapped one)
            }
        }
        hipError_t er = run(form);
        if (er != hipSuccess) return fail(MI_ERR_HIP, std::string("multi-vector product: ") + hipGetErrorString(er));
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

extern "C" int mi_bcsr4_spmm_info(mi_bcsr4_t A, int s, int* tile_built, int* form_in_use, int* longest_list, double us[5])
{
    CHECK_ARG(A && s >= 1 && s <= 8, "bad argument");
    if (tile_built) *tile_built = A->st_state == 1;
    if (form_in_use) {
        const char* e = getenv("MI355_SPMM_TILE");
        const int forced = e ? atoi(e) : -1;
        int form = kSpmmGather;
        if (A->st_state == 1 || spmm_form_possible(A, s, kSpmmSell)) {
            auto possible = [&](int f) { return (f == kSpmmGather || f == kSpmmSell || A->st_state == 1) && spmm_form_possible(A, s, f); };
            if (forced >= 0 && forced < kSpmmForms) form = possible(forced) ? forced : kSpmmGather;
            else if (A->spmm_choice[s] > 0) form = A->spmm_choice[s] - 1;
        }
        *form_in_use = form;
    }
    if (longest_list) {
        const SpmmTilePlan* Pl = spmm_plan_of(A, s) ? spmm_plan_of(A, s) : (A->st64.d_ptr ? &A->st64 : nullptr);
        *longest_list = Pl ? Pl->umax : 0;
    }
    if (us)
        for (int f = 0; f < kSpmmForms; f++) us[f] = A->spmm_us[s][f];
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
