// launch_spmk.hip — the matrix-powers step: one launch (spmk_ring.hpp) where the handle's ring plan allows it and it measures
// faster on this handle, else k chained launches.  Part of libmi355spmv.so (capi_internal.hpp).
#include "capi_internal.hpp"
#include "spmk_ring.hpp"

static bool env_is(const char* name, const char* v)
{
    const char* e = getenv(name);
    return e && !strcmp(e, v);
}

template <int D, bool NT, bool SKEW>
static hipError_t launch_fused_t(const mi_csr_s* H, const CsrView& V, const SpmkArgs& K, hipStream_t s, bool query, int* max_blocks)
{
    auto kern = spmk_csr_ring<256, 2048, 5120, D, kRingMaxB, NT, SKEW>;
    if (query) return hipOccupancyMaxActiveBlocksPerMultiprocessor(max_blocks, kern, 256, 0);
    hipLaunchKernelGGL(kern, dim3(H->ring.wgs), dim3(256), 0, s, V, reinterpret_cast<const int4*>(H->ring.d_plan), H->ring.d_slots,
                       reinterpret_cast<const int2*>(H->ring.d_rng), H->ring.uniform ? H->ring.bpw : 0, K);
    return hipGetLastError();
}

static hipError_t launch_fused(const mi_csr_s* H, const CsrView& V, const SpmkArgs& K, hipStream_t s, bool query = false, int* max_blocks = nullptr)
{
    const bool nt = H->ring.nt, sk = H->ring.skew;
    if (H->ring.cfg.depth == 4) {
        if (nt) return sk ? launch_fused_t<4, true, true>(H, V, K, s, query, max_blocks) : launch_fused_t<4, true, false>(H, V, K, s, query, max_blocks);
        return sk ? launch_fused_t<4, false, true>(H, V, K, s, query, max_blocks) : launch_fused_t<4, false, false>(H, V, K, s, query, max_blocks);
    }
    if (nt) return sk ? launch_fused_t<2, true, true>(H, V, K, s, query, max_blocks) : launch_fused_t<2, true, false>(H, V, K, s, query, max_blocks);
    return sk ? launch_fused_t<2, false, true>(H, V, K, s, query, max_blocks) : launch_fused_t<2, false, false>(H, V, K, s, query, max_blocks);
}

void spmk_release(mi_csr_t H)
{
    dfree(H->d_kflags);
    dfree(H->d_kdep_ptr);
    dfree(H->d_kdep_run);
    if (H->h_ktimeouts) (void)hipHostFree(H->h_ktimeouts);
    H->d_kflags = nullptr;
    H->d_kdep_ptr = H->d_kdep_run = nullptr;
    H->h_ktimeouts = H->d_ktimeouts = nullptr;
    H->kstep_setup = 0;
}

static CsrView unmapped_view(const mi_csr_s* H)
{
    CsrView V{};
    V.n = H->n;
    V.ncols = H->ncols;
    V.ptrow = H->d_ptrow;
    V.indcol = H->d_indcol;
    V.coef = H->d_coef;
    V.rowmap = nullptr;
    V.blk = nullptr;
    V.blk_span = nullptr;
    V.nblk = H->ring.nblk;
    return V;
}

// flags, dependency lists, give-up counter; -1 if this handle cannot run the one-launch form
static int spmk_setup(mi_csr_t H)
{
    if (H->kstep_setup) return H->kstep_setup;
    H->kstep_setup = -1;
    const RingTable& R = H->ring;
    // (a handle whose single products run the sliced stream still holds its ring plan: the one-launch step is built on that, and the
    // first k-step measures it against k launches of the sliced stream)
    const int kid = resolve_kernel(H);
    if (H->inner || H->n != H->ncols || (kid != MI_KERNEL_RING && kid != MI_KERNEL_SSTREAM) || !R.d_plan || R.ok_fraction < 0.90 || R.cfg.id != 4 || !R.lean || !R.all_in_loop ||
        R.cfg.depth == 3 || R.h_dep_ptr.empty() || R.wgs < kNXCD || R.wgs % kNXCD)
        return -1;
    int maxdep = 0;
    for (int g = 0; g < R.wgs; g++) maxdep = std::max(maxdep, R.h_dep_ptr[g + 1] - R.h_dep_ptr[g]);
    if (maxdep > 64) return -1; // a band much wider than a run: every power would wait for half the grid — k launches it is
    // every workgroup of the grid must be resident at once (a waiting workgroup keeps its slot)
    int per_cu = 0, dev = 0, cus = 0;
    SpmkArgs dummy{};
    if (launch_fused(H, unmapped_view(H), dummy, nullptr, true, &per_cu) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    if ((long long)per_cu * cus < R.wgs) return -1;
    hipError_t e;
    const size_t nflag = (size_t)R.wgs * kSpmkFlagStride;
    if ((e = hipMalloc(&H->d_kflags, sizeof(unsigned) * nflag)) != hipSuccess || (e = hipMemset(H->d_kflags, 0, sizeof(unsigned) * nflag)) != hipSuccess ||
        (e = hipMalloc(&H->d_kdep_ptr, sizeof(int) * R.h_dep_ptr.size())) != hipSuccess ||
        (e = hipMalloc(&H->d_kdep_run, sizeof(int) * std::max<size_t>(1, R.h_dep_run.size()))) != hipSuccess ||
        (e = hipMemcpy(H->d_kdep_ptr, R.h_dep_ptr.data(), sizeof(int) * R.h_dep_ptr.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (R.h_dep_run.size() && (e = hipMemcpy(H->d_kdep_run, R.h_dep_run.data(), sizeof(int) * R.h_dep_run.size(), hipMemcpyHostToDevice)) != hipSuccess) ||
        (e = hipHostMalloc((void**)&H->h_ktimeouts, sizeof(unsigned), hipHostMallocMapped)) != hipSuccess) {
        (void)hipGetLastError();
        spmk_release(H);
        H->kstep_setup = -1;
        return -1;
    }
    *H->h_ktimeouts = 0;
    if (hipHostGetDevicePointer((void**)&H->d_ktimeouts, H->h_ktimeouts, 0) != hipSuccess) {
        spmk_release(H);
        H->kstep_setup = -1;
        return -1;
    }
    H->kstep_epoch = 0;
    H->kstep_setup = 1;
    return 1;
}

static int spmk_chain(mi_csr_t H, int k, const double* d_x, double* const* d_y, hipStream_t s)
{
    const double* src = d_x;
    for (int p = 0; p < k; p++) {
        int rc = launch_spmv(H, src, d_y[p], s, false);
        if (rc) return rc;
        src = d_y[p];
    }
    return MI_OK;
}

static int spmk_fused(mi_csr_t H, int k, const double* d_x, double* const* d_y, hipStream_t s)
{
    SpmkArgs K{};
    K.x = d_x;
    for (int p = 0; p < k; p++) K.y[p] = d_y[p];
    for (int p = k; p < kSpmkFusedMaxK; p++) K.y[p] = d_y[k - 1];
    K.k = k;
    K.epoch = H->kstep_epoch;
    H->kstep_epoch += (unsigned)k; // the next launch's flags start above everything this one publishes
    K.flags = H->d_kflags;
    K.dep_ptr = H->d_kdep_ptr;
    K.dep_run = H->d_kdep_run;
    K.timeouts = H->d_ktimeouts;
    static const unsigned spin_max = 1u << (getenv("MI355_SPMK_SPIN_LOG2") ? std::max(8, std::min(30, atoi(getenv("MI355_SPMK_SPIN_LOG2")))) : 21);
    K.spin_max = spin_max;
    K.acquire = env_is("MI355_SPMK_ACQUIRE", "1") ? 1 : 0;
    hipError_t e = launch_fused(H, unmapped_view(H), K, s);
    if (e != hipSuccess) return fail(MI_ERR_HIP, std::string("one-launch powers step: ") + hipGetErrorString(e));
    return MI_OK;
}

int spmk_unmapped(mi_csr_t H, int k, const double* d_x, double* const* d_y, hipStream_t s)
{
    // a wait of an earlier one-launch step gave up (CUs held by somebody else's kernel): everything since is invalid
    if (H->h_ktimeouts && __atomic_load_n(H->h_ktimeouts, __ATOMIC_ACQUIRE) != 0)
        return fail(MI_ERR_HIP, "mi_spmk: a hand-off wait of the one-launch powers step gave up (workgroups of the grid were not all resident); "
                                "results since then are invalid — set MI355_SPMK_FUSED=0 when other kernels share the GPU");
    // (under stream capture: k launches — the one-launch step's flag epoch is a kernel argument, a replayed graph would reuse it)
    if (k < 2 || k > kSpmkFusedMaxK || env_is("MI355_SPMK_FUSED", "0") || stream_is_capturing(s) || spmk_setup(H) != 1) return spmk_chain(H, k, d_x, d_y, s);
    if (env_is("MI355_SPMK_FUSED", "1")) return spmk_fused(H, k, d_x, d_y, s);
    if (H->kstep_choice[k] == 0) {
        // first k-step of this handle at this k: both forms are the same bits, so run each a few times on the caller's own
        // vectors (outputs are fully overwritten either way) and keep the faster — the choice depends on matrix size and box
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            (void)hipGetLastError();
            return spmk_chain(H, k, d_x, d_y, s);
        }
        double us[2] = {0, 0};
        int rc = MI_OK;
        for (int round = 0; round < 2 && !rc; round++)
            for (int form = 0; form < 2 && !rc; form++) {
                const int warm = 2, timed = 5;
                for (int i = 0; i < warm && !rc; i++) rc = form ? spmk_fused(H, k, d_x, d_y, s) : spmk_chain(H, k, d_x, d_y, s);
                (void)hipEventRecord(e0, s);
                for (int i = 0; i < timed && !rc; i++) rc = form ? spmk_fused(H, k, d_x, d_y, s) : spmk_chain(H, k, d_x, d_y, s);
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                const double t = ms * 1e3 / timed;
                us[form] = us[form] > 0 ? std::min(us[form], t) : t;
                // a wait of the one-launch form gave up (the grid is not all resident: somebody else's kernel holds CUs): every further
                // launch of it would spin its whole budget again — the measurement ends here and the handle takes k launches
                if (form == 1 && H->h_ktimeouts && __atomic_load_n(H->h_ktimeouts, __ATOMIC_ACQUIRE) != 0) round = 2;
            }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (rc) return rc;
        H->kstep_us[k][0] = us[0];
        H->kstep_us[k][1] = us[1];
        const bool gave_up = H->h_ktimeouts && __atomic_load_n(H->h_ktimeouts, __ATOMIC_ACQUIRE) != 0;
        if (gave_up) *H->h_ktimeouts = 0; // (measured on scratch launches: the real product below is chained and valid)
        H->kstep_choice[k] = (!gave_up && us[1] > 0 && us[1] < 0.98 * us[0]) ? 1 : -1;
    }
    return H->kstep_choice[k] == 1 ? spmk_fused(H, k, d_x, d_y, s) : spmk_chain(H, k, d_x, d_y, s);
}

extern "C" int mi_csr_spmk_info(mi_csr_t A, int k, int* eligible, int* one_launch, double* us_k_launches, double* us_one_launch)
{
    CHECK_ARG(A && k >= 1 && k <= MI_MAX_POWERS, "bad argument");
    mi_csr_t H = A->inner ? A->inner : A;
    const bool small = k <= kSpmkFusedMaxK;
    if (eligible) *eligible = small && k >= 2 && H->kstep_setup == 1;
    if (one_launch) *one_launch = small && (env_is("MI355_SPMK_FUSED", "1") ? H->kstep_setup == 1 : H->kstep_choice[k] == 1);
    if (us_k_launches) *us_k_launches = small ? H->kstep_us[k][0] : 0.0;
    if (us_one_launch) *us_one_launch = small ? H->kstep_us[k][1] : 0.0;
    return MI_OK;
}
