// launch_spmm_tile.hip — instantiations of the multi-vector product's tile kernel (spmm_tile.hpp): 1..4 columns.
// Part of libmi355spmv.so (capi_internal.hpp).
#include "capi_internal.hpp"
#include "spmm_tile.hpp"

template <int S, int ARITH>
static hipError_t launch_t(const mi_bcsr4_s* A, const SpmmTilePlan* Pl, const Bcsr4View& V, const double* X, long long ldx, double* Y, long long ldy, hipStream_t st)
{
    constexpr int P = 3; // coefficient stages in flight per lane: 2, 3 and 4 measured within 1.5 % of each other
    auto kern = spmm_bcsr4_tile<S, ARITH, P>;
    static bool attr_set = false; // dynamic LDS beyond the default limit must be asked for, once per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesPerCU);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nwg = Pl->ntiles;
    Bcsr4Tile Tl{Pl->d_ptr, Pl->d_nodes, Pl->d_slots, Pl->d_rows};
    static const int chunk = getenv("MI355_SPMM_TILE_XCD_CHUNK") ? atoi(getenv("MI355_SPMM_TILE_XCD_CHUNK")) : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(kSpmmTileThreads), spmm_tile_lds(Pl, S), st, V, Tl, X, ldx, Y, ldy, nwg, chunk);
    return hipGetLastError();
}

// ---- eight lanes per block row (spmm_bcsr4_otile): 2, 4, 6, 8 columns on the 64-row plan
template <int S, int ARITH, bool NT>
static hipError_t launch_ot(const mi_bcsr4_s* A, const SpmmTilePlan* Pl, const Bcsr4View& V, const double* X, long long ldx, double* Y, long long ldy, hipStream_t st)
{
    constexpr int P = 4;
    auto kern = spmm_bcsr4_otile<S, ARITH, P, NT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesPerCU);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    Bcsr4Tile Tl{Pl->d_ptr, Pl->d_nodes, Pl->d_slots, Pl->d_rows};
    hipLaunchKernelGGL(kern, dim3((unsigned)Pl->ntiles), dim3(kSpmmTileThreads), spmm_tile_lds(Pl, S), st, V, Tl, X, ldx, Y, ldy, Pl->ntiles);
    return hipGetLastError();
}

template <int S>
static hipError_t launch_ot_s(const mi_bcsr4_s* A, const SpmmTilePlan* Pl, const Bcsr4View& V, int arith, bool nt, const double* X, long long ldx, double* Y, long long ldy, hipStream_t st)
{
    if (arith == MI_ARITH_BLOCKACC) return nt ? launch_ot<S, 1, true>(A, Pl, V, X, ldx, Y, ldy, st) : launch_ot<S, 1, false>(A, Pl, V, X, ldx, Y, ldy, st);
    return nt ? launch_ot<S, 0, true>(A, Pl, V, X, ldx, Y, ldy, st) : launch_ot<S, 0, false>(A, Pl, V, X, ldx, Y, ldy, st);
}

hipError_t spmm_otile_launch(const mi_bcsr4_s* A, const SpmmTilePlan* Pl, const Bcsr4View& V, int s, int arith, bool nt, const double* X, long long ldx,
                             double* Y, long long ldy, hipStream_t st)
{
    if (Pl->rows != 64) return hipErrorInvalidValue;
    switch (s) {
    case 2: return launch_ot_s<2>(A, Pl, V, arith, nt, X, ldx, Y, ldy, st);
    case 4: return launch_ot_s<4>(A, Pl, V, arith, nt, X, ldx, Y, ldy, st);
    case 6: return launch_ot_s<6>(A, Pl, V, arith, nt, X, ldx, Y, ldy, st);
    case 8: return launch_ot_s<8>(A, Pl, V, arith, nt, X, ldx, Y, ldy, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t spmm_tile_launch(const mi_bcsr4_s* A, const SpmmTilePlan* Pl, const Bcsr4View& V, int s, int arith, const double* X, long long ldx,
                            double* Y, long long ldy, hipStream_t st)
{
    const bool ba = arith == MI_ARITH_BLOCKACC;
    switch (s) {
    case 1: return ba ? launch_t<1, 1>(A, Pl, V, X, ldx, Y, ldy, st) : launch_t<1, 0>(A, Pl, V, X, ldx, Y, ldy, st);
    case 2: return ba ? launch_t<2, 1>(A, Pl, V, X, ldx, Y, ldy, st) : launch_t<2, 0>(A, Pl, V, X, ldx, Y, ldy, st);
    case 3: return ba ? launch_t<3, 1>(A, Pl, V, X, ldx, Y, ldy, st) : launch_t<3, 0>(A, Pl, V, X, ldx, Y, ldy, st);
    case 4: return ba ? launch_t<4, 1>(A, Pl, V, X, ldx, Y, ldy, st) : launch_t<4, 0>(A, Pl, V, X, ldx, Y, ldy, st);
    default: return hipErrorInvalidValue;
    }
}
