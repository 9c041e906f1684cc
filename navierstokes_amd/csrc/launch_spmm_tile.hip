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
    const int nwg = (A->nbrows + Pl->rows - 1) / Pl->rows;
    Bcsr4Tile Tl{Pl->d_ptr, Pl->d_nodes, Pl->d_slots};
    static const int chunk = getenv("MI355_SPMM_TILE_XCD_CHUNK") ? atoi(getenv("MI355_SPMM_TILE_XCD_CHUNK")) : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(kSpmmTileThreads), spmm_tile_lds(Pl, S), st, V, Tl, X, ldx, Y, ldy, nwg, chunk);
    return hipGetLastError();
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
