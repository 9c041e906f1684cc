// spmk_ring.hpp — the matrix-powers step y_p = A^p x, p = 1..k, as ONE launch: the ring kernel (spmv_ring.hpp, LEAN
// form) with a loop over the powers around its loop over the row blocks.
//
// Reference: SpM2V_CSR (mpk/SpM2V.cpp:79-112), SpM3V / SpM4V (mpk/SpMVmulti0.cpp:132-221) fuse the k sweeps into one
// traversal on the CPU (first-touch tables, one thread).  On the GPU every power is a row-parallel sweep and power p + 1
// of a row needs power p of the rows its columns name — rows of OTHER workgroups.  Here every persistent workgroup keeps
// its run of row blocks for all k powers (its plan records stay in LDS) and the hand-off between powers is per run, not
// global:
//   * a run's y values of power p are stored WRITE-THROUGH (agent-scope relaxed atomic stores = `global_store ... sc1`: they
//     do not stay dirty in the XCD's L2), every storing wave drains them (`s_waitcnt vmcnt(0)`), the workgroup meets at a
//     barrier and ONE lane stores epoch + p into the run's flag (`sc1`) — the "write-through payload, drained, then flag"
//     form of MI355X_MICROARCH.md (valid forms; price list rows handoff-flag / publish-large);
//   * before power p + 1 a workgroup polls (`sc1` loads, bounded, with back-off) the flags of exactly the runs whose rows
//     hold ANY column it loads (host-side list, ring_plan.hpp: build_run_deps — its two or four neighbours for a band as wide
//     as a run; over-reaching window fills and lanes included), meets at a barrier, and reads x = y_p with `sc1` loads (L1
//     bypassed).  No acquire fence: every load of handed-off bytes bypasses L1, and because no load ever touches a line
//     before its publication no cache of this XCD can hold a stale copy of it.  (With an agent-scope acquire per power —
//     `buffer_inv sc1` — the k = 4 step at 1 M rows measured 143 us, 10 % MORE than four launches; without it 113 us, 13 % less:
//     the invalidate, not the hand-off, was the price.  MI355_SPMK_ACQUIRE=1 puts it back for A/B.  tools/spmk_stress.py:
//     launches with a different x each, outputs poisoned, every word compared under uneven load.)
// Arithmetic is untouched: every row of every power is the same sequential fma chain as k chained products, and as the
// reference's fused CPU traversal — bit for bit.
//
// Residency: the hand-off needs every workgroup of the grid resident at once (a waiting workgroup holds its CU slot).  The
// grid is the ring plan's <= 512 workgroups at two per CU; the host launches this kernel only when the occupancy query
// confirms that, and every wait is bounded (a give-up is counted in host-visible memory, sticky, and fails the next call on
// the handle) — e.g. another process's kernel holding CUs makes the launch give up after ~4 s instead of hanging.
#pragma once
#include <hip/hip_runtime.h>

#include "spmv_ring.hpp"

namespace mi355 {

constexpr int kSpmkFusedMaxK = 8;

struct SpmkArgs {
    const double* x;
    double* y[kSpmkFusedMaxK];
    int k;
    unsigned epoch;        // flags of this launch count from here: power p (1-based) is published as epoch + p
    unsigned* flags;       // one per run (64 B apart)
    const int* dep_ptr;    // per run: its dependencies are dep_run[dep_ptr[g] .. dep_ptr[g + 1])
    const int* dep_run;
    unsigned* timeouts;    // host-visible
    unsigned spin_max;
    int acquire;           // A/B only: an agent-scope acquire (L1 invalidate) between powers
};

constexpr int kSpmkFlagStride = 16; // unsigneds: one 64-byte line per run

__device__ __forceinline__ double ld_coherent(const double* p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_coherent(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int T, int NNZB, int RING, int D, int MAXB, bool NT, bool SKEW>
__global__ __launch_bounds__(T) void spmk_csr_ring(CsrView A, const int4* __restrict__ plan, const unsigned short* __restrict__ slots,
                                                   const int2* __restrict__ run_rng, int bpw, SpmkArgs K)
{
    constexpr int PER = NNZB / T;
    constexpr bool PAIR = ring_pairs(T); // (ring_pair.hpp: the order of the 16-bit column stream this kernel shares with spmv_csr_ring)
    typedef unsigned short SlotVec __attribute__((ext_vector_type(PER)));
    constexpr int LDSN = NNZB + NNZB / 32 + 2;
    __shared__ __attribute__((aligned(16))) double s_cx_raw[2 * LDSN]; // two arrays of doubles (plain path), or LDSN {coef, x} pairs
    double* const s_c = s_cx_raw;
    double* const s_x = s_cx_raw + LDSN;
    RingCx* const s_cx = reinterpret_cast<RingCx*>(s_cx_raw);
    __shared__ double s_ring[RING];
    __shared__ int4 s_plan[2 * (MAXB + 2 * D + 2)];
    const int tid = threadIdx.x;
    const int bid = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int gw = (bid & (kNXCD - 1)) * (nwg / kNXCD) + (bid >> 3); // XCD-aware run order, as spmv_csr_ring
    const int2 rng = bpw > 0 ? make_int2(min(A.nblk, gw * bpw), min(A.nblk, (gw + 1) * bpw)) : run_rng[gw];
    const int b_begin = rng.x;
    const int nb = rng.y - rng.x;
    // A run without blocks owns no rows: nobody depends on it (build_run_deps lists only runs with rows) and it has nothing to do.
    if (nb <= 0) return;
    const int clast = A.ncols - 1;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

    for (int i = tid; i < 2 * nb; i += T) s_plan[i] = plan[2 * b_begin + i];
    __syncthreads();
    {
        const int4 l0 = s_plan[2 * (nb - 1)], l1 = s_plan[2 * (nb - 1) + 1];
        const int4 sent = make_int4(l0.x + l0.z, l0.y + l0.w, 0, 0);
        for (int i = tid; i < 2 * D + 2; i += T) {
            s_plan[2 * (nb + i)] = sent;
            // the empty blocks behind the run load x where the run's last block did (no new column is written): every load of
            // this kernel stays inside the range build_run_deps accounts for
            s_plan[2 * (nb + i) + 1] = make_int4(l1.x, 0, 0, 0);
        }
    }
    __syncthreads();
    const SlotVec* slotv = reinterpret_cast<const SlotVec*>(slots);
    const int bslot_last = A.nblk - 1;
    const int dep0 = K.dep_ptr[gw], ndep = K.dep_ptr[gw + 1] - dep0;

    for (int pw = 0; pw < K.k; pw++) {
        // the vectors of this power: x = y_{pw-1} (the caller's x for the first), y = y_pw — uniform selects over the kernel arguments
        const double* x = K.x;
        double* y = K.y[0];
#pragma unroll
        for (int q = 1; q < kSpmkFusedMaxK; q++)
            if (pw == q) {
                x = K.y[q - 1];
                y = K.y[q];
            }
        if (pw > 0) { // every run whose rows my columns name has published power pw
            const unsigned want = K.epoch + (unsigned)pw;
            for (int j = tid; j < ndep; j += T) {
                const unsigned* f = K.flags + (size_t)K.dep_run[dep0 + j] * kSpmkFlagStride;
                unsigned spins = 0;
                while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                    if (spins < 256) __builtin_amdgcn_s_sleep(1);
                    else __builtin_amdgcn_s_sleep(64);
                    ++spins;
                    if (spins == 4096 && __hip_atomic_load(K.timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
                    if (spins > K.spin_max) {
                        __hip_atomic_fetch_add(K.timeouts, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __syncthreads();
            if (K.acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }

        double c[D][PER];
        SlotVec sl[D];
        int2 pr[D];
        double xr[D];

        auto issue = [&](int lb, int s) {
            const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
            ring_load_coefs<T, PER, NT, PAIR>(c[s], A.coef + uni(m0.y), tid & ((uni(m1.w) & 1) ? -1 : 0));
            sl[s] = (slotv + (size_t)min(b_begin + lb, bslot_last) * T)[tid];
            const int* rp = A.ptrow + uni(m0.x) + tid;
            pr[s] = make_int2(rp[0], rp[1]);
            xr[s] = ld_coherent(x + min(uni(m1.x) + tid, clast));
        };

#pragma unroll
        for (int s = 0; s < D; s++) issue(s, s);
        {
            constexpr int FILL = (RING + T - 1) / T;
            const int4 q = s_plan[1];
            const int c0 = uni(q.x) + tid, cend = uni(q.x) + uni(q.y), qz = uni(q.z);
            // Loads stay inside what build_run_deps accounts for this run — the first block's new columns, or T entries from q.x —
            // so that no line of y_p is touched before the runs that own it have published it (narrow bands: a run's rows plus
            // band can span fewer than RING columns, and an unclamped fill would reach into rows of runs not on the dependency list)
            const int chi = min(max(cend, uni(q.x) + T) - 1, clast);
            double v[FILL];
#pragma unroll
            for (int u = 0; u < FILL; u++) v[u] = ld_coherent(x + min(c0 + u * T, chi));
#pragma unroll
            for (int u = 0; u < FILL; u++)
                if (c0 + u * T < cend) s_ring[ring_slot<RING>(c0 + u * T, qz)] = v[u];
        }

        for (int g = 0; g < nb; g += D) {
#pragma unroll
            for (int s = 0; s < D; s++) {
                const int lb = g + s;
                const int4 m0 = s_plan[2 * lb];
                const int r0 = uni(m0.x), p0 = uni(m0.y), nrows = uni(m0.z);
                __syncthreads();
                double xv[PER];
#pragma unroll
                for (int i = 0; i < PER; i++) xv[i] = s_ring[min((unsigned)sl[s][i], (unsigned)(RING - 1))];
                if (kRingMergedStage) ring_stage_cx<T, PER, SKEW, PAIR>(s_cx, c[s], xv, tid);
                else ring_stage<T, PER, SKEW, PAIR>(s_c, s_x, c[s], xv, tid);
                const int2 prs = pr[s];
                issue(lb + D, s);
                __syncthreads();
                {
                    const int4 q4 = s_plan[2 * (lb + 1) + 1];
                    const int qx = uni(q4.x), qy = uni(q4.y), qz = uni(q4.z);
                    const double xn = xr[(s + 1) % D];
                    if (tid < qy) s_ring[ring_slot<RING>(qx + tid, qz)] = xn;
                }
                if (tid < nrows) st_coherent(y + r0 + tid, kRingMergedStage ? ring_row_chain_cx<8, SKEW>(s_cx, prs.x - p0, prs.y - p0) : ring_row_chain<8, SKEW>(s_c, s_x, prs.x - p0, prs.y - p0));
            }
        }
        if (pw + 1 < K.k) { // publish: every storing wave's stores have left, then the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(K.flags + (size_t)gw * kSpmkFlagStride, K.epoch + (unsigned)pw + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

} // namespace mi355
