// spmv_mring.hpp — the multi-window ring kernel: spmv_csr_ring (spmv_ring.hpp) with K = 5 independent sliding x windows in
// LDS instead of one.  For 3-D mesh operators (the reference's scalar pressure operator, src/solve_newton.c), whose rows
// reach into three narrow column clusters two mesh planes apart: no single window can hold them, three small ones can, at
// any mesh size (mring_plan.hpp).  Everything but the refill is the ring kernel: persistent workgroups over runs of row
// blocks, matrix stream and new x columns D blocks ahead in registers, 16-bit precomputed LDS slots per nonzero, {coef, x}
// staged in LDS, one thread per row walking its segment with the sequential fma chain of the reference's SpMV_CSR_OPT/_FMA
// (mpk/SpMV.cpp:23-56) — bit-identical to every other kernel here.
//
// Refill: the plan lists a block's new columns as up to eight GROUPS of 64 consecutive columns, each with its first column and
// its first LDS slot; wave v of the workgroup prefetches groups v and v + 4, D blocks ahead, and stores them to their slots when
// the block becomes next.  Nothing is decoded here (one uniform LDS read per group and use), a group never wraps inside its
// window, and a block that needs more than eight groups starts a run, whose first block's windows the prologue fills whole
// from the run's record — so the steady-state loop has no global load under a condition (spmv_ring.hpp: LEAN).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mring_plan.hpp"
#include "spmv_ring.hpp"

namespace mi355 {

// TRACE (devtools build only, mi_debug_mring_trace): every workgroup leaves {start, end} of its run in 100 MHz ticks, its XCD and its
// number of blocks — the launch's timeline, for judging how the planner dealt the runs out.
template <int T, int NNZB, int D, int MAXB, bool MAPPED, bool NT, bool SKEW, bool TRACE = false>
__global__ __launch_bounds__(T) void spmv_csr_mring(CsrView A, const int4* __restrict__ plan, const int4* __restrict__ first,
                                                    const int* __restrict__ run_ok, const unsigned short* __restrict__ slots,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    const int2* __restrict__ run_rng, int nruns, long long* __restrict__ trace = nullptr)
{
    const long long t_start = TRACE ? (long long)wall_clock64() : 0;
    auto leave = [&](int nblocks) {
        if (TRACE && threadIdx.x == 0) {
            long long* o = trace + 4 * (size_t)blockIdx.x;
            o[0] = t_start;
            o[1] = (long long)wall_clock64();
            o[2] = (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) | ((long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) << 8); // HW_REG_XCC_ID | HW_REG_HW_ID[15:0] << 8 (cu_id [11:8], sh_id [12], se_id [15:13])
            o[3] = nblocks;
        }
    };
    constexpr int K = kMringK, W = kMringW, RING = K * W, PER = NNZB / T, R4 = kMringRec / 4;
    static_assert(T == 256 && kMringGroups == 8 && kMringRec == 20 && kMringFirst == 16, "two groups per wave; record layout");
    typedef unsigned short SlotVec __attribute__((ext_vector_type(PER)));
    constexpr bool PAIR = ring_pairs(T); // (ring_pair.hpp)
    constexpr int LDSN = NNZB + NNZB / 32 + 2;
    __shared__ __attribute__((aligned(16))) double s_cx_raw[2 * LDSN]; // two arrays of doubles (plain path), or LDSN {coef, x} pairs
    double* const s_c = s_cx_raw;
    double* const s_x = s_cx_raw + LDSN;
    RingCx* const s_cx = reinterpret_cast<RingCx*>(s_cx_raw);
    __shared__ double s_ring[RING];
    __shared__ int4 s_plan[R4 * (MAXB + 2 * D + 2)];
    const int* s_rec = reinterpret_cast<const int*>(s_plan);
    const int tid = threadIdx.x;
    // The per-run tables are in DISPATCH order (mring_plan.hpp): entry x * per_xcd + j is what the j-th workgroup of XCD x runs — a
    // contiguous share of the matrix per XCD (neighbouring runs share its L2), its longest runs first, so that when there are more
    // runs than resident workgroups the late-comers are the short ones; nruns = 8 * per_xcd table entries, some of them empty.
    const int bid = (int)blockIdx.x, per_xcd = (nruns + kNXCD - 1) / kNXCD;
    const int gw = (bid & (kNXCD - 1)) * per_xcd + (bid >> 3);
    if ((bid >> 3) >= per_xcd || gw >= nruns) return leave(0);
    const int2 rng = run_rng[gw];
    const int b_begin = rng.x, nb = rng.y - rng.x; // <= MAXB by construction of the plan
    if (nb <= 0) return leave(0);
    const int clast = A.ncols - 1;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const RingComm nocomm{};
    const int wave = uni(tid >> 6), lane = tid & 63;

    for (int i = tid; i < R4 * nb; i += T) s_plan[i] = plan[R4 * (size_t)b_begin + i];
    __syncthreads();
    { // empty sentinel blocks behind the run (fixed number of loads per loop iteration, spmv_ring.hpp)
        const int4 l0 = s_plan[R4 * (nb - 1)];
        const int4 sent = make_int4(l0.x + l0.z, l0.y + l0.w, 0, 0);
        const int gc = min(max(0, A.ncols - 64), l0.x);
        for (int i = tid; i < 2 * D + 2; i += T) {
            s_plan[R4 * (nb + i)] = sent;
            s_plan[R4 * (nb + i) + 1] = make_int4(0, 0, 0, 0);
            s_plan[R4 * (nb + i) + 2] = make_int4(gc, gc, gc, gc);
            s_plan[R4 * (nb + i) + 3] = make_int4(gc, gc, gc, gc);
            s_plan[R4 * (nb + i) + 4] = make_int4(-1, -1, -1, -1);
        }
    }
    __syncthreads();
    const int run_kind = uni(run_ok[gw]); // 0: plain path for the whole run; 1: loop; 3: loop + PLAIN blocks behind it
    if (!(run_kind & 1)) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_plan[R4 * lb];
            ring_simple_block<T, NNZB, MAPPED, false>(A, x, y, m0.x, m0.y, m0.z, m0.w, s_c, s_x, nocomm);
        }
        return leave(-nb);
    }

    double c[D][PER];
    SlotVec sl[D];
    const SlotVec* slotv = reinterpret_cast<const SlotVec*>(slots);
    const int bslot_last = A.nblk - 1;
    int2 pr[D];
    double xr[D][2]; // groups `wave` and `wave + 4` of the staged block's new columns, this lane's entry
    int rm[D];

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_plan[R4 * lb];
        const int flags = uni(s_rec[kMringRec * lb + 4]);
        ring_load_coefs<T, PER, NT, PAIR>(c[s], A.coef + uni(m0.y), tid & ((flags & 1) ? -1 : 0));
        sl[s] = (slotv + (size_t)min(b_begin + lb, bslot_last) * T)[tid];
        const int* rp = A.ptrow + uni(m0.x) + tid;
        pr[s] = make_int2(rp[0], rp[1]);
        if (MAPPED) rm[s] = (A.rowmap + uni(m0.x))[tid];
        const int g0 = uni(s_rec[kMringRec * lb + 8 + wave]), g1 = uni(s_rec[kMringRec * lb + 12 + wave]);
        xr[s][0] = x[min(g0 + lane, clast)];
        xr[s][1] = x[min(g1 + lane, clast)];
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    { // the first block's windows, whole, from the run's record: all loads of a thread in flight at once
        constexpr int FILLW = (W + T - 1) / T;
        const int4 f0 = first[4 * (size_t)gw], f1 = first[4 * (size_t)gw + 1], f2 = first[4 * (size_t)gw + 2], f3 = first[4 * (size_t)gw + 3];
        const int lo[K] = {uni(f0.x), uni(f0.y), uni(f0.z), uni(f0.w), uni(f1.x)};
        const int cnt[K] = {uni(f1.y), uni(f1.z), uni(f1.w), uni(f2.x), uni(f2.y)};
        const int off[K] = {uni(f2.z), uni(f2.w), uni(f3.x), uni(f3.y), uni(f3.z)};
        double v[K][FILLW];
#pragma unroll
        for (int w = 0; w < K; w++)
#pragma unroll
            for (int u = 0; u < FILLW; u++) v[w][u] = x[max(0, min(lo[w] + tid + u * T, clast))];
#pragma unroll
        for (int w = 0; w < K; w++)
#pragma unroll
            for (int u = 0; u < FILLW; u++) {
                const int i = tid + u * T;
                int sidx = off[w] + i;
                if (sidx >= W) sidx -= W;
                if (i < cnt[w]) s_ring[w * W + sidx] = v[w][u];
            }
    }

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s; // lb >= nb: an empty sentinel block
            const int4 m0 = s_plan[R4 * lb];
            const int r0 = uni(m0.x), p0 = uni(m0.y), nrows = uni(m0.z);
            __syncthreads(); // the rings hold block lb's windows; staging is free again
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) xv[i] = s_ring[min((unsigned)sl[s][i], (unsigned)(RING - 1))];
            if (kRingMergedStage) ring_stage_cx<T, PER, SKEW, PAIR>(s_cx, c[s], xv, tid);
            else ring_stage<T, PER, SKEW, PAIR>(s_c, s_x, c[s], xv, tid);
            const int2 prs = pr[s];
            const int rms = MAPPED ? rm[s] : 0;
            issue(lb + D, s); // refill this stage with block lb + D
            __syncthreads(); // staging complete; nobody gathers block lb from the rings any more
            { // the groups of block lb + 1 (requested D blocks ago into stage (s+1)%D) into their slots
                const unsigned pk0 = (unsigned)uni(s_rec[kMringRec * (lb + 1) + 16 + (wave >> 1)]);
                const unsigned pk1 = (unsigned)uni(s_rec[kMringRec * (lb + 1) + 18 + (wave >> 1)]);
                const unsigned s0 = (pk0 >> (16 * (wave & 1))) & 0xffffu, s1 = (pk1 >> (16 * (wave & 1))) & 0xffffu;
                if (s0 != 0xffffu) s_ring[s0 + lane] = xr[(s + 1) % D][0];
                if (s1 != 0xffffu) s_ring[s1 + lane] = xr[(s + 1) % D][1];
            }
            if (tid < nrows) y[MAPPED ? rms : r0 + tid] = kRingMergedStage ? ring_row_chain_cx<8, SKEW>(s_cx, prs.x - p0, prs.y - p0) : ring_row_chain<8, SKEW>(s_c, s_x, prs.x - p0, prs.y - p0);
        }
    }
    // PLAIN blocks of this run, behind the loop (spmv_ring.hpp)
    if (!(run_kind & 2)) return leave(nb);
    for (int lb = 0; lb < nb; lb++) {
        if (uni(s_rec[kMringRec * lb + 4]) != 2) continue;
        const int4 m0 = s_plan[R4 * lb];
        ring_simple_block<T, NNZB, MAPPED, false>(A, x, y, uni(m0.x), uni(m0.y), uni(s_rec[kMringRec * lb + 6]), uni(m0.w), s_c, s_x, nocomm);
    }
    leave(nb);
}

} // namespace mi355
