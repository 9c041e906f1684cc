// spmv_mring.hpp — the multi-window ring kernel: spmv_csr_ring (spmv_ring.hpp) with K = 5 independent sliding x windows in
// LDS instead of one.  For 3-D mesh operators (the reference's scalar pressure operator, src/solve_newton.c), whose rows
// reach into three narrow column clusters two mesh planes apart: no single window can hold them, three small ones can, at
// any mesh size (mring_plan.hpp).  Everything but the refill is the ring kernel: persistent workgroups over runs of row
// blocks, matrix stream and new x columns D blocks ahead in registers, 16-bit precomputed LDS slots per nonzero, {coef, x}
// staged in LDS, one thread per row walking its segment with the sequential fma chain of the reference's SpMV_CSR_OPT/_FMA
// (mpk/SpMV.cpp:23-56) — bit-identical to every other kernel here.
//
// Refill: a block's new columns are the concatenation of up to K ranges [lo_w, lo_w + cnt_w) (one per window, from the plan
// record), each a whole number of 64-column groups — so a wave refills one window at a time and decodes its share of the
// record with scalar instructions.  Every wave prefetches four such groups D blocks ahead (1024 columns per block on this
// path; no extra memory traffic, no dependent load) and writes them to their windows' rings when the block becomes next.
// A block that would bring more starts a run (mring_plan.hpp), whose first block's windows are filled whole up front: the
// steady-state loop has no global load under a condition, which is what keeps the prefetched blocks in flight (spmv_ring.hpp, LEAN).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mring_plan.hpp"
#include "spmv_ring.hpp"

namespace mi355 {

// The refill part of a block's plan record, wave-uniform (SGPRs): m1 = {flags, total, lo[4], pk[4]}, lo = lo[0..3],
// pk = pk[0..3]; pk = count | base index << 11, counts are multiples of 64 (mring_plan.hpp).
struct MringRec { int4 m1, lo, pk; };
// Group g (64 consecutive entries of the concatenated new-column ranges) belongs to ONE window: everything about it is
// scalar; a lane adds its lane id.  slot < 0: the group lies past the record's last new column.
struct MringNew { int col, slot; };
__device__ __forceinline__ MringNew mring_decode(int g /* uniform */, int lane, const MringRec& R)
{
    // branch-free on purpose: written with ?: chains hipcc turns the (scalar) window selection into a dozen s_cbranch per
    // call, six calls per block — measured as +50 % on the whole kernel.  Masks keep it a straight run of SALU instructions.
    const int i0 = g * 64;
    const int lo[kMringK] = {R.lo.x, R.lo.y, R.lo.z, R.lo.w, R.m1.z};
    const unsigned pk[kMringK] = {(unsigned)R.pk.x, (unsigned)R.pk.y, (unsigned)R.pk.z, (unsigned)R.pk.w, (unsigned)R.m1.w};
    int first = 0, col0 = 0, s0 = 0, off = 0, hit = 0;
#pragma unroll
    for (int w = 0; w < kMringK; w++) {
        const int cnt = (int)(pk[w] & 2047u);
        const int in = -(int)((unsigned)(i0 - first) < (unsigned)cnt); // all ones if group g lies in window w's range
        const int c = lo[w] + (i0 - first);
        col0 |= in & c;
        s0 |= in & (c - (int)(pk[w] >> 11) * kMringW);
        off |= in & (w * kMringW);
        hit |= in;
        first += cnt;
    }
    MringNew r;
    // (a group past the record's last new column still loads — a fixed number of loads per block — but from the neighbourhood of
    // this workgroup's own columns: x[lane] for everybody is one cache line asked for by every wave of the GPU at once, measured
    // as +0.3 us per block and unused group)
    r.col = (hit ? col0 : lo[0] + i0) + lane;
    int sl = s0 + lane;
    if (sl >= kMringW) sl -= kMringW;
    r.slot = hit ? off + sl : -1;
    return r;
}

template <int T, int NNZB, int D, int MAXB, bool MAPPED, bool NT, bool SKEW>
__global__ __launch_bounds__(T) void spmv_csr_mring(CsrView A, const int4* __restrict__ plan, const int* __restrict__ run_ok,
                                                    const unsigned short* __restrict__ slots, const double* __restrict__ x,
                                                    double* __restrict__ y, const int2* __restrict__ run_rng)
{
    constexpr int K = kMringK, W = kMringW, RING = K * W, PER = NNZB / T;
    typedef unsigned short SlotVec __attribute__((ext_vector_type(PER)));
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    __shared__ double s_ring[RING];
    __shared__ int4 s_plan[4 * (MAXB + 2 * D + 2)];
    const int tid = threadIdx.x;
    const int bid = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int gw = (bid & (kNXCD - 1)) * (nwg / kNXCD) + (bid >> 3); // neighbouring runs share an XCD's L2
    const int2 rng = run_rng[gw];
    const int b_begin = rng.x, nb = rng.y - rng.x; // <= MAXB by construction of the plan
    if (nb <= 0) return;
    const int clast = A.ncols - 1;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni4 = [&](const int4& v) { return make_int4(uni(v.x), uni(v.y), uni(v.z), uni(v.w)); };
    const RingComm nocomm{};

    for (int i = tid; i < 4 * nb; i += T) s_plan[i] = plan[4 * (size_t)b_begin + i];
    __syncthreads();
    { // empty sentinel blocks behind the run (fixed number of loads per loop iteration, spmv_ring.hpp)
        const int4 l0 = s_plan[4 * (nb - 1)];
        const int4 sent = make_int4(l0.x + l0.z, l0.y + l0.w, 0, 0);
        for (int i = tid; i < 2 * D + 2; i += T) {
            s_plan[4 * (nb + i)] = sent;
            s_plan[4 * (nb + i) + 1] = make_int4(0, 0, 0, 0);
            s_plan[4 * (nb + i) + 2] = make_int4(0, 0, 0, 0);
            s_plan[4 * (nb + i) + 3] = make_int4(0, 0, 0, 0);
        }
    }
    __syncthreads();
    const int run_kind = uni(run_ok[gw]); // 0: plain path for the whole run; 1: loop; 3: loop + PLAIN blocks behind it
    if (!(run_kind & 1)) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_plan[4 * lb];
            ring_simple_block<T, NNZB, MAPPED, false>(A, x, y, m0.x, m0.y, m0.z, m0.w, s_c, s_x, nocomm);
        }
        return;
    }

    double c[D][PER];
    SlotVec sl[D];
    const SlotVec* slotv = reinterpret_cast<const SlotVec*>(slots);
    const int bslot_last = A.nblk - 1;
    int2 pr[D];
    constexpr int NX = kMringFast / T; // groups of 64 new columns a wave prefetches per block: 4 waves x NX x 64 columns in all
    double xr[D][NX];      // entries (wave + 4 j) * 64 + lane of the staged block's new columns
    int rm[D];
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_plan[4 * lb];
        MringRec R;
        R.m1 = uni4(s_plan[4 * lb + 1]);
        R.lo = uni4(s_plan[4 * lb + 2]);
        R.pk = uni4(s_plan[4 * lb + 3]);
        const double* cb = A.coef + uni(m0.y) + (tid & ((R.m1.x & 1) ? -1 : 0));
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (NT) c[s][i] = __builtin_nontemporal_load(&cb[i * T]);
            else c[s][i] = cb[i * T];
        }
        sl[s] = (slotv + (size_t)min(b_begin + lb, bslot_last) * T)[tid];
        const int* rp = A.ptrow + uni(m0.x) + tid;
        pr[s] = make_int2(rp[0], rp[1]);
        if (MAPPED) rm[s] = (A.rowmap + uni(m0.x))[tid];
#pragma unroll
        for (int j = 0; j < NX; j++) xr[s][j] = x[max(0, min(mring_decode(wave + (T / 64) * j, lane, R).col, clast))];
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    { // the first block's windows, whole: all loads of a thread in flight at once
        constexpr int FILLW = (W + T - 1) / T;
        const int4 m1 = uni4(s_plan[1]), m2 = uni4(s_plan[2]), m3 = uni4(s_plan[3]);
        const bool served = m1.x == 1; // a PLAIN first block keeps its row count where lo[0] lives
        const int lo[K] = {m2.x, m2.y, m2.z, m2.w, m1.z};
        const unsigned pk[K] = {served ? (unsigned)m3.x : 0u, served ? (unsigned)m3.y : 0u, served ? (unsigned)m3.z : 0u,
                                served ? (unsigned)m3.w : 0u, served ? (unsigned)m1.w : 0u};
        double v[K][FILLW];
#pragma unroll
        for (int w = 0; w < K; w++)
#pragma unroll
            for (int u = 0; u < FILLW; u++) v[w][u] = x[max(0, min(lo[w] + tid + u * T, clast))];
#pragma unroll
        for (int w = 0; w < K; w++) {
            const int cnt = (int)(pk[w] & 2047u), base = (int)(pk[w] >> 11) * W;
#pragma unroll
            for (int u = 0; u < FILLW; u++)
                if (tid + u * T < cnt) s_ring[w * W + ring_slot<W>(lo[w] + tid + u * T, base)] = v[w][u];
        }
    }

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s; // lb >= nb: an empty sentinel block
            const int4 m0 = s_plan[4 * lb];
            const int r0 = uni(m0.x), p0 = uni(m0.y), nrows = uni(m0.z);
            __syncthreads(); // the rings hold block lb's windows; staging is free again
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) xv[i] = s_ring[min((unsigned)sl[s][i], (unsigned)(RING - 1))];
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int k = SKEW ? sk(tid + i * T) : tid + i * T;
                s_c[k] = c[s][i];
                s_x[k] = xv[i];
            }
            const int2 prs = pr[s];
            const int rms = MAPPED ? rm[s] : 0;
            issue(lb + D, s); // refill this stage with block lb + D
            __syncthreads(); // staging complete; nobody gathers block lb from the rings any more
            { // new columns of block lb + 1 (requested D blocks ago into stage (s+1)%D) into their windows
                MringRec Q; // (re-read from LDS: keeping D records in SGPRs spills)
                Q.m1 = uni4(s_plan[4 * (lb + 1) + 1]);
                Q.lo = uni4(s_plan[4 * (lb + 1) + 2]);
                Q.pk = uni4(s_plan[4 * (lb + 1) + 3]);
                const int total = Q.m1.y;
#pragma unroll
                for (int j = 0; j < NX; j++) { // (the plan guarantees total <= NX * T inside a run: no unpipelined refill here)
                    const MringNew nw = mring_decode(wave + (T / 64) * j, lane, Q);
                    if (nw.slot >= 0) s_ring[nw.slot] = xr[(s + 1) % D][j];
                }
                (void)total;
            }
            if (tid < nrows) y[MAPPED ? rms : r0 + tid] = ring_row_chain<8, SKEW>(s_c, s_x, prs.x - p0, prs.y - p0);
        }
    }
    // PLAIN blocks of this run, behind the loop (spmv_ring.hpp)
    if (!(run_kind & 2)) return;
    for (int lb = 0; lb < nb; lb++) {
        const int4 m1 = s_plan[4 * lb + 1];
        if (uni(m1.x) != 2) continue;
        const int4 m0 = s_plan[4 * lb];
        ring_simple_block<T, NNZB, MAPPED, false>(A, x, y, uni(m0.x), uni(m0.y), uni(s_plan[4 * lb + 2].x), uni(m0.w), s_c, s_x, nocomm);
    }
}

} // namespace mi355
