// spmv_ring.hpp — the "ring" fp64 CSR SpMV kernel for gfx950 (MI355X): persistent
// workgroups, a sliding x window in LDS, the matrix stream pipelined D row blocks
// ahead in registers.  This is the kernel mi_spmv*() launches for banded /
// FE-ordered matrices; spmv_kernels.hpp holds the general fallbacks.
//
// Why this shape (all numbers measured on MI355X with tools/kbench on the 5 M row /
// 75 M nnz S15 matrix; profiles/NOTES.md §4.4 has the table):
//   * a CSR-stream kernel that gathers x[col] from global memory tops out near
//     3.0-3.4 TB/s algorithmic: 75 M eight-byte gathers are one L1 tag lookup each
//     and a 64/128-byte L2 line per miss, i.e. the gather, not HBM, is the bound;
//   * staging each row block's whole x window in LDS (non-sliding) moves 33 KB of x
//     per 24 KB of matrix and is slower still;
//   * so the window SLIDES: a workgroup walks a run of consecutive row blocks and
//     keeps x[cmin_b .. cmax_b] of the current block in an LDS ring indexed by
//     column.  Moving to block b+1 loads only the columns that entered the window
//     (about rows-per-block of them for a banded matrix), x is read from L2 ~once
//     per run, and every per-nonzero gather is a ds_read_b64.
//   * the per-block latencies (HBM for the stream, L2 for the new columns) are
//     taken off the critical path by running both D blocks ahead in registers.
//     hipcc only pipelines this if the steady-state loop is written so that its
//     waitcnt pass can COUNT: every load unconditional (addresses clamped, never
//     predicated), column ids unsigned (a signed id is sign-extended right behind
//     its load = a wait per load), raw ptrow values kept until use, a fixed number
//     of loads per iteration (empty sentinel blocks instead of "if (b < nb)"), and
//     NO global-memory fallback inside the loop (a structurised if/else makes every
//     later wait a full drain).  Runs the ring cannot serve are therefore detected
//     up front (host-side plan) and take the plain per-block path.
//
// Numerics: identical to every other kernel here — each row is one sequential fma
// chain in CSR order, bit-equal to the reference's SpMV_CSR_OPT/_FMA
// (mpk/SpMV.cpp:23-56).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "push_exchange.hpp"
#include "spmv_kernels.hpp"

namespace mi355 {

// Per-block plan records, built on the host once per matrix (ring_plan.hpp):
//   plan[2b]   = {first row, first nnz, rows, nnz}
//   plan[2b+1] = {first new column, number of new columns, ring base, flags}
// The ring holds column c at slot (c - base) mod RING, with base a multiple of
// RING chosen so that c - base is in [0, 2*RING) for every column of the window.
template <int RING>
__device__ __forceinline__ int ring_slot(int c, int base)
{
    const int p = c - base;
    return p >= RING ? p - RING : p;
}

// The fused multi-GPU step (FUSED = true; mi_part_spmv_push_dev): ONE launch does a rank's whole product.
//   * the peer push of push_exchange.hpp (this rank's entries into the neighbours' windows, then the flags) is done by the
//     ghost-touching runs themselves, FIRST thing, before they wait: those runs are short by construction (ring_plan.hpp,
//     kGhostRunSlack), so the ~3 us of a push (stores, system-scope fence, flag) fit into their slack and no workgroup is
//     added to the grid (with 8 extra push workgroups in front of 512 resident ones, eight ring workgroups started 3 us
//     late: 26.8 instead of 23 us per step at N = 8).  A push depends on nothing, so pushing before waiting cannot
//     deadlock.  push_wgs > 0 (extra workgroups in front) remains as the fallback for a plan without ghost runs;
//   * the others are the ring kernel over ALL local rows in their natural order, columns numbered [ghosts of lower
//     ranks | owned | ghosts of higher ranks] (partition.hpp: build_combined) so that a band stays a band across the
//     partition boundary and boundary rows are ring-served like the rest.  Ghost columns are read from this rank's
//     receive window (`halo`, uncached memory the neighbours' kernels write), owned ones from x.  A run that touches a
//     ghost (run_halo: the first and last few runs of a banded partition) first waits — bounded, loud — until every
//     neighbour's flag shows this step.
// Against the four-launch form (push, interior, wait + copy, boundary) this removes three launches of ~4 us each from
// a step whose whole interior kernel is 23 us at 8 ranks.
struct RingComm {
    const PushLink* links;
    const int* send_idx;
    const unsigned* flags; // my window's flag slots
    const int* nb;         // neighbours to wait for
    const double* halo;    // my window's data, this step's parity
    const int* run_halo;   // per run: touches a ghost column
    const int* run_link;   // per run: first push link this run serves (then every npush_runs-th), or -1
    unsigned* timeouts;    // host-visible
    int n_links, n_nb, n_local, n_left, push_wgs, npush_runs;
    int gate_push;         // this rank has no ghost reader: its push workgroups wait for flag >= step - 1 themselves
    unsigned step;
    unsigned spin_max; // polls before a wait gives up (2^kPushSpinLog2Default unless MI355_PUSH_SPIN_LOG2 says otherwise)
};

// The dot epilogue (DOT = true; mi_spmv_dot_dev / mi_spmv_orthogonalize_dev): while a row's value is still in its thread's
// register it is also multiplied into that thread's running sum of b[row] * y[row]; at the end of the run the workgroup
// reduces its threads' sums with a fixed tree and writes ONE partial — the finishing workgroup(s) of the consumer (the
// orthogonalize update, blas1_kernels.hpp) add the <= 512 partials in a fixed order.  Deterministic, not the CPU's
// left-to-right order (as every reduction of this library).  It replaces the separate dot pass between two products of a
// Krylov step (mpk/SpMVmulti.cpp:563-569): b is read once per row here (8 B) instead of y and b being read again (16 B) by a
// kernel of its own.  b's loads ride in the D-deep prefetch like every other load of the loop (unconditional, index clamped).
struct RingDot {
    const double* b;  // the dot's other vector, rows' numbering
    double* partial;  // one double per workgroup of the launch
};

// one push link by the T threads of a workgroup (push_exchange.hpp: halo_push_kernel's body)
template <int T>
__device__ __forceinline__ void ring_push_link(const RingComm& C, const double* __restrict__ x, int l)
{
    const int tid = threadIdx.x;
    const PushLink L = C.links[l];
    double* dst = (C.step & 1u) ? L.dst[1] : L.dst[0]; // (a select, not an indexed local array: that would live in scratch)
    if (L.first >= 0) {
        for (int i = tid; i < L.count; i += T) push_store(dst + i, x[L.first + i]);
    } else {
        for (int i = tid; i < L.count; i += T) push_store(dst + i, x[C.send_idx[L.send_off + i]]);
    }
    push_drain();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(L.flag, C.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Dedicated push workgroups (a plan without ghost runs; the blocked form): if NO run / workgroup of this rank's launch waits for
// the neighbours — the rank receives nothing, C.gate_push — the pushers themselves wait until every neighbour has flagged
// step - 1 before they overwrite this step's parity (push_exchange.hpp: THE INVARIANT).  A rank with ghosts skips this: its
// ghost readers of the previous step's launch waited already.
template <int T>
__device__ __forceinline__ void ring_push_gate(const RingComm& C)
{
    if (!C.gate_push) return;
    push_wait_flags(C.flags, C.nb, C.n_nb, C.step, 1u, C.timeouts, C.spin_max, threadIdx.x, T);
    __syncthreads();
}

template <bool FUSED>
__device__ __forceinline__ double ring_ldx(const double* __restrict__ x, const RingComm& C, int c)
{
    if (FUSED) { // [0, n_left) ghosts in front | owned | ghosts behind (halo order: front ghosts first)
        const int o = c - C.n_left;
        return *(o < 0 ? C.halo + c : (o < C.n_local ? x + o : C.halo + (c - C.n_local)));
    }
    return x[c];
}

// Plain per-block SpMV (global gather, any row length): the path of runs the ring
// cannot serve.  Same arithmetic, no pipelining.
template <int T, int NNZB, bool MAPPED, bool FUSED>
__device__ __forceinline__ void ring_simple_block(const CsrView& A, const double* __restrict__ x,
                                                  double* __restrict__ y, int r0, int p0, int nrows, int nn,
                                                  double* s_c, double* s_x, const RingComm& C)
{
    const int tid = threadIdx.x;
    __syncthreads();
    if (nn > NNZB) { // a single row longer than a block: chunked, chain carried by thread 0
        double sacc = 0.0;
        for (int bs = p0; bs < p0 + nn; bs += NNZB) {
            const int m = min(NNZB, p0 + nn - bs);
            __syncthreads();
            for (int k = tid; k < m; k += T) {
                s_c[sk(k)] = A.coef[bs + k];
                s_x[sk(k)] = ring_ldx<FUSED>(x, C, A.indcol[bs + k]);
            }
            __syncthreads();
            if (tid == 0)
                for (int k = 0; k < m; k++) sacc = fma(s_c[sk(k)], s_x[sk(k)], sacc);
        }
        if (tid == 0) y[MAPPED ? A.rowmap[r0] : r0] = sacc;
        return;
    }
    for (int k = tid; k < nn; k += T) {
        s_c[sk(k)] = A.coef[p0 + k];
        s_x[sk(k)] = ring_ldx<FUSED>(x, C, A.indcol[p0 + k]);
    }
    __syncthreads();
    for (int r = r0 + tid; r < r0 + nrows; r += T) {
        const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
        y[MAPPED ? A.rowmap[r] : r] = row_chain<8>(s_c, s_x, a, e);
    }
}

// Row chain over the staged {coef, x} of one row, operands fetched U at a time.  SKEW: the
// staging arrays carry one pad slot per 32 entries (sk(), spmv_kernels.hpp) — needed when many
// rows have a length that is a multiple of 8 (neighbouring lanes' segments then start a multiple of
// 64 B apart and pile onto two LDS banks), at the price of index arithmetic per term.  Without it a
// row is a plain contiguous segment: full batches of U are read with immediate offsets and run
// unpredicated, only the tail (< U terms) tests against the row end; reads past the row end stay
// inside the (padded) staging array and are never used.  (S15 rows, ISA of the chain phase:
// ~170 VALU instructions per row with the skewed form, ~45 without.)
template <int U, bool SKEW>
__device__ __forceinline__ double ring_row_chain(const double* s_c, const double* s_x, int ra, int re)
{
    if (SKEW) return row_chain<U>(s_c, s_x, ra, re);
    double s = 0.0;
    int k = ra;
    for (; k + U <= re; k += U) {
        double cc[U], xx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            cc[u] = s_c[k + u];
            xx[u] = s_x[k + u];
        }
#pragma unroll
        for (int u = 0; u < U; u++) s = fma(cc[u], xx[u], s);
    }
    if (k < re) {
        double cc[U], xx[U];
#pragma unroll
        for (int u = 0; u < U - 1; u++) {
            cc[u] = s_c[k + u];
            xx[u] = s_x[k + u];
        }
#pragma unroll
        for (int u = 0; u < U - 1; u++)
            if (k + u < re) s = fma(cc[u], xx[u], s);
    }
    return s;
}

// (ring_load_coefs / ring_stage — the value stream into registers and on into the staging arrays, 16 bytes per lane where the
// thread owns nonzero pairs — live in spmv_kernels.hpp beside the row chains; spmk_ring.hpp and spmv_mring.hpp use them too.)
// T threads, NNZB nonzeros per row block (PER = NNZB/T per thread), RING doubles of
// x window, D blocks of prefetch, runs of at most MAXB blocks per workgroup.
// LDS = 16 B * (NNZB + NNZB/32) staging + 8 B * RING + 32 B * (MAXB + 2D + 2) plan.
//
// The column stream is read from `slots`, not from A.indcol: the ring slot of every nonzero as a
// 16-bit number, precomputed with the plan (ring_plan.hpp: build_ring_slots) and stored per
// block in thread order, so that one 2*PER-byte load hands a thread all its PER slots.  The
// matrix stream is 10 instead of 12 bytes per nonzero and PER+1 instead of 2*PER loads per
// thread and block; the arithmetic (and so every bit of y) is that of the CSR arrays.
//
// NT: the matrix values are loaded non-temporally.  A matrix much larger than the 256 MB Infinity
// Cache is read once per product, and kept out of the L2 / Infinity Cache replacement it stops
// displacing x, the plan and the y lines being written (C4: 190 -> 169 us).  A matrix that fits
// the cache is better served by it across repeated products (C2: 38 us temporal, 44 us NT), so
// mi_csr_create times both and keeps the faster.
//
// LEAN: the plan guarantees that no block of a served run (but a run's first) brings more than T new columns and that no
// block holds more than T rows — then the steady-state loop has NO global load under a condition (no unpipelined refill, no
// second pass over the rows), and that is what lets hipcc's waitcnt pass count across iterations: with either fallback
// compiled in, every block's row chains sit behind an s_waitcnt vmcnt(0) — the loads just issued for block lb + D are
// drained before block lb is reduced, so the "D blocks ahead" are never in flight together (ISA of the general form: vmcnt(0)
// in front of the chains of every stage; of the LEAN form: vmcnt(33) with D = 4, one full drain per D blocks at the loop
// header).  mi_csr_create launches the LEAN instantiation whenever the plan allows (all natural-order bands do).
template <int T, int NNZB, int RING, int D, int MAXB, bool MAPPED, bool NT, bool SKEW, bool FUSED = false, bool LEAN = false, bool DOT = false>
__global__ __launch_bounds__(T) void spmv_csr_ring(CsrView A, const int4* __restrict__ plan,
                                                   const int* __restrict__ run_ok,
                                                   const unsigned short* __restrict__ slots,
                                                   const double* __restrict__ x, double* __restrict__ y,
                                                   const int2* __restrict__ run_rng, int bpw, RingComm C, RingDot Dt = RingDot{nullptr, nullptr})
{
    static_assert(!DOT || (LEAN && !FUSED && !MAPPED), "the dot epilogue exists for the LEAN single-GPU instantiation (all rows inside the counted loop)");
    constexpr int PER = NNZB / T;
    constexpr bool PAIR = ring_pairs(T);
    typedef unsigned short SlotVec __attribute__((ext_vector_type(PER)));
    static_assert(RING <= 65536, "ring slots are stored in 16 bits");
    constexpr int LDSN = NNZB + NNZB / 32 + 2;
    __shared__ __attribute__((aligned(16))) double s_cx_raw[2 * LDSN]; // two arrays of doubles (plain path), or LDSN {coef, x} pairs
    double* const s_c = s_cx_raw;
    double* const s_x = s_cx_raw + LDSN;
    RingCx* const s_cx = reinterpret_cast<RingCx*>(s_cx_raw);
    __shared__ double s_ring[RING];
    __shared__ int4 s_plan[2 * (MAXB + 2 * D + 2)];
    const int tid = threadIdx.x;
    if (FUSED && (int)blockIdx.x < C.push_wgs) { // fallback: dedicated push workgroups in front of the grid
        if ((int)blockIdx.x < C.n_links) ring_push_gate<T>(C); // (uniform per workgroup)
        for (int l = blockIdx.x; l < C.n_links; l += C.push_wgs) ring_push_link<T>(C, x, l);
        return;
    }
    // XCD-aware run order: workgroups with equal (blockIdx & 7) share an XCD (observed
    // round-robin dispatch, a speed assumption only) and get neighbouring runs, so
    // the x columns one run loads are L2 hits for the next.
    const int bid = FUSED ? (int)blockIdx.x - C.push_wgs : (int)blockIdx.x;
    const int nwg = FUSED ? (int)gridDim.x - C.push_wgs : (int)gridDim.x;
    const int gw = (bid & (kNXCD - 1)) * (nwg / kNXCD) + (bid >> 3);
    // this workgroup's run: blocks [rng.x, rng.y).  Plans with consecutive runs of bpw blocks (everything but the fused
    // multi-GPU piece) say so with bpw > 0 and spare the kernel a dependent global load in front of its plan loads.
    const int2 rng = bpw > 0 ? make_int2(min(A.nblk, gw * bpw), min(A.nblk, (gw + 1) * bpw)) : run_rng[gw];
    const int b_begin = rng.x;
    const int nb = rng.y - rng.x; // <= MAXB by construction of the plan
    if (FUSED && C.npush_runs > 0) { // push duty of this run, before anything that could wait
        const int l0 = C.run_link[gw];
        if (l0 >= 0)
            for (int l = l0; l < C.n_links; l += C.npush_runs) ring_push_link<T>(C, x, l);
    }
    if (nb <= 0) {
        if (DOT && tid == 0) Dt.partial[gw] = 0.0; // every workgroup of the launch owns one partial
        return;
    }
    const int clast = A.ncols - 1;
    // the plan is read back from LDS at a uniform address: tell the compiler so (SGPRs, scalar
    // address arithmetic for the stream loads instead of 64-bit vector adds per load)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

    for (int i = tid; i < 2 * nb; i += T) s_plan[i] = plan[2 * b_begin + i];
    __syncthreads();
    // empty sentinel blocks behind the run: they keep the number of loads per loop
    // iteration fixed; their (clamped) addresses stay inside the padded arrays
    {
        const int4 l0 = s_plan[2 * (nb - 1)];
        const int4 sent = make_int4(l0.x + l0.z, l0.y + l0.w, 0, 0);
        for (int i = tid; i < 2 * D + 2; i += T) {
            s_plan[2 * (nb + i)] = sent;
            s_plan[2 * (nb + i) + 1] = make_int4(0, 0, 0, 0);
        }
    }
    __syncthreads();
    if (FUSED && C.run_halo[gw]) { // this run reads ghosts: every neighbour's entries of this step must have landed
        push_wait_flags(C.flags, C.nb, C.n_nb, C.step, 0u, C.timeouts, C.spin_max, tid, T);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    const int run_kind = uni(run_ok[gw]); // 0: plain path for the whole run; 1: ring loop; 3: ring loop + PLAIN blocks behind it
    if (!(run_kind & 1)) {
        for (int lb = 0; lb < nb; lb++) {
            const int4 m0 = s_plan[2 * lb];
            ring_simple_block<T, NNZB, MAPPED, FUSED>(A, x, y, m0.x, m0.y, m0.z, m0.w, s_c, s_x, C);
        }
        return;
    }

    double c[D][PER];
    SlotVec sl[D];
    const SlotVec* slotv = reinterpret_cast<const SlotVec*>(slots);
    const int bslot_last = A.nblk - 1;
    int2 pr[D];   // raw ptrow pair of this thread's row in the staged block
    double xr[D]; // the column this thread puts into the ring when the staged block becomes current
    int rm[D];    // rowmap[row] (MAPPED only)
    double br[D]; // b[row] of this thread's row (DOT only)
    double dacc = 0.0;
    const int rlast = A.n - 1;

    auto issue = [&](int lb, int s) {
        const int4 m0 = s_plan[2 * lb], m1 = s_plan[2 * lb + 1];
        // plain tid-strided addresses, no clamps: lanes past the block's last nonzero / last row
        // read what lies behind it (initialised padding at the very end, ring_plan.hpp:
        // kRingPadNnz / kRingPadRows) and their values are never used
        // (a sentinel block behind the run — flags 0 — collapses to one address per load instead of
        // streaming 16 KB of the next run's values nobody uses: 2 % of the kernel's HBM traffic)
        ring_load_coefs<T, PER, NT, PAIR>(c[s], A.coef + uni(m0.y), tid & ((uni(m1.w) & 1) ? -1 : 0));
        sl[s] = (slotv + (size_t)min(b_begin + lb, bslot_last) * T)[tid];
        const int* rp = A.ptrow + uni(m0.x) + tid;
        pr[s] = make_int2(rp[0], rp[1]);
        if (MAPPED) rm[s] = (A.rowmap + uni(m0.x))[tid];
        if (DOT) br[s] = Dt.b[min(uni(m0.x) + tid, rlast)];
        xr[s] = ring_ldx<FUSED>(x, C, min(uni(m1.x) + tid, clast));
    };

#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    { // the whole first window (up to RING columns): all of a thread's loads in flight at once — one
      // round trip instead of one per T columns, which at ~1 us each is a visible share of a short
      // run (a 1 M-row matrix gives each workgroup 14 blocks, ~30 us in all)
        constexpr int FILL = (RING + T - 1) / T;
        const int4 q = s_plan[1];
        const int c0 = uni(q.x) + tid, cend = uni(q.x) + uni(q.y), qz = uni(q.z);
        double v[FILL];
#pragma unroll
        for (int u = 0; u < FILL; u++) v[u] = ring_ldx<FUSED>(x, C, min(c0 + u * T, clast));
#pragma unroll
        for (int u = 0; u < FILL; u++)
            if (c0 + u * T < cend) s_ring[ring_slot<RING>(c0 + u * T, qz)] = v[u];
    }

    for (int g = 0; g < nb; g += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            const int lb = g + s; // lb >= nb: an empty sentinel block
            const int4 m0 = s_plan[2 * lb];
            const int r0 = uni(m0.x), p0 = uni(m0.y), nrows = uni(m0.z);
            __syncthreads(); // ring holds block lb's window; staging is free again
            // ---- gather from the ring, park {coef, x} in the staging arrays.  Every thread writes
            // its PER fixed slots; slots past the block's last nonzero receive copies of that
            // nonzero (the clamped loads) and are never read by a row chain.
            double xv[PER];
#pragma unroll
            for (int i = 0; i < PER; i++) xv[i] = s_ring[min((unsigned)sl[s][i], (unsigned)(RING - 1))]; // slots are < RING by construction
            if (kRingMergedStage) ring_stage_cx<T, PER, SKEW, PAIR>(s_cx, c[s], xv, tid);
            else ring_stage<T, PER, SKEW, PAIR>(s_c, s_x, c[s], xv, tid);
            const int2 prs = pr[s];
            const int rms = MAPPED ? rm[s] : 0;
            const double brs = DOT ? br[s] : 0.0;
            // ---- refill this stage with block lb + D
            issue(lb + D, s);
            __syncthreads(); // staging complete; nobody gathers block lb from the ring any more
            // ---- ring entries of block lb + 1 (requested D blocks ago into stage (s+1)%D).
            // They overwrite only columns behind lb+1's window, which lb's gather is done with.
            {
                const int4 q4 = s_plan[2 * (lb + 1) + 1];
                const int qx = uni(q4.x), qy = uni(q4.y), qz = uni(q4.z);
                const double xn = xr[(s + 1) % D];
                if (LEAN || qy <= T) {
                    if (tid < qy) s_ring[ring_slot<RING>(qx + tid, qz)] = xn;
                } else { // more than T new columns at once: a window restart, or the ragged edge of a relabelled band.  Four
                         // loads in flight per thread and round trip (one per round trip made this path 4x as long)
                    for (int c0 = qx + tid; c0 < qx + qy; c0 += 4 * T) {
                        double v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) v[u] = ring_ldx<FUSED>(x, C, min(c0 + u * T, clast));
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (c0 + u * T < qx + qy) s_ring[ring_slot<RING>(c0 + u * T, qz)] = v[u];
                    }
                }
            }
            // ---- row chains
            if (DOT) {
                if (tid < nrows) {
                    const double yv = kRingMergedStage ? ring_row_chain_cx<8, SKEW>(s_cx, prs.x - p0, prs.y - p0) : ring_row_chain<8, SKEW>(s_c, s_x, prs.x - p0, prs.y - p0);
                    y[r0 + tid] = yv;
                    dacc = fma(brs, yv, dacc);
                }
            } else if (tid < nrows) y[MAPPED ? rms : r0 + tid] = kRingMergedStage ? ring_row_chain_cx<8, SKEW>(s_cx, prs.x - p0, prs.y - p0) : ring_row_chain<8, SKEW>(s_c, s_x, prs.x - p0, prs.y - p0);
            if (!LEAN)
                for (int r = r0 + tid + T; r < r0 + nrows; r += T) { // blocks of very short rows
                    const int a = A.ptrow[r] - p0, e = A.ptrow[r + 1] - p0;
                    y[MAPPED ? A.rowmap[r] : r] = kRingMergedStage ? ring_row_chain_cx<8, SKEW>(s_cx, a, e) : ring_row_chain<8, SKEW>(s_c, s_x, a, e);
                }
        }
    }
    if (DOT) { // this workgroup's partial of b . y: wave shuffle tree, then the T / 64 waves through LDS, fixed order
        __syncthreads(); // the staging arrays are free
        double v = dacc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) s_c[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            double t = s_c[0];
            for (int w = 1; w < T / 64; w++) t += s_c[w];
            Dt.partial[gw] = t;
        }
        return; // (the host launches this instantiation only for plans without PLAIN blocks or plain runs)
    }
    // PLAIN blocks of this run (ring_plan.hpp: a row the window cannot hold, at most kRingMaxPlain per run): the loop above
    // passed over them as over empty blocks; here, outside the counted pipeline, with direct gathers
    if (!(run_kind & 2)) return; // (almost every run: walking the records for nothing is ~5 us behind a 72-block run)
    for (int lb = 0; lb < nb; lb++) {
        const int4 m1 = s_plan[2 * lb + 1];
        if (uni(m1.w) != 2) continue;
        const int4 m0 = s_plan[2 * lb];
        ring_simple_block<T, NNZB, MAPPED, FUSED>(A, x, y, uni(m0.x), uni(m0.y), uni(m1.x), uni(m0.w), s_c, s_x, C);
    }
}

// (The fused multi-GPU step on the BLOCKED matrix lives in spmv_bcsr4_ext.hpp.)

} // namespace mi355
