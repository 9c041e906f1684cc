// spmv_bcsr4_ext.hpp — the one-launch push step of a BLOCKED (FE) rank: the ghosts staged once per step (round 5).
//
// The first form of this step (spmv_bcsr4_fused, removed) read every ghost use straight from the receive window.  The window is
// uncached and an FE node's entries are used by ~15 block rows: with a slab's two boundary planes (38 648 ghosts for 163 k rows at
// N = 8) that form took 39 us where four separate launches took 28.7, and it lost to this one at 13 784 and 7 944 ghosts too.
//
// The matrix is numbered as x_ext is, [owned | halo].  The grid, in dispatch order:
//   * n_work PUSH workgroups, one per 8 KB chunk of a link (push_exchange.hpp: kPushChunk): my entries into the neighbour's window with
//     write-through 16-byte stores, a drain, a relaxed ticket; the link's last chunk raises the flag.  No fence: an L2 write-back
//     under the running product cost 8 us per step.
//   * xwgs - n_work COPY workgroups: wait for every neighbour's flag of this step (relaxed polls, ONE acquire), copy the window into
//     `stage` — an ordinary, cached device buffer of the handle — with write-through stores, drain; the last of them (a ticket) writes
//     the step number to kExtReadyLines separate lines.
//   * the product's units (capi_part.hip builds the table): 64 block rows, four lanes per block row, spmv_bcsr4's loop — except for
//     units whose rows name a ghost node.  Those come LAST, poll ONE of the ready lines (all of them on one address delayed the word
//     by 8 us; bounded by spin_max, loud through `timeouts`), and are dealt as four workgroups with SIXTEEN lanes per block row
//     (bcsr4_ext_row16): they start ~10 us late and have a quarter of the load chain in front of them.
// A block column below n_local / 4 reads the caller's x, one above reads `stage` — a select of the base address, no branch.
//
// Why no cache invalidate is needed behind the wait: `stage` is read by nobody in this launch before the ready lines say so (only
// ghost-marked units name halo columns, and they wait first), the launch began with the usual invalidate, and the copy workgroups'
// stores are write-through and drained before the ticket — so no XCD's L2 and no CU's L1 can hold a line of it from before.
// (The halo part of x_ext itself would not do as the staging place: the 128-byte line where it begins also holds the last owned
// entries, which anybody may have read already.)
//
// Ranks that SHARE a card (a test arrangement, or more ranks than GPUs) must not wait in a quarter of their workgroups: four
// processes' waiting workgroups fill every wave slot of the card before the last process's exchange workgroups have one — seen as
// a give-up of all four (bench.py --gpus 4 --workload fe on one card).  There the step is TWO launches of this kernel: exchange +
// the units that name no ghost, then the units that do (nowait); capi_part.hip decides (a neighbour's window lives on this device).
//
// Same arithmetic and order as spmv_bcsr4 in both row forms: same bits.  Measured: profiles/NOTES.md R5.6.
#pragma once
#include "push_exchange.hpp"
#include "spmv_kernels.hpp"

namespace mi355 {

constexpr int kExtReadyLines = 64, kExtReadyStride = 32; // 128-byte lines

struct ExtComm {
    const PushLink* links;
    const int2* work;        // {link, chunk} items of the push
    const int* link_chunks;  // chunks per link
    unsigned* tickets;       // per link: chunks out so far
    const int* send_idx;
    const unsigned* flags;   // my window's flag slots
    const int* nb;
    const double* halo;      // my window's data, this step's parity
    double* stage;           // [n_halo] the cached copy the product reads
    unsigned* ready;         // [0] inbound workgroups done (a ticket), then kExtReadyLines lines of kExtReadyStride unsigneds: the step whose ghosts `stage` holds
    unsigned* timeouts;
    int n_work, n_nb, n_local, n_halo, xwgs; // workgroups [0, n_work) push, [n_work, xwgs) wait and copy, the rest multiply
    unsigned step, spin_max;
    int nowait;              // the launch holds only units behind a finished exchange (the two-launch form): nobody polls
    unsigned long long* trace; // devtools (TR): per workgroup {start, wait over, end} in s_memrealtime ticks
};

// 16 bytes that leave for their destination at once (sc0 sc1), as push_store's 8
typedef double ext_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void push_store2(double* p, double2 v)
{
    ext_v2d t = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
}

// a quarter of block row bi (lane q of its four) with P blocks in flight: spmv_bcsr4's loop, x blocks from the caller's x or from `stage`
template <int P>
__device__ __forceinline__ double bcsr4_ext_row(const Bcsr4View& A, const double* __restrict__ x, const double* __restrict__ xs, unsigned nbl, int bi, int q)
{
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int ia0 = A.ptrow[bi], ia1 = A.ptrow[bi + 1];
    auto xat = [&](unsigned c) { return reinterpret_cast<const double2*>((c < nbl ? x : xs) + 4 * (size_t)c); };
    double s = 0.0;
    if (ia0 < ia1) {
        const int last = ia1 - 1;
        const double* cq = A.coef + 4 * q;
        double2 a01[P], a23[P], x01[P], x23[P];
        unsigned cn[P];
#pragma unroll
        for (int t = 0; t < P; t++) {
            const int blk = min(ia0 + t, last);
            const double2* row = reinterpret_cast<const double2*>(cq + 16 * (size_t)blk);
            a01[t] = row[0];
            a23[t] = row[1];
            cn[t] = ucol[blk];
        }
#pragma unroll
        for (int t = 0; t < P; t++) {
            const double2* xb = xat(cn[t]);
            x01[t] = xb[0];
            x23[t] = xb[1];
        }
#pragma unroll
        for (int t = 0; t < P; t++) cn[t] = ucol[min(ia0 + P + t, last)];
        for (int ia = ia0; ia < ia1; ia += P) {
#pragma unroll
            for (int t = 0; t < P; t++) {
                const double2 c01 = a01[t], c23 = a23[t], v01 = x01[t], v23 = x23[t];
                const int nb = min(ia + t + P, last);
                const double2* nrow = reinterpret_cast<const double2*>(cq + 16 * (size_t)nb);
                a01[t] = nrow[0];
                a23[t] = nrow[1];
                const double2* nxb = xat(cn[t]);
                x01[t] = nxb[0];
                x23[t] = nxb[1];
                cn[t] = ucol[min(ia + t + 2 * P, last)];
                if (ia + t < ia1) {
                    s = fma(c01.x, v01.x, s);
                    s = fma(c01.y, v01.y, s);
                    s = fma(c23.x, v23.x, s);
                    s = fma(c23.y, v23.y, s);
                }
            }
        }
    }
    return s;
}

// The same quarter row by FOUR lanes (j = 0..3; 16 lanes per block row = one DPP row): lane j loads blocks j, j + 4, j + 8, ... — U of
// them in flight at once, a row of 4 U blocks in one round trip for the indices and one for the blocks — and the running sum goes round
// the four lanes in block order (row_ror:4), so the fma chain is the CSR row's chain, term by term: same bits.  A workgroup that
// starts late (it waited for the exchange) has a quarter of the load chain in front of it.
__device__ __forceinline__ double dpp_ror4(double v)
{
    const long long b = __double_as_longlong(v);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x124, 0xf, 0xf, false); // row_ror:4: lane i of a 16-lane row takes lane (i - 4) mod 16
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x124, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <int U>
__device__ __forceinline__ void bcsr4_ext_row16(const Bcsr4View& A, const double* __restrict__ x, const double* __restrict__ xs, unsigned nbl, int bi, int q, int j,
                                                double* __restrict__ y)
{
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int ia0 = A.ptrow[bi], ia1 = A.ptrow[bi + 1];
    auto xat = [&](unsigned c) { return reinterpret_cast<const double2*>((c < nbl ? x : xs) + 4 * (size_t)c); };
    double s = 0.0;
    const int last = ia1 - 1;
    const double* cq = A.coef + 4 * q;
    for (int base = ia0; base < ia1; base += 4 * U) {
        unsigned cn[U];
        double2 a01[U], a23[U], x01[U], x23[U];
#pragma unroll
        for (int u = 0; u < U; u++) cn[u] = ucol[min(base + 4 * u + j, last)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const double2* row = reinterpret_cast<const double2*>(cq + 16 * (size_t)min(base + 4 * u + j, last));
            a01[u] = row[0];
            a23[u] = row[1];
            const double2* xb = xat(cn[u]);
            x01[u] = xb[0];
            x23[u] = xb[1];
        }
        __builtin_amdgcn_sched_barrier(0); // every load of the segment is out before the chain starts (the scheduler would sink them into it, pair by pair)
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                if (j == jj && base + 4 * u + jj < ia1) {
                    s = fma(a01[u].x, x01[u].x, s);
                    s = fma(a01[u].y, x01[u].y, s);
                    s = fma(a23[u].x, x23[u].x, s);
                    s = fma(a23[u].y, x23[u].y, s);
                }
                const double t = dpp_ror4(s);
                s = (j == ((jj + 1) & 3)) ? t : s;
            }
        }
    }
    // the lane that took the row's last block holds the sum (behind it the value only travels on)
    if (j == ((ia1 - ia0 - 1) & 3) || (ia1 == ia0 && j == 0)) y[4 * (size_t)bi + q] = s;
}

// units[wg] = {first block row, mode}: mode bit 0 the workgroup waits for the exchange (its rows name ghost nodes), bit 1 it runs the
// 16-lanes-per-row form on 16 block rows (else 4 lanes per row, 64 block rows, P blocks in flight per thread)
// the exchange workgroups of a staged step (blockIdx.x < C.xwgs): push or wait + copy, as the header says; true: this workgroup was one
template <int T, bool TR>
__device__ __forceinline__ bool ext_exchange(const ExtComm& C, const double* __restrict__ x)
{
    const int tid = threadIdx.x;
    auto stamp = [&](int k) {
        if (TR && tid == 0) C.trace[3 * (size_t)blockIdx.x + k] = __builtin_amdgcn_s_memrealtime();
    };
    if ((int)blockIdx.x < C.n_work) { // ---- outbound: one {link, chunk} item of my entries into a neighbour's window
        if (!C.links) return true; // (devtools timing runs only)
        const int2 it = C.work[blockIdx.x]; // (write-through stores and a drain, no fence: an L2 write-back under the running product cost 8 us per step)
        const PushLink L = C.links[it.x];
        double* dst = (C.step & 1u) ? L.dst[1] : L.dst[0];
        const int i0 = it.y * kPushChunk, i1 = min(L.count, i0 + kPushChunk);
        if (L.first >= 0 && ((L.first | i0) & 1) == 0 && (((uintptr_t)dst | (uintptr_t)x) & 15) == 0) {
            const double2* s2 = reinterpret_cast<const double2*>(x + L.first + i0);
            const int n2 = (i1 - i0) >> 1;
            for (int i = tid; i < n2; i += T) push_store2(dst + i0 + 2 * i, s2[i]);
            if (((i1 - i0) & 1) && tid == 0) push_store(dst + i1 - 1, x[L.first + i1 - 1]);
        } else if (L.first >= 0) {
            for (int i = i0 + tid; i < i1; i += T) push_store(dst + i, x[L.first + i]);
        } else {
            for (int i = i0 + tid; i < i1; i += T) push_store(dst + i, x[C.send_idx[L.send_off + i]]);
        }
        push_drain(); // every storing wave: its stores have left for the peer
        __syncthreads();
        stamp(1);
        if (tid == 0) {
            const unsigned done = __hip_atomic_fetch_add(&C.tickets[it.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)done == C.link_chunks[it.x] - 1) { // the link's last chunk: every other chunk's workgroup drained before it took its ticket
                __hip_atomic_store(&C.tickets[it.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(L.flag, C.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        stamp(2);
        return true;
    }
    if ((int)blockIdx.x < C.xwgs) { // ---- inbound: the window into `stage` once every neighbour's entries of this step have landed
        const int cw = (int)blockIdx.x - C.n_work, ncw = C.xwgs - C.n_work;
        for (int j = tid; j < C.n_nb; j += T) { // (push_wait_flags with relaxed polls: an acquire per poll is a cache invalidate per poll, under the running product)
            const unsigned* f = C.flags + (size_t)C.nb[j] * kWinFlagStride;
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - C.step) < 0) {
                if (spins < 4096) __builtin_amdgcn_s_sleep(2);
                else __builtin_amdgcn_s_sleep(127);
                ++spins;
                if (spins == 64 && __hip_atomic_load(C.timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break; // an earlier wait gave up already
                if (spins > C.spin_max) {
                    __hip_atomic_fetch_add(C.timeouts, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // system scope, once: the data the flags announce
        stamp(1);
        if ((((uintptr_t)C.halo | (uintptr_t)C.stage) & 15) == 0) { // 16 bytes per access: the window is uncached, every load is a trip to memory
            const int n2 = C.n_halo >> 1;
            for (int i = cw * T + tid; i < n2; i += ncw * T) push_store2(C.stage + 2 * i, reinterpret_cast<const double2*>(C.halo)[i]);
            if ((C.n_halo & 1) && cw == 0 && tid == 0) push_store(C.stage + C.n_halo - 1, C.halo[C.n_halo - 1]);
        } else {
            for (int i = cw * T + tid; i < C.n_halo; i += ncw * T) push_store(C.stage + i, __builtin_nontemporal_load(C.halo + i));
        }
        push_drain();
        __syncthreads();
        // the last of them says so on kExtReadyLines separate lines, one store of one wave: the waiting workgroups poll one line each
        // (all of them on ONE address queued the memory channel behind it: the word came through 8 us late)
        __shared__ int s_last;
        if (tid == 0) {
            const unsigned done = __hip_atomic_fetch_add(C.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (int)done == ncw - 1;
            if (s_last) __hip_atomic_store(C.ready, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last && tid < kExtReadyLines) __hip_atomic_store(C.ready + kExtReadyStride * (1 + tid), C.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(2);
        return true;
    }
    return false;
}

// a workgroup whose rows name ghosts: wait (thread 0 polls ONE of the ready lines; bounded, loud) until `stage` holds this step's
__device__ __forceinline__ void ext_wait_ready(const ExtComm& C)
{
    if (threadIdx.x == 0) {
        const unsigned* line = C.ready + kExtReadyStride * (1 + ((int)blockIdx.x & (kExtReadyLines - 1)));
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - C.step) < 0) {
            if (spins < 4096) __builtin_amdgcn_s_sleep(1);
            else __builtin_amdgcn_s_sleep(127);
            ++spins;
            if (spins == 64 && __hip_atomic_load(C.timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break; // an earlier wait gave up already
            if (spins > C.spin_max) {
                __hip_atomic_fetch_add(C.timeouts, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
}

template <int P, int U, int T, bool TR = false>
__global__ __launch_bounds__(T) void spmv_bcsr4_fused_ext(Bcsr4View A, const double* __restrict__ x, double* __restrict__ y, ExtComm C, const int2* __restrict__ units)
{
    const int tid = threadIdx.x;
    auto stamp = [&](int k) {
        if (TR && tid == 0) C.trace[3 * (size_t)blockIdx.x + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    if (ext_exchange<T, TR>(C, x)) return;
    const int2 unit = units[(int)blockIdx.x - C.xwgs];
    const unsigned nbl = (unsigned)C.n_local >> 2;
    const double* xs = C.stage - (size_t)C.n_local; // block column c >= nbl: stage + 4 (c - nbl)
    if ((unit.y & 1) && !C.nowait) ext_wait_ready(C); // its rows name ghost nodes: `stage` must hold this step's
    stamp(1);
    if (unit.y & 2) {
        const int bi = unit.x + (tid >> 4), l = tid & 15;
        if (bi < A.nbrows) bcsr4_ext_row16<U>(A, x, xs, nbl, bi, l & 3, l >> 2, y);
    } else {
        const int bi = unit.x + (tid >> 2), q = tid & 3;
        if (bi < A.nbrows) y[4 * (size_t)bi + q] = bcsr4_ext_row<P>(A, x, xs, nbl, bi, q);
    }
    if (TR) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same step for SCALAR rows (a rank whose piece has no 4x4 structure and whose halo is too wide for the sliced stream's or the ring's
// fused forms — a 3-D mesh operator over ranks: a plane of ghosts each side): spmv_csr_stream's row blocks (spmv_kernels.hpp: <= NNZB
// nonzeros, coalesced loads, the x gather, one thread per row walking its LDS segment: the CSR fma chain) on the piece numbered
// [owned | halo], the gather taking columns >= n_local from `stage`.  order[i] = the block workgroup i (behind the exchange) takes,
// bit 31 set if its rows name a ghost: those come last and wait for the exchange.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NNZB, bool NT, bool TR = false>
__global__ __launch_bounds__(kWG) void spmv_csr_fused_ext(CsrView A, const double* __restrict__ x, double* __restrict__ y, ExtComm C, const unsigned* __restrict__ order)
{
    constexpr int PER = NNZB / kWG;
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];
    const int tid = threadIdx.x;
    auto stamp = [&](int k) {
        if (TR && tid == 0) C.trace[3 * (size_t)blockIdx.x + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    if (ext_exchange<kWG, TR>(C, x)) return;
    const unsigned ob = order[(int)blockIdx.x - C.xwgs];
    const int b = (int)(ob & 0x7fffffffu);
    if ((ob >> 31) && !C.nowait) ext_wait_ready(C);
    stamp(1);
    const unsigned nl = (unsigned)C.n_local;
    const double* xs = C.stage - (size_t)C.n_local; // column c >= n_local: stage[c - n_local]
    auto xat = [&](unsigned c) { return (c < nl ? x : xs)[c]; };
    const int2 d0 = A.blk[b];
    const int2 d1 = A.blk[b + 1];
    const int r0 = d0.x, p0 = d0.y, r1 = d1.x, p1 = d1.y;
    const int nn = p1 - p0;
    const int myrow = r0 + tid;
    if (nn == 0) { // a block of empty rows
        for (int r = myrow; r < r1; r += kWG) y[r] = 0.0;
    } else if (nn <= NNZB) { // (spmv_csr_stream's phases, comments there)
        const int last = nn - 1;
        const int rowc = min(myrow, r1 - 1);
        const int pa = A.ptrow[rowc];
        const int pe = A.ptrow[rowc + 1];
        const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
        double c[PER];
        unsigned j[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * kWG, last);
            if (NT) {
                c[i] = __builtin_nontemporal_load(&A.coef[p0 + k]);
                j[i] = __builtin_nontemporal_load(&ucol[p0 + k]);
            } else {
                c[i] = A.coef[p0 + k];
                j[i] = ucol[p0 + k];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        double xv[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) xv[i] = xat(j[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * kWG, last);
            s_c[sk(k)] = c[i];
            s_x[sk(k)] = xv[i];
        }
        __syncthreads();
        if (myrow < r1) y[myrow] = row_chain<8>(s_c, s_x, pa - p0, pe - p0);
        for (int r = myrow + kWG; r < r1; r += kWG) { // blocks of very short rows hold more than kWG rows
            const int ra = A.ptrow[r] - p0, re = A.ptrow[r + 1] - p0;
            y[r] = row_chain<8>(s_c, s_x, ra, re);
        }
    } else { // one row longer than a block: chunk by chunk, thread 0 carries the chain
        double s = 0.0;
        for (int base = p0; base < p1; base += NNZB) {
            const int m = min(NNZB, p1 - base);
            for (int k = tid; k < m; k += kWG) {
                s_c[sk(k)] = A.coef[base + k];
                s_x[sk(k)] = xat((unsigned)A.indcol[base + k]);
            }
            __syncthreads();
            if (tid == 0)
                for (int k = 0; k < m; k++) s = fma(s_c[sk(k)], s_x[sk(k)], s);
            __syncthreads();
        }
        if (tid == 0) y[r0] = s;
    }
    if (TR) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(2);
    }
}

} // namespace mi355
