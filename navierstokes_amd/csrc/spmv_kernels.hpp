// spmv_kernels.hpp — fp64 CSR SpMV kernels for gfx950 (MI355X, CDNA4).
//
// The operation is y = A x for the reference's csrmatrix (mpk/SpMV.h:18-24),
// i.e. SpMV_CSR* of mpk/SpMV.cpp:6-85.  The path is HBM-bound (12 B of matrix
// per 2 flop), so everything here is about streaming coef/indcol with fully
// coalesced reads and keeping the x gather and the per-row reduction off the
// critical path.  No MFMA: there is no dense contraction in a 15-nnz row.
//
// Numerics contract (all kernels): each row is ONE sequential fma chain in CSR
// order — s = fma(coef[k], x[indcol[k]], s) — bit-identical to the reference's
// SpMV_CSR_OPT / SpMV_CSR_FMA object code (mpk/SpMV.cpp:23-56).  A tree
// reduction would be a few % cheaper in LDS traffic but would give up
// bit-parity with the reference; the row chain is hidden under the HBM stream.
//
// Kernel "stream" (the CSR-stream shape): a 256-thread workgroup owns a block
// of consecutive rows holding <= NNZB nonzeros (row-block table built once at
// mi_csr_create).  Phase 1: all threads stream the block's coef/indcol range
// with unit-stride loads (the range is contiguous in CSR), gather x, and park
// {coef, x[col]} in LDS.  Phase 2: one thread per row walks its LDS segment
// with the fma chain and stores y (coalesced, rows are consecutive).
// XCD-aware block order: blockIdx -> (xcd = b & 7, slot = b >> 3) -> row block
// xcd * per_xcd + slot, so each XCD (own 4 MiB L2) sweeps one contiguous 1/8 of
// the rows and the x window it gathers from stays in ITS L2 instead of being
// pulled into all eight.
//
// The gather is what bounds this kernel (~3.0-3.4 TB/s algorithmic on MI355X): for
// matrices whose column window fits LDS the ring kernel of spmv_ring.hpp, which
// serves the gather from a sliding LDS window, is the one that gets launched.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ring_pair.hpp"

namespace mi355 {

constexpr int kWG = 256;        // threads per workgroup (4 waves of 64)
constexpr int kNXCD = 8;        // XCDs on MI355X, each with a private L2

// Device view of a CSR matrix plus its row-block table.
struct CsrView {
    int n;                 // rows
    int ncols;             // length of x
    const int* ptrow;      // [n+1]
    const int* indcol;     // [nnz]
    const double* coef;    // [nnz]
    const int* rowmap;     // [n] or nullptr: row r writes y[rowmap[r]]
    const int2* blk;       // [nblk+1] {first row, first nnz}; blk[nblk] = {n, nnz} (stream kernel)
    const int2* blk_span;  // unused by the product kernels (kept for tools/kbench experiments)
    int nblk;
};

// LDS index skew: one extra slot every 32 entries, so that per-row walks with a
// power-of-two stride (FE rows are 16/32/48/56 long) do not land on one bank pair.
__device__ __forceinline__ int sk(int k) { return k + (k >> 5); }

__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    const int per = (nblk + kNXCD - 1) / kNXCD;
    return (bid & (kNXCD - 1)) * per + (bid >> 3);
}

// Chunked form: the work is cut into chunks of `chunk` consecutive workgroups, dealt to the XCDs round-robin.  Each XCD
// (blockIdx & 7) then walks chunks c = xcd, xcd + 8, ... — contiguous work inside a chunk (its L2 reuses the x neighbourhood)
// while all eight XCDs stay in the same region of the matrix at any time (one moving front through memory, as in dispatch
// order) instead of eight distant streams.  Only whole rounds of 8 chunks are remapped; the tail keeps its dispatch order, so
// the grid needs no padding.
__device__ __forceinline__ int xcd_remap_chunked(int bid, int n, int chunk)
{
    const int span = kNXCD * chunk;
    if (bid >= (n / span) * span) return bid;
    const int xcd = bid & (kNXCD - 1), slot = bid >> 3;
    const int round = slot / chunk, within = slot - round * chunk;
    return (round * kNXCD + xcd) * chunk + within;
}

// ---------------------------------------------------------------------------
// stream kernel.  NNZB: nonzeros per row block (LDS = 2 * 8 B * sk(NNZB)).
// ---------------------------------------------------------------------------
// Row chain with batched operand fetch: the fma chain of a row stays strictly
// sequential, but its LDS operands are fetched U at a time (2U ds_reads in
// flight, one wait) instead of one wait per term.  Slots past the row end are
// clamped to the last valid slot and their fma is skipped, so the body has no
// divergent loads.
template <int U>
__device__ __forceinline__ double row_chain(const double* s_c, const double* s_x, int ra, int re)
{
    double s = 0.0;
    for (int k0 = ra; k0 < re; k0 += U) {
        double cc[U], xx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = min(k0 + u, re - 1);
            cc[u] = s_c[sk(k)];
            xx[u] = s_x[sk(k)];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (k0 + u < re) s = fma(cc[u], xx[u], s);
    }
    return s;
}

// The value stream of one block into a thread's registers, and from there (with the x values gathered from the ring) into the
// staging arrays.  PAIR (ring_pair.hpp, configuration 4): the thread owns nonzero pairs and loads them 16 bytes at a time — a
// block starts at any nonzero, so the pairs are only 8-byte aligned in memory (global loads need dword alignment only); staged,
// a pair is 16-byte aligned in the plain layout (one ds_write_b128) and two 8-byte writes in the skewed one.
// `lane` = tid, or 0 for a sentinel block behind the run (one address per load instead of 16 KB of values nobody uses).
typedef double RingCoef2 __attribute__((ext_vector_type(2), aligned(8)));
typedef double RingLds2 __attribute__((ext_vector_type(2)));

template <int T, int PER, bool NT, bool PAIR>
__device__ __forceinline__ void ring_load_coefs(double (&c)[PER], const double* __restrict__ block_base, int lane)
{
    if (PAIR) {
        static_assert(!PAIR || PER % 2 == 0, "pairs");
        const RingCoef2* cb = reinterpret_cast<const RingCoef2*>(block_base) + lane;
#pragma unroll
        for (int i = 0; i < PER / 2; i++) {
            RingCoef2 v;
            if (NT) v = __builtin_nontemporal_load(&cb[i * T]);
            else v = cb[i * T];
            c[2 * i] = v.x;
            c[2 * i + 1] = v.y;
        }
    } else {
        const double* cb = block_base + lane;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (NT) c[i] = __builtin_nontemporal_load(&cb[i * T]);
            else c[i] = cb[i * T];
        }
    }
}

template <int T, int PER, bool SKEW, bool PAIR>
__device__ __forceinline__ void ring_stage(double* s_c, double* s_x, const double (&c)[PER], const double (&xv)[PER], int tid)
{
    if (PAIR) {
#pragma unroll
        for (int i = 0; i < PER / 2; i++) {
            const int k0 = 2 * (tid + i * T);
            if (SKEW) { // k0 is even: k0 and k0 + 1 share their group of 32, so their padded slots are neighbours
                const int k = sk(k0);
                s_c[k] = c[2 * i];
                s_c[k + 1] = c[2 * i + 1];
                s_x[k] = xv[2 * i];
                s_x[k + 1] = xv[2 * i + 1];
            } else {
                *reinterpret_cast<RingLds2*>(s_c + k0) = RingLds2{c[2 * i], c[2 * i + 1]};
                *reinterpret_cast<RingLds2*>(s_x + k0) = RingLds2{xv[2 * i], xv[2 * i + 1]};
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = SKEW ? sk(tid + i * T) : tid + i * T;
            s_c[k] = c[i];
            s_x[k] = xv[i];
        }
    }
}

// MERGED staging (ring_pair.hpp: kRingMergedStage): the staging area holds {coef, x} PAIRS, 16 bytes per nonzero, instead of two arrays
// of doubles — a term of a row chain is then ONE 16-byte-aligned ds_read_b128 where the two-array form reads its operands as
// ds_read2_b64 pairs at 8-byte alignment (a row starts at any nonzero), which cost four times the LDS cycles per byte
// (MI355X_MICROARCH.md, LDS).  With rows of odd length the lanes' 16-byte slots fall on distinct banks (stride 16 L mod 256).
typedef double RingCx __attribute__((ext_vector_type(2)));

template <int T, int PER, bool SKEW, bool PAIR>
__device__ __forceinline__ void ring_stage_cx(RingCx* s_cx, const double (&c)[PER], const double (&xv)[PER], int tid)
{
    if (PAIR) {
#pragma unroll
        for (int i = 0; i < PER / 2; i++) {
            const int k0 = 2 * (tid + i * T), k = SKEW ? sk(k0) : k0; // (k0 even: its neighbour shares the group of 32)
            s_cx[k] = RingCx{c[2 * i], xv[2 * i]};
            s_cx[k + 1] = RingCx{c[2 * i + 1], xv[2 * i + 1]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < PER; i++) s_cx[SKEW ? sk(tid + i * T) : tid + i * T] = RingCx{c[i], xv[i]};
    }
}

template <int U, bool SKEW>
__device__ __forceinline__ double ring_row_chain_cx(const RingCx* s_cx, int ra, int re)
{
    double s = 0.0;
    if (SKEW) {
        for (int k0 = ra; k0 < re; k0 += U) {
            RingCx v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = s_cx[sk(min(k0 + u, re - 1))];
#pragma unroll
            for (int u = 0; u < U; u++)
                if (k0 + u < re) s = fma(v[u].x, v[u].y, s);
        }
        return s;
    }
    int k = ra;
    for (; k + U <= re; k += U) {
        RingCx v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = s_cx[k + u];
#pragma unroll
        for (int u = 0; u < U; u++) s = fma(v[u].x, v[u].y, s);
    }
    if (k < re) {
        RingCx v[U];
#pragma unroll
        for (int u = 0; u < U - 1; u++) v[u] = s_cx[k + u];
#pragma unroll
        for (int u = 0; u < U - 1; u++)
            if (k + u < re) s = fma(v[u].x, v[u].y, s);
    }
    return s;
}

// NT: matrix stream loaded non-temporally (see spmv_ring.hpp; chosen per matrix by mi_csr_create)
template <int NNZB, bool NT = false>
__global__ __launch_bounds__(kWG) void spmv_csr_stream(CsrView A, const double* __restrict__ x,
                                                       double* __restrict__ y)
{
    constexpr int PER = NNZB / kWG; // nonzeros per thread in phase 1
    constexpr int LDSN = NNZB + NNZB / 32 + 1;
    __shared__ double s_c[LDSN];
    __shared__ double s_x[LDSN];

    const int b = xcd_remap(blockIdx.x, A.nblk);
    if (b >= A.nblk) return;
    const int tid = threadIdx.x;
    const int2 d0 = A.blk[b];
    const int2 d1 = A.blk[b + 1];
    const int r0 = d0.x, p0 = d0.y, r1 = d1.x, p1 = d1.y;
    const int nn = p1 - p0;
    const int myrow = r0 + tid;

    if (nn == 0) { // a block of empty rows
        for (int r = myrow; r < r1; r += kWG) y[A.rowmap ? A.rowmap[r] : r] = 0.0;
        return;
    }
    if (nn <= NNZB) {
        // Phase 1 is written WITHOUT per-element branches: every thread issues
        // exactly PER coef loads, PER indcol loads and PER gathers, with the slot
        // index clamped to the last valid nonzero (lanes past the end re-read and
        // re-write that last element with identical values — harmless).  With
        // branches hipcc serialises the loads (one round trip per element); as
        // straight-line code all 2*PER stream loads are in flight together, then
        // all PER gathers.
        const int last = nn - 1;
        const int rowc = min(myrow, r1 - 1);
        const int pa = A.ptrow[rowc];     // raw row extents, consumed only in phase 2
        const int pe = A.ptrow[rowc + 1];
        // Column ids are kept UNSIGNED: a signed index is sign-extended for the 64-bit
        // gather address, and hipcc schedules that extension right behind each index
        // load — one s_waitcnt per load, i.e. PER serial round trips to HBM.
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
        // keep the three groups (stream loads | gathers | LDS stores) apart: left to
        // itself the scheduler re-fuses them into per-element load-wait-gather-wait-store
        __builtin_amdgcn_sched_barrier(0);
        double xv[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) xv[i] = x[j[i]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int k = min(tid + i * kWG, last);
            s_c[sk(k)] = c[i];
            s_x[sk(k)] = xv[i];
        }
        __syncthreads();
        if (myrow < r1) y[A.rowmap ? A.rowmap[myrow] : myrow] = row_chain<8>(s_c, s_x, pa - p0, pe - p0);
        for (int r = myrow + kWG; r < r1; r += kWG) { // blocks of very short rows hold more than kWG rows
            const int ra = A.ptrow[r] - p0, re = A.ptrow[r + 1] - p0;
            y[A.rowmap ? A.rowmap[r] : r] = row_chain<8>(s_c, s_x, ra, re);
        }
    } else {
        // one row longer than a block: stream it chunk by chunk; the chain itself
        // is inherently serial, thread 0 carries it (exactness over speed: FE rows
        // never get here, their length is bounded by the mesh valence).
        double s = 0.0;
        for (int base = p0; base < p1; base += NNZB) {
            const int m = min(NNZB, p1 - base);
            for (int k = tid; k < m; k += kWG) {
                s_c[sk(k)] = A.coef[base + k];
                s_x[sk(k)] = x[A.indcol[base + k]];
            }
            __syncthreads();
            if (tid == 0)
                for (int k = 0; k < m; k++) s = fma(s_c[sk(k)], s_x[sk(k)], s);
            __syncthreads();
        }
        if (tid == 0) y[A.rowmap ? A.rowmap[r0] : r0] = s;
    }
}

// (the rowpar kernel — one thread per row straight from global memory — lives in spmv_rowpar.hpp)

// ---------------------------------------------------------------------------
// BCSR 4x4, row-major blocks (mpk/SpMV.h:26-33, fill order mpk/utils.cpp:83-94).
// Four lanes per block row (lane q owns row 4*bi+q); per block, lane q reads its
// 4 coefficients as two 16-B loads (the 4 lanes together read the block's 128 B
// contiguously) and the 4 x values of the block column.  fma order = for block,
// for j — identical to SpMV_BCSR_FMA (mpk/SpMV.cpp:150-178).
// ---------------------------------------------------------------------------
struct Bcsr4View {
    int nbrows, nbcols;
    const int* ptrow;
    const int* indcol;
    const double* coef; // 16 per block
    const int* browmap; // nullptr, or block row bi writes y[4 * browmap[bi] + q] (reordered matrices, reorder.hpp)
};

// Software pipeline per lane, P blocks deep: while block ia's four fmas run, the values and x
// entries of blocks ia+1 .. ia+P are in flight and the block columns of ia+P+1 .. ia+2P are being
// fetched (x[4*col] cannot be requested before col has arrived, so columns run one round ahead of
// values and x).  Indices are clamped to the row's last block and loads are unconditional; only
// the fmas test against the row end.  A lane's lifetime is (blocks per row) x (memory latency) / P:
// measured on the FE matrix (one box): P = 1 1 177 GFLOP/s, P = 2 1 256, P = 3 1 263, P = 4 1 083
// (registers cost occupancy); kBcsrDepth = 2 is what mi_bcsr4_spmv* launches.
// Tried and dropped: an XCD-aware block-row order (each XCD's L2 gets a contiguous eighth of the
// rows) cuts the HBM reads from 771 to 673 MB per product (minimum 660) but not the time — it is
// 2-3 % SLOWER at every depth, so the kernel is not traffic-bound; non-temporal loads of the block
// values lose badly (870 GFLOP/s: a lane's two 16-byte loads touch the same 128-byte line twice).
constexpr int kBcsrDepth = 2;
// xcd_chunk > 0: workgroups take their block rows in XCD-chunked order (xcd_remap_chunked); the grid is then padded
template <int P>
__global__ __launch_bounds__(kWG) void spmv_bcsr4(Bcsr4View A, const double* __restrict__ x,
                                                  double* __restrict__ y, int xcd_chunk, int nwg)
{
    const int wg = xcd_chunk > 0 ? xcd_remap_chunked(blockIdx.x, nwg, xcd_chunk) : (int)blockIdx.x;
    if (wg >= nwg) return;
    const int g = wg * kWG + threadIdx.x;
    const int bi = g >> 2, q = g & 3;
    if (bi >= A.nbrows) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int ia0 = A.ptrow[bi], ia1 = A.ptrow[bi + 1];
    double s = 0.0;
    if (ia0 < ia1) {
        const int last = ia1 - 1;
        const double* cq = A.coef + 4 * q;
        double2 a01[P], a23[P], x01[P], x23[P];
        unsigned cn[P]; // columns of blocks ia+P+t
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
            const double2* xb = reinterpret_cast<const double2*>(x + 4 * (size_t)cn[t]);
            x01[t] = xb[0];
            x23[t] = xb[1];
        }
#pragma unroll
        for (int t = 0; t < P; t++) cn[t] = ucol[min(ia0 + P + t, last)];
        for (int ia = ia0; ia < ia1; ia += P) {
#pragma unroll
            for (int t = 0; t < P; t++) {
                const double2 c01 = a01[t], c23 = a23[t], v01 = x01[t], v23 = x23[t];
                // refill stage t with block ia+t+P (its column arrived a round ago), then ask for
                // the column of block ia+t+2P
                const int nb = min(ia + t + P, last);
                const double2* nrow = reinterpret_cast<const double2*>(cq + 16 * (size_t)nb);
                a01[t] = nrow[0];
                a23[t] = nrow[1];
                const double2* nxb = reinterpret_cast<const double2*>(x + 4 * (size_t)cn[t]);
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
    y[4 * (size_t)(A.browmap ? A.browmap[bi] : bi) + q] = s;
}

// ---------------------------------------------------------------------------
// spmv_bcsr4 with the x nodes of a workgroup's 64 block rows gathered ONCE into an LDS tile (the tile kernel's idea,
// spmv_tile.hpp, at node granularity): the host lists each workgroup's distinct block columns (ascending) and gives every block
// the 16-bit position of its column in that list.  The inner loop then waits for coefficients only — no column -> x round trip
// through L1/L2 per block — and x is fetched once per workgroup instead of once per block (spmv_bcsr4 moves 1.14-1.17x its
// format's bytes, the excess being x).  Same lanes, same fma order, same bits.  UMAX nodes of tile (32 bytes each).
// ---------------------------------------------------------------------------
constexpr int kBtileNodes = 1024;
struct Bcsr4Tile {
    const int* wg_ptr;            // [nwg + 1] first list entry of each workgroup
    const unsigned* nodes;        // distinct block columns per workgroup
    const unsigned short* slots;  // per block: position of its column in its workgroup's list
    const int* rows = nullptr;    // multi-vector tiles only (spmm_tile.hpp): block row of every lane group, -1 - r for unused places
};

template <int P>
__global__ __launch_bounds__(kWG) void spmv_bcsr4_tile(Bcsr4View A, Bcsr4Tile Tl, const double* __restrict__ x, double* __restrict__ y, int nwg)
{
    __shared__ __attribute__((aligned(16))) double s_x[4 * kBtileNodes];
    const int wg = (int)blockIdx.x;
    if (wg >= nwg) return;
    const int tid = threadIdx.x;
    const int g = wg * kWG + tid;
    const int bi = min(g >> 2, A.nbrows - 1), q = g & 3; // (lanes past the last block row shadow it and store nothing)
    const bool live = (g >> 2) < A.nbrows;
    const int u0 = Tl.wg_ptr[wg], U = Tl.wg_ptr[wg + 1] - u0;
    // the tile: every thread fetches whole nodes (two 16-byte loads), list first
    for (int i = tid; i < U; i += kWG) {
        const unsigned node = Tl.nodes[u0 + i];
        const double2* xb = reinterpret_cast<const double2*>(x + 4 * (size_t)node);
        const double2 v0 = xb[0], v1 = xb[1];
        reinterpret_cast<double2*>(s_x)[2 * i] = v0;
        reinterpret_cast<double2*>(s_x)[2 * i + 1] = v1;
    }
    const int ia0 = A.ptrow[bi], ia1 = A.ptrow[bi + 1];
    const int last = max(ia1 - 1, ia0);
    const double* cq = A.coef + 4 * q;
    double2 a01[P], a23[P];
    unsigned sl[P];
#pragma unroll
    for (int t = 0; t < P; t++) { // the first coefficient stages are in flight across the barrier
        const int blk = min(ia0 + t, last);
        const double2* row = reinterpret_cast<const double2*>(cq + 16 * (size_t)blk);
        a01[t] = row[0];
        a23[t] = row[1];
        sl[t] = Tl.slots[blk];
    }
    __syncthreads();
    double s = 0.0;
    for (int ia = ia0; ia < ia1; ia += P) {
#pragma unroll
        for (int t = 0; t < P; t++) {
            const double2 c01 = a01[t], c23 = a23[t];
            const double2* xs = reinterpret_cast<const double2*>(s_x + 4 * sl[t]);
            const double2 v01 = xs[0], v23 = xs[1];
            const int nb = min(ia + t + P, last);
            const double2* nrow = reinterpret_cast<const double2*>(cq + 16 * (size_t)nb);
            a01[t] = nrow[0];
            a23[t] = nrow[1];
            sl[t] = Tl.slots[nb];
            if (ia + t < ia1) {
                s = fma(c01.x, v01.x, s);
                s = fma(c01.y, v01.y, s);
                s = fma(c23.x, v23.x, s);
                s = fma(c23.y, v23.y, s);
            }
        }
    }
    if (live) y[4 * (size_t)(A.browmap ? A.browmap[bi] : bi) + q] = s;
}

} // namespace mi355

namespace mi355 {

// ---------------------------------------------------------------------------
// Multi-vector product on the blocked matrix: Y[:, j] = A X[:, j] for j < S, the matrix read ONCE for
// the S vectors — the operation of the reference's s-step kernel MatMatMult_SeqBAIJ_4_AVX2
// (src/kernels/spmm_avx2.c:7-109; X and Y dense column-major with leading dimension lda = 4 * mbs, :23).
// Same lane layout as spmv_bcsr4 (four lanes per block row, lane q owns row 4*bi+q, two 16-byte
// loads of coefficients per block) with S accumulators per lane; the x blocks of the S columns are
// gathered through L2.  Per block and lane 2 + 2 S loads feed 4 S fmas, so from S = 2 on the loop is
// no longer latency-bound the way the single-vector kernel is, and the matrix bytes per flop fall
// with S: bytes = 132 per block + 16 S per row (x read + y written once per column).
// ARITH 0: every row of every column is ONE continuous fma chain over the row's blocks — bit-equal to
//          SpMV_BCSR_FMA (mpk/SpMV.cpp:150-178) column by column, hence to the CSR fma chain.
// ARITH 1: per block the four products are chained from zero and the partial is ADDED to the row's
//          running value — the association spmm_avx2.c:77-88 (and SpM2V_BCSR_OPT, mpk/SpM2V.cpp:502-507)
//          use.  (The reference's horizontal sum at :96-101 adds four identical broadcast lanes and so
//          returns 4 A X; that defect is not reproduced.)
// PF: the next block's S x blocks are requested one block ahead (register-hungry: on for small S).
// ---------------------------------------------------------------------------
// XCD: workgroup b works on chunk (b & 7) * ceil(nwg / 8) + (b >> 3) of the block rows, so that each XCD (round-robin
// dispatch) sweeps one contiguous eighth of the matrix and the x blocks its L2 fetched for one workgroup serve the next;
// in dispatch order all eight L2s walk the same region and each fetches every x block for itself (S columns of them).
// The four lanes of a block row need the same x block; loaded by each of them it costs four times the L1 bandwidth of the
// one copy (a 16-byte-per-lane load occupies the CU's L1 path for 16 cycles whatever the addresses), and from four columns
// on that path, not HBM, bounds the kernel.  quad_bcast hands lane K's value to all four lanes of its quad through DPP
// (`v_mov_b32 … quad_perm:[K,K,K,K]`): two full-rate VALU moves per double instead of a memory instruction.
template <int K>
__device__ __forceinline__ double quad_bcast(double v)
{
    constexpr int ctrl = K * 0x55; // quad_perm:[K,K,K,K]
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// one block's update of accumulator j from coefficients (c01, c23) and x block (v01, v23)
template <int ARITH>
__device__ __forceinline__ double spmm_block_update(double acc, double2 c01, double2 c23, double x0, double x1, double x2, double x3)
{
    if (ARITH == 0) {
        acc = fma(c01.x, x0, acc);
        acc = fma(c01.y, x1, acc);
        acc = fma(c23.x, x2, acc);
        return fma(c23.y, x3, acc);
    }
    double p = fma(c01.x, x0, 0.0);
    p = fma(c01.y, x1, p);
    p = fma(c23.x, x2, p);
    p = fma(c23.y, x3, p);
    return __dadd_rn(acc, p);
}

// S a multiple of 4: lane q of a block row loads the x blocks of columns q, q + 4, ... only (S/4 x blocks instead of S) and
// the quad shares them through DPP.  Same arithmetic, same order per (row, column) as spmm_bcsr4: bit-identical.
// What bounds these kernels is not L1 bandwidth but memory-level parallelism — a lane's chain over its ~15 blocks is
// sequential, so its speed is (loads in flight) / latency: with one block of look-ahead the quad form, which issues FEWER
// loads per block, was SLOWER at four columns (254 vs 176 us) and faster at eight only because the plain form there has
// no look-ahead at all.  The registers the shared x blocks free are therefore spent on a P-deep software pipeline like
// spmv_bcsr4's: stage t holds coefficients and x blocks of block ia + t, refilled with block ia + t + P as soon as it is
// consumed; block columns run another P ahead (an x address needs its column first).
template <int S, int ARITH, bool XCD, int P>
__global__ __launch_bounds__(kWG) void spmm_bcsr4_quad(Bcsr4View A, const double* __restrict__ X, long long ldx,
                                                       double* __restrict__ Y, long long ldy, int nwg)
{
    static_assert(S % 4 == 0, "quad sharing needs a multiple of four columns");
    constexpr int G = S / 4;
    const int wg = XCD ? xcd_remap(blockIdx.x, nwg) : (int)blockIdx.x;
    if (wg >= nwg) return;
    const int g = wg * kWG + threadIdx.x;
    // no early exit per lane: DPP reads the quad's other lanes, which must be live; the last block row is clamped instead
    const int bi_raw = g >> 2, q = g & 3;
    const int bi = min(bi_raw, A.nbrows - 1);
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int ia0 = A.ptrow[bi], ia1 = A.ptrow[bi + 1];
    double acc[S];
#pragma unroll
    for (int j = 0; j < S; j++) acc[j] = 0.0;
    if (ia0 < ia1) { // uniform within a quad (the four lanes share bi)
        const int last = ia1 - 1;
        const double* cq = A.coef + 4 * q;
        const double* Xq = X + (size_t)q * ldx; // this lane's first column; its others are 4 * ldx apart
        double2 a01[P], a23[P], x01[P][G], x23[P][G];
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
        for (int t = 0; t < P; t++)
#pragma unroll
            for (int u = 0; u < G; u++) {
                const double2* xb = reinterpret_cast<const double2*>(Xq + (size_t)(4 * u) * ldx + 4 * (size_t)cn[t]);
                x01[t][u] = xb[0];
                x23[t][u] = xb[1];
            }
#pragma unroll
        for (int t = 0; t < P; t++) cn[t] = ucol[min(ia0 + P + t, last)];
        for (int ia = ia0; ia < ia1; ia += P) {
#pragma unroll
            for (int t = 0; t < P; t++) {
                const double2 c01 = a01[t], c23 = a23[t];
                double2 v01[G], v23[G];
#pragma unroll
                for (int u = 0; u < G; u++) { v01[u] = x01[t][u]; v23[u] = x23[t][u]; }
                // refill stage t with block ia + t + P (its column arrived a round ago), then ask for the column of ia + t + 2P
                const int nb = min(ia + t + P, last);
                const double2* nrow = reinterpret_cast<const double2*>(cq + 16 * (size_t)nb);
                a01[t] = nrow[0];
                a23[t] = nrow[1];
#pragma unroll
                for (int u = 0; u < G; u++) {
                    const double2* xb = reinterpret_cast<const double2*>(Xq + (size_t)(4 * u) * ldx + 4 * (size_t)cn[t]);
                    x01[t][u] = xb[0];
                    x23[t][u] = xb[1];
                }
                cn[t] = ucol[min(ia + t + 2 * P, last)];
                if (ia + t < ia1) { // uniform within the quad
#pragma unroll
                    for (int u = 0; u < G; u++) { // columns 4u + K come from lane K of the quad
                        acc[4 * u + 0] = spmm_block_update<ARITH>(acc[4 * u + 0], c01, c23, quad_bcast<0>(v01[u].x), quad_bcast<0>(v01[u].y), quad_bcast<0>(v23[u].x), quad_bcast<0>(v23[u].y));
                        acc[4 * u + 1] = spmm_block_update<ARITH>(acc[4 * u + 1], c01, c23, quad_bcast<1>(v01[u].x), quad_bcast<1>(v01[u].y), quad_bcast<1>(v23[u].x), quad_bcast<1>(v23[u].y));
                        acc[4 * u + 2] = spmm_block_update<ARITH>(acc[4 * u + 2], c01, c23, quad_bcast<2>(v01[u].x), quad_bcast<2>(v01[u].y), quad_bcast<2>(v23[u].x), quad_bcast<2>(v23[u].y));
                        acc[4 * u + 3] = spmm_block_update<ARITH>(acc[4 * u + 3], c01, c23, quad_bcast<3>(v01[u].x), quad_bcast<3>(v01[u].y), quad_bcast<3>(v23[u].x), quad_bcast<3>(v23[u].y));
                    }
                }
            }
        }
    }
    if (bi_raw < A.nbrows) {
        const size_t orow = 4 * (size_t)(A.browmap ? A.browmap[bi] : bi) + q;
#pragma unroll
        for (int j = 0; j < S; j++) Y[(size_t)j * ldy + orow] = acc[j];
    }
}

template <int S, int ARITH, bool PF, bool XCD>
__global__ __launch_bounds__(kWG) void spmm_bcsr4(Bcsr4View A, const double* __restrict__ X, long long ldx,
                                                  double* __restrict__ Y, long long ldy, int nwg)
{
    const int wg = XCD ? xcd_remap(blockIdx.x, nwg) : (int)blockIdx.x;
    if (wg >= nwg) return;
    const int g = wg * kWG + threadIdx.x;
    const int bi = g >> 2, q = g & 3;
    if (bi >= A.nbrows) return;
    const unsigned* ucol = reinterpret_cast<const unsigned*>(A.indcol);
    const int ia0 = A.ptrow[bi], ia1 = A.ptrow[bi + 1];
    double acc[S];
#pragma unroll
    for (int j = 0; j < S; j++) acc[j] = 0.0;
    if (ia0 < ia1) {
        const int last = ia1 - 1;
        const double* cq = A.coef + 4 * q;
        // stage: coefficients + column of the block about to be consumed, its x blocks (PF) and the column after it
        const double2* row = reinterpret_cast<const double2*>(cq + 16 * (size_t)ia0);
        double2 a01 = row[0], a23 = row[1];
        unsigned col = ucol[ia0];
        unsigned coln = ucol[min(ia0 + 1, last)];
        double2 x01[S], x23[S];
        if (PF) {
#pragma unroll
            for (int j = 0; j < S; j++) {
                const double2* xb = reinterpret_cast<const double2*>(X + (size_t)j * ldx + 4 * (size_t)col);
                x01[j] = xb[0];
                x23[j] = xb[1];
            }
        }
        for (int ia = ia0; ia < ia1; ia++) {
            const double2 c01 = a01, c23 = a23;
            double2 v01[S], v23[S];
            if (PF) {
#pragma unroll
                for (int j = 0; j < S; j++) { v01[j] = x01[j]; v23[j] = x23[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < S; j++) {
                    const double2* xb = reinterpret_cast<const double2*>(X + (size_t)j * ldx + 4 * (size_t)col);
                    v01[j] = xb[0];
                    v23[j] = xb[1];
                }
            }
            // refill: block ia+1 (clamped, unconditional), the column of ia+2
            const int nb = min(ia + 1, last);
            const double2* nrow = reinterpret_cast<const double2*>(cq + 16 * (size_t)nb);
            a01 = nrow[0];
            a23 = nrow[1];
            col = coln;
            coln = ucol[min(ia + 2, last)];
            if (PF) {
#pragma unroll
                for (int j = 0; j < S; j++) {
                    const double2* xb = reinterpret_cast<const double2*>(X + (size_t)j * ldx + 4 * (size_t)col);
                    x01[j] = xb[0];
                    x23[j] = xb[1];
                }
            }
#pragma unroll
            for (int j = 0; j < S; j++) {
                if (ARITH == 0) {
                    double s = acc[j];
                    s = fma(c01.x, v01[j].x, s);
                    s = fma(c01.y, v01[j].y, s);
                    s = fma(c23.x, v23[j].x, s);
                    s = fma(c23.y, v23[j].y, s);
                    acc[j] = s;
                } else {
                    double p = fma(c01.x, v01[j].x, 0.0);
                    p = fma(c01.y, v01[j].y, p);
                    p = fma(c23.x, v23[j].x, p);
                    p = fma(c23.y, v23[j].y, p);
                    acc[j] = __dadd_rn(acc[j], p);
                }
            }
        }
    }
    const size_t orow = 4 * (size_t)(A.browmap ? A.browmap[bi] : bi) + q;
#pragma unroll
    for (int j = 0; j < S; j++) Y[(size_t)j * ldy + orow] = acc[j];
}

} // namespace mi355
