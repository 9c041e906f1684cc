// spmv_sstream.hpp — y = A x for banded scalar CSR matrices as ONE contiguous stream per wave (round 4; MI_KERNEL_SSTREAM).
//
// SpMV_CSR{,_OPT,_FMA,_AVX2}(y, x, A), mpk/SpMV.cpp:6-85.  Arithmetic unchanged: every row is ONE sequential fma chain over its
// nonzeros in CSR order — the bits of SpMV_CSR_FMA (mpk/SpMV.cpp:41-56).
//
// The ring kernel (spmv_ring.hpp) moves a 2048-nonzero block through four phases — coalesced loads, LDS gather, {coef, x} staging,
// one thread per row walking its LDS segment — and its per-block pipeline (2.04 us per block and workgroup, the same whether the
// matrix comes from HBM or the Infinity Cache) is what bounds it at 1 M rows and keeps it 5-8 % over its memory skeleton at 5 M.
// spmv_bcsr_sell.hpp showed what a stream without phases does for the blocked format; this is the same idea for scalar rows:
//   * a LANE owns a ROW PAIR and walks both rows nonzero by nonzero — no staging, no row-chain phase, every lane busy;
//   * the library keeps a SLICED copy of the matrix: a slice is 128 consecutive rows (one wave), padded to its longest row; step j of
//     a slice holds the j-th nonzero of each of its rows as 16 bytes per lane — {a(row 2l, j), a(row 2l + 1, j)}: ONE contiguous KiB per
//     wave-instruction, loaded non-temporally — plus one 32-bit word per lane with the two 13-bit LDS slots of their columns and
//     three flags (padding place per row; first step of a slice): 10 bytes per nonzero, like the ring's 16-bit column stream;
//   * x lives in an LDS RING indexed by column (8192 entries, slot = column mod 8192) that slides with the rows: a workgroup is four
//     waves (ONE per SIMD: of two waves on a SIMD the one dispatched first starves the other, spmv_bcsr_sell.hpp) working on the four
//     neighbouring slices of one 512-row ROUND; between rounds — two barriers — the window takes in its new columns, which were
//     loaded a round ahead into registers (at most 1024 per round: a matrix that needs more, or whose rows reach further apart than
//     the ring holds, is not eligible and keeps the ring / multi-window / tile / stream kernels);
//   * a wave's stream is contiguous (its slices of consecutive rounds lie one after the other), prefetched D steps ahead with
//     unconditional counted loads; a finished slice's 128 sums are PARKED in LDS (20 per wave) and stored, a KiB per slice, when the
//     park is full or the workgroup's range ends — stores issued among streaming loads cost the read stream many times their bytes.
// Prototype numbers (tools/sstream_bench.hip, profiles/r04_sstream_bench.txt; S15, every bit checked against the host's fma chain):
// 5 M rows 131-135 us = 0.93-0.95 of 8 TB/s on the CSR byte model (the ring kernel 137-150 us on the same pool), with the y stores
// compiled out 116-118 us; 1 M rows 30.7 us with temporal loads (the matrix lives in the Infinity Cache there) against 31-33 us.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace mi355 {

typedef double ss_v2d __attribute__((ext_vector_type(2)));
constexpr int kSsRing = 8192;            // x entries in the LDS ring (64 KB)
constexpr int kSsSliceRows = 128;        // rows per slice = two per lane
constexpr int kSsRound = 512;            // rows per round of a workgroup: four waves' slices
constexpr int kSsNewMax = 1024;          // new window columns per round (four per thread, a round ahead in registers)
constexpr int kSsPark = 20;              // slices of y parked per wave: 4 x 20 KiB beside the 64 KiB ring
constexpr int kSsTail = 2;               // rounds of a workgroup whose sums stay parked until its end (the others are stored earlier)
constexpr unsigned kSsPad = 0x8000u;     // slot flag (either half): padding place, not multiplied
constexpr unsigned kSsFirst = 0x4000u;   // slot flag (low half): first step of a slice
constexpr int kSsPadSteps = 64;          // steps of padding behind the last one: the stream's loads run ahead unclamped (D <= 16)
constexpr int kSsMaxWgs = 256;           // one workgroup per CU

struct SsView {
    const ss_v2d* val;     // [steps + kSsPadSteps][64]
    const unsigned* slot;  // [steps + kSsPadSteps][64]: low half = row 2l, high half = row 2l + 1
    const int* wptr;       // [nwg * 4 + 1] first step of each wave's stream
    const int* rptr;       // [nwg + 1] first round of each workgroup
    const int2* win;       // [rounds] {first new column, count} the window takes in before the round (a workgroup's first round: its first fill)
    int nwg, n, ncols;
    const int* rowmap;     // nullptr, or row r writes y[rowmap[r]] (partition pieces, the relabelled twin of reorder.hpp)
};

// ---- host: the plan --------------------------------------------------------------------------------------------------------------
struct SsPlanHost {
    bool eligible = false;
    const char* why = "";            // first reason the matrix is not eligible
    int nwg = 0, rounds = 0;
    long long steps = 0, pad_places = 0;
    std::vector<int> wptr, rptr, slice_step, slice_len; // slice_step / slice_len[4 * round + wave] = first step / steps of that slice (value refills)
    std::vector<int2> win;
    std::vector<unsigned> slot;      // [steps + kSsPadSteps][64]
};

// build the plan of an n x ncols pattern (columns ascending or not: a row's nonzeros keep their CSR order); max_padding = padded places per nonzero allowed
inline void build_sstream_plan(int n, int ncols, const int* ptrow, const int* indcol, double max_padding, SsPlanHost& P, bool want_slots = true)
{
    P = SsPlanHost();
    const long long nnz = n > 0 ? ptrow[n] : 0;
    if (n <= 0 || nnz <= 0) { P.why = "empty matrix"; return; }
    const int rounds = (n + kSsRound - 1) / kSsRound;
    int nwg = std::min(kSsMaxWgs, rounds);
    if (nwg >= 8) nwg = nwg / 8 * 8; // a multiple of the XCD count keeps the workgroup -> XCD dealing of the kernel
    P.nwg = nwg;
    P.rounds = rounds;
    P.rptr.resize((size_t)nwg + 1);
    for (int g = 0; g <= nwg; g++) P.rptr[g] = (int)((long long)rounds * g / nwg);
    // column extent of every round
    std::vector<int> cmin((size_t)rounds, 0x7fffffff), cmax((size_t)rounds, -1);
    for (int r = 0; r < rounds; r++) {
        int lo = 0x7fffffff, hi = -1;
        const int i1 = std::min(n, (r + 1) * kSsRound);
        for (int k = ptrow[r * kSsRound]; k < ptrow[i1]; k++) {
            lo = std::min(lo, indcol[k]);
            hi = std::max(hi, indcol[k]);
        }
        cmin[r] = lo;
        cmax[r] = hi;
    }
    // windows: per workgroup a monotone upper end; everything a round names must lie within kSsRing below it
    P.win.assign((size_t)rounds, make_int2(0, 0));
    for (int g = 0; g < nwg; g++) {
        int allmin = 0x7fffffff;
        for (int r = P.rptr[g]; r < P.rptr[g + 1]; r++) allmin = std::min(allmin, cmin[r]);
        int whi = 0;
        for (int r = P.rptr[g]; r < P.rptr[g + 1]; r++) {
            const int nhi = std::max(whi, cmax[r] + 1);
            if (r == P.rptr[g]) {
                const int lo = std::max(std::max(0, nhi - kSsRing), std::min(allmin, nhi));
                P.win[r] = make_int2(lo, nhi - lo);
            } else {
                P.win[r] = make_int2(whi, nhi - whi);
                if (nhi - whi > kSsNewMax) { P.why = "a round brings more new columns than the window takes in at once"; return; }
            }
            whi = nhi;
            if (cmin[r] != 0x7fffffff && cmin[r] < whi - kSsRing) { P.why = "a round's rows reach further apart than the LDS ring holds"; return; }
        }
    }
    // streams: workgroup by workgroup, wave by wave, round by round
    P.wptr.assign((size_t)nwg * 4 + 1, 0);
    P.slice_step.assign((size_t)rounds * 4, 0);
    P.slice_len.assign((size_t)rounds * 4, 0);
    long long t = 0, places = 0;
    for (int g = 0; g < nwg; g++)
        for (int wv = 0; wv < 4; wv++) {
            P.wptr[(size_t)g * 4 + wv] = (int)t;
            for (int r = P.rptr[g]; r < P.rptr[g + 1]; r++) {
                const int row0 = r * kSsRound + wv * kSsSliceRows;
                int L = 1;
                for (int i = row0; i < std::min(n, row0 + kSsSliceRows); i++) L = std::max(L, ptrow[i + 1] - ptrow[i]);
                P.slice_step[(size_t)r * 4 + wv] = (int)t;
                P.slice_len[(size_t)r * 4 + wv] = L;
                t += L;
                places += (long long)L * kSsSliceRows;
            }
        }
    P.wptr[(size_t)nwg * 4] = (int)t;
    P.steps = t;
    P.pad_places = places - nnz;
    if (t + kSsPadSteps >= 0x7fffffffLL / 64) { P.why = "too many steps for 32-bit offsets"; return; }
    if ((double)P.pad_places > max_padding * (double)nnz) { P.why = "row lengths vary too much inside the 128-row slices (padding)"; return; }
    P.eligible = true;
    if (!want_slots) return;
    P.slot.assign((size_t)(t + kSsPadSteps) * 64, kSsPad | (kSsPad << 16) | kSsFirst);
    auto fill = [&](int r0, int r1) {
        for (int r = r0; r < r1; r++)
            for (int wv = 0; wv < 4; wv++) {
                const int row0 = r * kSsRound + wv * kSsSliceRows;
                const size_t base = (size_t)P.slice_step[(size_t)r * 4 + wv] * 64;
                const int L = P.slice_len[(size_t)r * 4 + wv];
                for (int j = 0; j < L; j++)
                    for (int l = 0; l < 64; l++) {
                        unsigned s = 0;
                        for (int h = 0; h < 2; h++) {
                            const int i = row0 + 2 * l + h;
                            unsigned sh = kSsPad;
                            if (i < n && j < ptrow[i + 1] - ptrow[i]) sh = (unsigned)(indcol[ptrow[i] + j] & (kSsRing - 1));
                            s |= sh << (16 * h);
                        }
                        if (j == 0) s |= kSsFirst;
                        P.slot[base + (size_t)j * 64 + l] = s;
                    }
            }
    };
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt > 16 ? 16 : (nt < 1 ? 1 : nt);
    if (rounds < 64 || nt == 1) fill(0, rounds);
    else {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; k++) th.emplace_back(fill, (int)((long long)rounds * k / nt), (int)((long long)rounds * (k + 1) / nt));
        for (std::thread& x : th) x.join();
    }
}

// replay of the plan against the matrix (host-only tests): every nonzero's slot is its column mod kSsRing and the column lies inside
// the window when its round runs; the padding places are flagged; returns nullptr or the first violation
inline const char* check_sstream_plan(const SsPlanHost& P, int n, const int* ptrow, const int* indcol)
{
    if (!P.eligible) return nullptr;
    for (int g = 0; g < P.nwg; g++) {
        int wlo = 0, whi = 0;
        for (int r = P.rptr[g]; r < P.rptr[g + 1]; r++) {
            const int2 w = P.win[r];
            if (r == P.rptr[g]) { wlo = w.x; whi = w.x + w.y; }
            else {
                if (w.x != whi) return "a round's new columns do not continue the window";
                whi += w.y;
                if (w.y > kSsNewMax) return "too many new columns";
            }
            wlo = std::max(wlo, whi - kSsRing);
            for (int wv = 0; wv < 4; wv++) {
                const int row0 = r * kSsRound + wv * kSsSliceRows;
                const size_t base = (size_t)P.slice_step[(size_t)r * 4 + wv] * 64;
                int L = 1;
                for (int i = row0; i < std::min(n, row0 + kSsSliceRows); i++) L = std::max(L, ptrow[i + 1] - ptrow[i]);
                for (int j = 0; j < L; j++)
                    for (int l = 0; l < 64; l++) {
                        const unsigned s = P.slot[base + (size_t)j * 64 + l];
                        if (l == 0 && ((s & kSsFirst) != 0) != (j == 0)) return "slice-begin flag misplaced";
                        for (int h = 0; h < 2; h++) {
                            const int i = row0 + 2 * l + h;
                            const unsigned sh = (s >> (16 * h)) & 0xffffu;
                            const bool real = i < n && j < ptrow[i + 1] - ptrow[i];
                            if (!real) {
                                if (!(sh & kSsPad)) return "a padding place is not flagged";
                                continue;
                            }
                            const int c = indcol[ptrow[i] + j];
                            if (sh & kSsPad) return "a nonzero is flagged as padding";
                            if ((sh & (kSsRing - 1)) != (unsigned)(c & (kSsRing - 1))) return "slot is not the column's ring slot";
                            if (c < wlo || c >= whi) return "a column lies outside the window when its round runs";
                        }
                    }
            }
        }
    }
    return nullptr;
}

// ---- device ------------------------------------------------------------------------------------------------------------------------
// (re)fills the sliced values from the CSR values: one wave per slice (setup and value refreshes; never per product)
__global__ __launch_bounds__(64) void csr_to_sstream_kernel(int nslices, int n, const int* __restrict__ ptrow, const double* __restrict__ coef,
                                                            const int* __restrict__ slice_step, const int* __restrict__ slice_len, ss_v2d* __restrict__ val)
{
    const int lane = threadIdx.x;
    for (int sidx = blockIdx.x; sidx < nslices; sidx += gridDim.x) { // sidx = 4 * round + wave: rows [128 * sidx, 128 * sidx + 128)
        const int i0 = kSsSliceRows * sidx + 2 * lane, i1 = i0 + 1;
        const int p0 = i0 < n ? ptrow[i0] : 0, n0 = i0 < n ? ptrow[i0 + 1] - p0 : 0;
        const int p1 = i1 < n ? ptrow[i1] : 0, n1 = i1 < n ? ptrow[i1 + 1] - p1 : 0;
        const int t0 = slice_step[sidx], L = slice_len[sidx];
        for (int j = 0; j < L; j++) {
            ss_v2d v = {0.0, 0.0};
            if (j < n0) v.x = coef[p0 + j];
            if (j < n1) v.y = coef[p1 + j];
            val[(size_t)(t0 + j) * 64 + lane] = v;
        }
    }
}

// D steps of the stream in flight per lane; ONE workgroup of four waves per CU.  Workgroup b is taken as logical workgroup
// (b % 8) * (G / 8) + b / 8 so that the workgroups that share an XCD stream neighbouring rows (their x lines meet in that XCD's L2).
// ABL (tools/sstream_ablate.hip only; invalid results): 1 no LDS gather, 2 no y stores, 4 no new-column loads / window refills at the round boundaries.
template <int D, bool NT, int ABL = 0>
__global__ __launch_bounds__(256) void spmv_sstream(SsView S, const double* __restrict__ x, double* __restrict__ y)
{
    __shared__ double ring[kSsRing];
    __shared__ ss_v2d s_park[4 * kSsPark * 64];
    const int per = S.nwg >> 3;
    const int g = per > 0 && (S.nwg & 7) == 0 ? ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    ss_v2d* park = s_park + wv * kSsPark * 64 + lane;
    int parked = 0, park_first = 0; // (wave-uniform) the slices of rounds park_first .. park_first + parked - 1 are parked
    const int r_begin = S.rptr[g], r_end = S.rptr[g + 1];
    if (r_begin >= r_end) return;
    const int t0 = __builtin_amdgcn_readfirstlane(S.wptr[g * 4 + wv]);
    const int t_end = __builtin_amdgcn_readfirstlane(S.wptr[g * 4 + wv + 1]);
    const int clast = S.ncols - 1;
    // the stream's first D steps go out FIRST: they travel while the window fills (the fill's wait covers them: one memory latency
    // in front of the loop instead of two)
    const ss_v2d* vb = S.val + lane;
    const unsigned* sb = S.slot + lane;
    ss_v2d a[D];
    unsigned sl[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        a[d] = NT ? __builtin_nontemporal_load(vb + (size_t)(t0 + d) * 64) : vb[(size_t)(t0 + d) * 64];
        sl[d] = sb[(size_t)(t0 + d) * 64];
    }
    int r = r_begin; // the round this wave's current slice belongs to
    double nx[kSsNewMax / 256]; // the NEXT round's new columns, a round ahead in registers
    int2 wn = S.win[min(r + 1, r_end - 1)];
#pragma unroll
    for (int u = 0; u < kSsNewMax / 256; u++) nx[u] = (ABL & 4) ? 0.0 : x[min(wn.x + tid + 256 * u, clast)];
    { // first fill of the window
        const int2 w = S.win[r_begin];
        for (int c = w.x + tid; c < w.x + w.y; c += 256) ring[c & (kSsRing - 1)] = x[c];
    }
    __syncthreads();
    double acc0 = 0.0, acc1 = 0.0;
    auto store = [&](int round, ss_v2d v) {
        const int row0 = round * kSsRound + wv * kSsSliceRows + 2 * lane;
        if ((ABL & 2) && v.x != 123.456) return;
        if (S.rowmap) { // (wave-uniform) mapped rows: two 8-byte stores wherever the map sends them
            if (row0 < S.n) y[S.rowmap[row0]] = v.x;
            if (row0 + 1 < S.n) y[S.rowmap[row0 + 1]] = v.y;
        } else if (row0 + 1 < S.n) *reinterpret_cast<ss_v2d*>(y + row0) = v;
        else if (row0 < S.n) y[row0] = v.x;
    };
    auto flush = [&]() {
        for (int j = 0; j < parked; j++) store(park_first + j, park[j * 64]);
        parked = 0;
    };
    auto emit = [&]() { // this wave's slice of round r is complete
        if (parked == 0) park_first = r;
        park[parked * 64] = ss_v2d{acc0, acc1};
        // (stored when the park is full — and kSsTail rounds before the wave's last, so that the stores behind the last load are few:
        // at 1 M rows a workgroup has 7-8 rounds and would otherwise store its whole share of y after everything else)
        if (++parked == kSsPark || r == r_end - 1 - kSsTail) flush();
    };
    for (int t = t0; t < t_end; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int i = t + d;
            if (i < t_end) { // (wave-uniform)
                const unsigned s = sl[d];
                if ((__builtin_amdgcn_readfirstlane(s) & kSsFirst) && i != t0) { // the slice is complete; every wave of the workgroup comes by here once per round
                    emit();
                    acc0 = acc1 = 0.0;
                    r++;
                    __syncthreads(); // every wave is through with round r - 1: the ring entries about to be overwritten are dead
                    if (!(ABL & 4)) {
#pragma unroll
                        for (int u = 0; u < kSsNewMax / 256; u++) {
                            const int c = wn.x + tid + 256 * u;
                            if (c < wn.x + wn.y) ring[c & (kSsRing - 1)] = nx[u];
                        }
                    }
                    __syncthreads();
                    if (!(ABL & 4)) {
                        wn = S.win[min(r + 1, r_end - 1)];
#pragma unroll
                        for (int u = 0; u < kSsNewMax / 256; u++) nx[u] = x[min(wn.x + tid + 256 * u, clast)];
                    }
                }
                const double x0 = (ABL & 1) ? 1.0 + lane : ring[s & (kSsRing - 1)], x1 = (ABL & 1) ? 0.5 : ring[(s >> 16) & (kSsRing - 1)];
                const double n0 = fma(a[d].x, x0, acc0), n1 = fma(a[d].y, x1, acc1);
                acc0 = (s & kSsPad) ? acc0 : n0; // padding places are not multiplied
                acc1 = (s & (kSsPad << 16)) ? acc1 : n1;
            }
            a[d] = NT ? __builtin_nontemporal_load(vb + (size_t)(i + D) * 64) : vb[(size_t)(i + D) * 64];
            sl[d] = sb[(size_t)(i + D) * 64];
        }
    }
    emit();
    flush();
}

} // namespace mi355
