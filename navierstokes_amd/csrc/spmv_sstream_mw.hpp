// spmv_sstream_mw.hpp — the sliced stream (spmv_sstream.hpp) for matrices whose rows name SEVERAL column neighbourhoods: 3-D mesh
// operators in natural node order (a row of plane z names nodes of planes z - 1, z, z + 1: three windows 29 k columns apart on the
// 170^3-cell P1 pressure operator of src/integration.c), which the single 8192-entry window cannot hold (round 5).
//
// SpMV_CSR{,_OPT,_FMA,_AVX2}(y, x, A), mpk/SpMV.cpp:6-85; same arithmetic as spmv_sstream: one sequential fma chain per row in CSR order.
//
// Everything of the sliced stream stays — the sliced copy (10 bytes per nonzero), a lane per row pair (here rows l and l + 64 of the slice:
// the lanes of a wave then read NEIGHBOURING entries of the ring at a step of a mesh operator, where rows 2 l, 2 l + 1 read every other one
// and pay two-way bank conflicts: 19 us of the 5 M-row launch), four waves on the four slices of a 512-row round, the stream D steps ahead,
// y parked in LDS — except the window: the LDS ring is cut into FOUR sub-rings of 2048 entries,
// position = 2048 k + (column mod 2048).  The planner finds, per round, the (at most four) column intervals its rows name (gaps wider than
// kSsMwGap split them), follows each interval from round to round in ONE sub-ring (an interval of round r + 1 inherits the sub-ring of the
// interval of round r it touches), and records per round and sub-ring the NEW columns to take in (at most 512: what exceeds that comes in
// over earlier rounds): the kernel loads them a round ahead into registers, as spmv_sstream does for its one window.  The 16-bit slots of the stream are positions in the cut ring, so the stream loop is
// spmv_sstream's, instruction for instruction.  A matrix is eligible if no round names more than four intervals, none wider than a
// sub-ring, and no round takes in more than kSsMwNewMax columns; whatever is not keeps the multi-window ring kernel (spmv_mring.hpp).
#pragma once
#include <array>

#include "spmv_sstream.hpp"

namespace mi355 {

constexpr int kSsMwRings = 4;
constexpr int kSsMwCap = kSsRing / kSsMwRings; // 2048 columns per sub-ring
constexpr int kSsMwGap = 512;                  // columns further apart than this start a new interval
constexpr int kSsMwWinNew = 512;               // new columns per round and sub-ring: 256 column PAIRS, one 16-byte load per thread and sub-ring, a round ahead in registers
constexpr int kSsMwNewMax = kSsMwRings * kSsMwWinNew;
constexpr int kSsMwTabMax = 480;               // rounds per workgroup at most: its intake table (32 bytes per round) is staged in LDS beside the ring and the park
constexpr int kSsMwFill = 16;                  // first-fill columns per thread loaded in one batch (4096; more take further batches of 8)

typedef double ss_v2d_u __attribute__((ext_vector_type(2), aligned(8))); // two neighbouring doubles wherever they lie

struct SsMwPlanHost : SsPlanHost {
    std::vector<int2> winK;  // [4 * rounds] {first new column, count} per sub-ring, taken in before the round (a workgroup's first round: its first fill)
};

struct SsMwView {
    SsView S;
    const int2* winK;
};

// the rounds of the workgroups: spmv_sstream's dealing without ghost adjustments (inside each XCD's chunk the first workgroups take the extras)
inline void ss_deal_plain(int nwg, int rounds, std::vector<int>& rptr)
{
    std::vector<int> cnt((size_t)nwg, rounds / nwg);
    const int extra = rounds % nwg;
    const int per = nwg % 8 == 0 ? nwg / 8 : nwg, chunks = nwg / per;
    for (int xcd = 0; xcd < chunks; xcd++) {
        const int e = (int)((long long)extra * (xcd + 1) / chunks - (long long)extra * xcd / chunks);
        for (int j = 0; j < e; j++) cnt[(size_t)xcd * per + j]++;
    }
    rptr.assign((size_t)nwg + 1, 0);
    for (int g = 0; g < nwg; g++) rptr[g + 1] = rptr[g] + cnt[g];
}

inline void build_sstream_mw_plan(int n, int ncols, const int* ptrow, const int* indcol, double max_padding, SsMwPlanHost& P, int shift = 0)
{
    P = SsMwPlanHost();
    const long long nnz = n > 0 ? ptrow[n] : 0;
    if (n <= 0 || nnz <= 0) { P.why = "empty matrix"; return; }
    if (ncols < 2) { P.why = "fewer than two columns"; return; } // (the intake loads column pairs)
    if (shift != 0 && shift != 1) { P.why = "bad shift"; return; }
    P.shift = shift;
    const int nv = n + shift;
    auto PT = [&](int v) { const int i = v - shift; return ptrow[i < 0 ? 0 : (i > n ? n : i)]; };
    const int rounds = (nv + kSsRound - 1) / kSsRound;
    int nwg = std::min(kSsMaxWgs, rounds);
    if (nwg >= 8) nwg = nwg / 8 * 8;
    P.nwg = nwg;
    P.rounds = rounds;
    ss_deal_plain(nwg, rounds, P.rptr);
    // the column intervals of every round
    struct Iv { int lo, hi; }; // [lo, hi)
    std::vector<std::array<Iv, kSsMwRings>> riv((size_t)rounds);
    std::vector<int> rnk((size_t)rounds, 0);
    {
        // (rounds are independent: over the host's threads, like the slot fill below — the 5 M-row mesh sorts 74 M column indices here)
        unsigned nth = std::thread::hardware_concurrency();
        nth = nth > 16 ? 16 : (nth < 1 ? 1 : nth);
        if (rounds < 64) nth = 1;
        std::vector<const char*> bad((size_t)nth, nullptr);
        auto scan = [&](unsigned th, int ra, int rb) {
            std::vector<int> cols;
            for (int r = ra; r < rb && !bad[th]; r++) {
                cols.assign(indcol + PT(r * kSsRound), indcol + PT(std::min(nv, (r + 1) * kSsRound)));
                std::sort(cols.begin(), cols.end());
                cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
                int nk = 0;
                for (size_t i = 0; i < cols.size() && !bad[th]; i++) {
                    if (i == 0 || cols[i] - cols[i - 1] > kSsMwGap) {
                        if (nk == kSsMwRings) { bad[th] = "a round's rows name more than four column neighbourhoods"; break; }
                        riv[r][nk++] = Iv{cols[i], cols[i] + 1};
                    } else riv[r][nk - 1].hi = cols[i] + 1;
                }
                for (int j = 0; j < nk && !bad[th]; j++)
                    if (riv[r][j].hi - riv[r][j].lo > kSsMwCap) bad[th] = "a column neighbourhood is wider than a sub-ring";
                rnk[r] = nk;
            }
        };
        if (nth == 1) scan(0, 0, rounds);
        else {
            std::vector<std::thread> th;
            for (unsigned k = 0; k < nth; k++) th.emplace_back(scan, k, (int)((long long)rounds * k / nth), (int)((long long)rounds * (k + 1) / nth));
            for (std::thread& x : th) x.join();
        }
        for (unsigned k = 0; k < nth; k++)
            if (bad[k]) { P.why = bad[k]; return; }
    }
    // sub-rings: an interval follows the sub-ring of the interval it continues
    P.winK.assign((size_t)rounds * kSsMwRings, make_int2(0, 0));
    std::vector<std::array<signed char, kSsMwRings>> kof((size_t)rounds); // sub-ring of interval j of round r
    struct Rk { int lo, hi, need; bool used, cont; };                     // what sub-ring k holds for round r; cont: it continues round r - 1's
    std::vector<std::array<Rk, kSsMwRings>> rk((size_t)rounds);
    for (int g = 0; g < nwg; g++) {
        const int r0 = P.rptr[g], r1 = P.rptr[g + 1];
        if (r1 - r0 > kSsMwTabMax) { P.why = "too many rounds per workgroup for the staged intake table"; return; }
        // (a) forward: which sub-ring every interval lives in
        for (int r = r0; r < r1; r++) {
            for (int k = 0; k < kSsMwRings; k++) rk[r][k] = Rk{0, 0, 0, false, false};
            for (int j = 0; j < rnk[r]; j++) kof[r][j] = -1;
            if (r > r0)
                for (int j = 0; j < rnk[r]; j++) { // continue a sub-ring whose content this interval touches
                    const Iv iv = riv[r][j];
                    for (int k = 0; k < kSsMwRings; k++)
                        if (rk[r - 1][k].used && !rk[r][k].used && iv.lo <= rk[r - 1][k].hi && iv.hi >= rk[r - 1][k].lo) {
                            // The window wave writes a round's intake WHILE the round before it runs (see the kernel): a neighbourhood
                            // that moves backwards would need columns written where that round still reads
                            if (iv.lo < rk[r - 1][k].lo) { P.why = "a column neighbourhood moves backwards"; return; }
                            kof[r][j] = (signed char)k;
                            rk[r][k] = Rk{iv.lo, iv.hi, iv.hi, true, true};
                            break;
                        }
                }
            for (int j = 0; j < rnk[r]; j++) {
                if (kof[r][j] >= 0) continue;
                int k = -1; // a new neighbourhood: a sub-ring that holds nothing this round — nor, since its columns are written early, the round before
                for (int q = 0; q < kSsMwRings && k < 0; q++)
                    if (!rk[r][q].used && !(r > r0 && rk[r - 1][q].used)) k = q;
                if (k < 0) { P.why = "no free sub-ring for a new column neighbourhood"; return; }
                kof[r][j] = (signed char)k;
                rk[r][k] = Rk{riv[r][j].lo, riv[r][j].hi, riv[r][j].hi, true, false};
                // a neighbourhood wider than one round's intake appears: its sub-ring starts taking it in some rounds earlier (holding
                // nothing anybody reads yet), down to the workgroup's first fill if need be — which takes whatever it is given
                int rr = r;
                for (int left = riv[r][j].hi - riv[r][j].lo - kSsMwWinNew; left > 0 && rr > r0; left -= kSsMwWinNew) {
                    if (rk[rr - 1][k].used || (rr - 1 > r0 && rk[rr - 2][k].used)) { P.why = "a wide new column neighbourhood finds its sub-ring busy in the rounds before"; return; }
                    rk[rr][k].cont = true;
                    rk[rr - 1][k] = Rk{riv[r][j].lo, riv[r][j].lo, riv[r][j].lo, true, false};
                    rr--;
                }
            }
        }
        // (b) backward: a sub-ring takes in at most kSsMwWinNew columns per round, so what a later round needs beyond that comes in earlier
        for (int r = r1 - 2; r >= r0; r--)
            for (int k = 0; k < kSsMwRings; k++)
                if (rk[r][k].used && rk[r + 1][k].used && rk[r + 1][k].cont) rk[r][k].need = std::max(rk[r][k].need, rk[r + 1][k].need - kSsMwWinNew);
        // (c) forward: the intakes
        for (int r = r0; r < r1; r++) {
            int pairs = 0;
            for (int k = 0; k < kSsMwRings; k++) {
                const Rk& c = rk[r][k];
                if (!c.used) continue;
                if (c.need - c.lo > kSsMwCap) { P.why = "a column neighbourhood is wider than a sub-ring (with what it takes in ahead)"; return; }
                int from = c.lo;
                if (c.cont) {
                    const Rk& b = rk[r - 1][k];
                    from = std::min(c.need, std::max(b.need, c.lo)); // the part above what the sub-ring holds
                    // nothing round r - 1 still reads may lie where the intake goes: the sub-ring's columns 2048 below the new ones
                    if (c.need > from && c.need - kSsMwCap > b.lo) { P.why = "a column neighbourhood moves on faster than its sub-ring has room for"; return; }
                    if (c.need - from > kSsMwWinNew) { P.why = "a round brings a sub-ring more new columns than the window wave takes in at once"; return; }
                } else if (r > r0 && c.need - from > kSsMwWinNew) { P.why = "a new column neighbourhood is wider than the window wave takes in at once"; return; }
                P.winK[(size_t)r * kSsMwRings + k] = make_int2(from, c.need - from);
                pairs += (c.need - from + 1) / 2;
            }
            (void)pairs;
        }
    }
    // streams: workgroup by workgroup, wave by wave, round by round (as build_sstream_plan)
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
                for (int v = row0; v < std::min(nv, row0 + kSsSliceRows); v++) L = std::max(L, PT(v + 1) - PT(v));
                if (row0 < nv) P.max_slice_nnz = std::max(P.max_slice_nnz, PT(std::min(nv, row0 + kSsSliceRows)) - PT(row0));
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
    P.win.assign((size_t)rounds, make_int2(0, 0)); // (the one-window table of spmv_sstream: unused here)
    P.wg_halo.assign((size_t)nwg, 0);
    P.wg.assign((size_t)nwg, SsWg());
    for (int g = 0; g < nwg; g++) {
        SsWg& W = P.wg[g];
        W.r_begin = P.rptr[g];
        W.r_end = P.rptr[g + 1];
        for (int k = 0; k < 5; k++) W.t[k] = P.wptr[(size_t)g * 4 + k];
        W.link = -1;
    }
    P.eligible = true;
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
                            const int v = row0 + l + 64 * h; // (the lane's rows are l and l + 64 of its slice: neighbouring lanes then read neighbouring x entries of a mesh operator)
                            unsigned sh = kSsPad;
                            if (v < nv && j < PT(v + 1) - PT(v)) {
                                const int c = indcol[PT(v) + j];
                                int k = 0;
                                for (int q = 0; q < rnk[r]; q++)
                                    if (c >= riv[r][q].lo && c < riv[r][q].hi) k = kof[r][q];
                                sh = (unsigned)(k * kSsMwCap + (c & (kSsMwCap - 1)));
                            }
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

// replay of the plan against the matrix (host-only tests): the cut ring is simulated — every intake written where the kernel writes it —
// and every nonzero must find ITS column at its slot when its round runs; returns nullptr or the first violation
inline const char* check_sstream_mw_plan(const SsMwPlanHost& P, int n, const int* ptrow, const int* indcol)
{
    if (!P.eligible) return nullptr;
    const int shift = P.shift, nv = n + shift;
    auto PT = [&](int v) { const int i = v - shift; return ptrow[i < 0 ? 0 : (i > n ? n : i)]; };
    if (P.rptr[0] != 0 || P.rptr[P.nwg] != P.rounds) return "the workgroups' rounds do not cover the matrix";
    std::vector<int> ring((size_t)kSsRing);
    for (int g = 0; g < P.nwg; g++) {
        const SsWg& W = P.wg[g];
        if (W.r_begin != P.rptr[g] || W.r_end != P.rptr[g + 1]) return "a workgroup record disagrees with the round table";
        if (W.r_end - W.r_begin > kSsMwTabMax) return "too many rounds per workgroup";
        for (int k = 0; k < 5; k++)
            if (W.t[k] != P.wptr[(size_t)g * 4 + k]) return "a workgroup record disagrees with the stream table";
        std::fill(ring.begin(), ring.end(), -1);
        auto intake = [&](int r) -> const char* {
            int newcols = 0;
            for (int k = 0; k < kSsMwRings; k++) {
                const int2 w = P.winK[(size_t)r * kSsMwRings + k];
                if (w.y < 0 || w.y > kSsMwCap) return "an intake is wider than a sub-ring";
                if (r > P.rptr[g] && w.y > kSsMwWinNew) return "too many new columns for one sub-ring";
                newcols += w.y;
                for (int c = w.x; c < w.x + w.y; c++) ring[(size_t)k * kSsMwCap + (c & (kSsMwCap - 1))] = c;
            }
            (void)newcols;
            return nullptr;
        };
        if (P.rptr[g] < P.rptr[g + 1])
            if (const char* bad = intake(P.rptr[g])) return bad;
        for (int r = P.rptr[g]; r < P.rptr[g + 1]; r++) {
            // the window wave writes round r + 1's intake WHILE round r runs: round r must read right with it already in place
            if (r + 1 < P.rptr[g + 1])
                if (const char* bad = intake(r + 1)) return bad;
            for (int wv = 0; wv < 4; wv++) {
                const int row0 = r * kSsRound + wv * kSsSliceRows;
                const size_t base = (size_t)P.slice_step[(size_t)r * 4 + wv] * 64;
                int L = 1;
                for (int v = row0; v < std::min(nv, row0 + kSsSliceRows); v++) L = std::max(L, PT(v + 1) - PT(v));
                if (L != P.slice_len[(size_t)r * 4 + wv]) return "slice length disagrees with the rows";
                for (int j = 0; j < L; j++)
                    for (int l = 0; l < 64; l++) {
                        const unsigned s = P.slot[base + (size_t)j * 64 + l];
                        if (l == 0 && ((s & kSsFirst) != 0) != (j == 0)) return "slice-begin flag misplaced";
                        for (int h = 0; h < 2; h++) {
                            const int v = row0 + l + 64 * h; // (the lane's rows are l and l + 64 of its slice: neighbouring lanes then read neighbouring x entries of a mesh operator)
                            const unsigned sh = (s >> (16 * h)) & 0xffffu;
                            const bool real = v < nv && j < PT(v + 1) - PT(v);
                            if (!real) {
                                if (!(sh & kSsPad)) return "a padding place is not flagged";
                                continue;
                            }
                            if (sh & kSsPad) return "a nonzero is flagged as padding";
                            if (ring[sh & (kSsRing - 1)] != indcol[PT(v) + j]) return "a nonzero's slot does not hold its column when its round runs";
                        }
                    }
            }
        }
    }
    return nullptr;
}

// ---- device ------------------------------------------------------------------------------------------------------------------------
inline hipError_t ss_mw_upload(const SsMwPlanHost& P, SsDevice& Dv)
{
    hipError_t e = ss_upload(P, Dv, false);
    if (e != hipSuccess) return e;
    if ((e = hipMalloc(&Dv.winK, sizeof(int2) * P.winK.size())) != hipSuccess ||
        (e = hipMemcpy(Dv.winK, P.winK.data(), sizeof(int2) * P.winK.size(), hipMemcpyHostToDevice)) != hipSuccess) {
        ss_free(Dv);
        return e;
    }
    return hipSuccess;
}

// the four intakes of a round, wave-uniform
struct SsMw4 {
    int lo0, n0, lo1, n1, lo2, n2, lo3, n3;
};
__device__ __forceinline__ SsMw4 ss_mw_load(const int2* __restrict__ winK, int r)
{
    // SCALAR loads (the round number made provably uniform): a vector load here sits in the stream's queue, and loads return in order
    const int2* w = winK + 4 * (size_t)__builtin_amdgcn_readfirstlane(r);
    const int2 a = w[0], b = w[1], c = w[2], d = w[3];
    return SsMw4{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}
// element idx of the four intakes laid end to end: its column and its place in the cut ring (false: beyond the last)
__device__ __forceinline__ bool ss_mw_locate(const SsMw4& w, int idx, int& c, int& pos)
{
    const int p1 = w.n0, p2 = p1 + w.n1, p3 = p2 + w.n2, p4 = p3 + w.n3;
    const int k = (idx >= p1) + (idx >= p2) + (idx >= p3);
    const int pre = k == 0 ? 0 : (k == 1 ? p1 : (k == 2 ? p2 : p3));
    const int lo = k == 0 ? w.lo0 : (k == 1 ? w.lo1 : (k == 2 ? w.lo2 : w.lo3));
    c = lo + idx - pre;
    pos = k * kSsMwCap + (c & (kSsMwCap - 1));
    return idx < p4;
}

// Four waves, as spmv_sstream; the sub-rings take in a round's new columns TWO rounds ahead — loaded into registers during round r - 2,
// written into the cut ring during round r - 1 (the planner guarantees that nothing that round reads lies where they go), so that a round
// boundary is one barrier: ONE 16-byte load per thread and sub-ring (two neighbouring columns; at most 512 new columns per sub-ring and round — the
// planner spreads what exceeds that over earlier rounds), the workgroup's intake table staged in LDS at its start.
// What was tried on the way (the 5 M-row mesh; the same launch with the intake compiled out: 124 us): the four intakes laid end to end
// over the threads (comparison chains per element), eight 8-byte loads per thread, the table through vector loads consumed at once —
// 181 us (loads return in order: the table's wait drained the stream); the table a round ahead — 166; through scalar loads and four
// 16-byte loads per thread — 188 (a scalar load shares its counter with the LDS reads of the loop: they wait for it); a FIFTH wave that
// keeps the window by itself (one barrier per round, the stream waves' queues pure) — 246-292 us however cheap its address arithmetic:
// the two waves that then share a SIMD stand in each other's way, and everybody waits for them at the barrier.
template <int D, bool NT>
__global__ __launch_bounds__(256) void spmv_sstream_mw(SsMwView V, const double* __restrict__ x, double* __restrict__ y)
{
    const SsView& S = V.S;
    const int g = ss_logical_wg(S, (int)blockIdx.x);
    const SsWg W = S.wg[g]; // (uniform address: one scalar load)
    if (W.r_begin >= W.r_end) return;
    __shared__ double ring[kSsRing];
    __shared__ ss_v2d s_park[4 * kSsPark * 64];
    __shared__ int2 s_tab[kSsMwRings * kSsMwTabMax]; // this workgroup's rounds of the intake table
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_begin = W.r_begin, r_end = W.r_end;
    const int clast = S.ncols - 1;
    ss_v2d* park = s_park + wv * kSsPark * 64 + lane;
    int parked = 0, park_first = 0;
    const int t0 = wv == 0 ? W.t[0] : (wv == 1 ? W.t[1] : (wv == 2 ? W.t[2] : W.t[3]));
    const int t_end = wv == 0 ? W.t[1] : (wv == 1 ? W.t[2] : (wv == 2 ? W.t[3] : W.t[4]));
    const ss_v2d* vb = S.val + lane;
    const unsigned* sb = S.slot + lane;
    ss_v2d a[D];
    unsigned sl[D];
    int r = r_begin;
    ss_v2d nx[kSsMwRings];
    int2 wn[kSsMwRings]; // the intakes nx was loaded for
    auto issue = [&](int round) { // thread t: columns lo_k + 2 t, lo_k + 2 t + 1 of every sub-ring's intake
        const int2* tb = s_tab + kSsMwRings * (min(round, r_end - 1) - r_begin);
#pragma unroll
        for (int k = 0; k < kSsMwRings; k++) {
            wn[k] = tb[k];
            if (wn[k].y > 0) nx[k] = *reinterpret_cast<const ss_v2d_u*>(x + min(max(wn[k].x + 2 * tid, 0), clast - 1)); // (wave-uniform branch; 16 bytes at an 8-byte aligned address)
        }
    };
    auto take_in = [&]() {
#pragma unroll
        for (int k = 0; k < kSsMwRings; k++) {
            const int c = wn[k].x + 2 * tid, j = 2 * tid;
            // (a pair clamped at the vector's end was loaded one column down: its first word is then the neighbour's)
            if (j < wn[k].y) ring[k * kSsMwCap + (c & (kSsMwCap - 1))] = c > clast - 1 ? nx[k].y : nx[k].x;
            if (j + 1 < wn[k].y) ring[k * kSsMwCap + ((c + 1) & (kSsMwCap - 1))] = nx[k].y;
        }
    };
    {
        // the first fill — every sub-ring's whole interval — in flight at once, IN FRONT of the stream's first D steps (spmv_sstream.hpp says why)
        const SsMw4 w0 = ss_mw_load(V.winK, r_begin);
        const int total = w0.n0 + w0.n1 + w0.n2 + w0.n3;
        double fx[kSsMwFill];
#pragma unroll
        for (int u = 0; u < kSsMwFill; u++) {
            int c, pos;
            (void)ss_mw_locate(w0, tid + 256 * u, c, pos);
            fx[u] = x[min(max(c, 0), clast)];
        }
        if (r_begin + 1 < r_end) { // ... and the second round's intake with it (the table is not in LDS yet)
            const SsMw4 w1 = ss_mw_load(V.winK, r_begin + 1);
            wn[0] = make_int2(w1.lo0, w1.n0);
            wn[1] = make_int2(w1.lo1, w1.n1);
            wn[2] = make_int2(w1.lo2, w1.n2);
            wn[3] = make_int2(w1.lo3, w1.n3);
#pragma unroll
            for (int k = 0; k < kSsMwRings; k++)
                if (wn[k].y > 0) nx[k] = *reinterpret_cast<const ss_v2d_u*>(x + min(max(wn[k].x + 2 * tid, 0), clast - 1));
        }
        int2 tb[2] = {make_int2(0, 0), make_int2(0, 0)};
        const int nt = kSsMwRings * (r_end - r_begin);
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (tid + 256 * u < nt) tb[u] = V.winK[(size_t)kSsMwRings * r_begin + tid + 256 * u];
#pragma unroll
        for (int d = 0; d < D; d++) {
            a[d] = NT ? __builtin_nontemporal_load(vb + (size_t)(t0 + d) * 64) : vb[(size_t)(t0 + d) * 64];
            sl[d] = sb[(size_t)(t0 + d) * 64];
        }
#pragma unroll
        for (int u = 0; u < kSsMwFill; u++) {
            int c, pos;
            if (ss_mw_locate(w0, tid + 256 * u, c, pos)) ring[pos] = fx[u];
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (tid + 256 * u < nt) s_tab[tid + 256 * u] = tb[u];
        for (int i0 = 512 + tid; i0 < nt; i0 += 256) s_tab[i0] = V.winK[(size_t)kSsMwRings * r_begin + i0]; // (more than 128 rounds per workgroup)
        if (r_begin + 1 < r_end) take_in();
        for (int i0 = 256 * kSsMwFill + tid; i0 < total; i0 += 256 * 8) { // (first fills of more than 4096 columns)
            double f8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int c, pos;
                (void)ss_mw_locate(w0, i0 + 256 * u, c, pos);
                f8[u] = x[min(max(c, 0), clast)];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int c, pos;
                if (ss_mw_locate(w0, i0 + 256 * u, c, pos)) ring[pos] = f8[u];
            }
        }
    }
    __syncthreads(); // (the table is in LDS; the first fill is in place)
    // Round r + 1's columns are WRITTEN while round r runs — nothing round r reads lies where they go (the planner's promise, replayed by
    // check_sstream_mw_plan) — so a round boundary is ONE barrier, and the writes are off the path between barriers.
    if (r_begin + 1 < r_end) issue(r_begin + 2);
    double acc0 = 0.0, acc1 = 0.0;
    auto store = [&](int round, ss_v2d v) {
        const int v0 = round * kSsRound + wv * kSsSliceRows + lane, v1 = v0 + 64; // the lane's view rows: two 8-byte stores, each a wave's 512 contiguous bytes
        if (S.rowmap) { // (wave-uniform) mapped rows: wherever the map sends them
            if (v0 >= S.shift && v0 < S.n) y[S.rowmap[v0 - S.shift]] = v.x;
            if (v1 < S.n) y[S.rowmap[v1 - S.shift]] = v.y;
        } else {
            if (v0 >= S.shift && v0 < S.n) y[v0] = v.x;
            if (v1 < S.n) y[v1] = v.y;
        }
    };
    auto flush = [&]() {
        for (int j = 0; j < parked; j++) store(park_first + j, park[j * 64]);
        parked = 0;
    };
    auto emit = [&]() { // this wave's slice of round r is complete
        if (parked == 0) park_first = r;
        park[parked * 64] = ss_v2d{acc0, acc1};
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
                    __syncthreads(); // every wave is through with round r - 1, and round r's columns (written during it) are in place
                    if (r + 1 < r_end) {
                        take_in();    // round r + 1's, loaded during round r - 1
                        issue(r + 2);
                    }
                }
                const double x0 = ring[s & (kSsRing - 1)], x1 = ring[(s >> 16) & (kSsRing - 1)];
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
