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
//
// Round 5 (tools/sstream_trace.hip, profiles/r05_sstream_trace_before.txt: s_memrealtime per workgroup at 1 M rows): the loop streams
// at 7.1 TB/s of real bytes — the fabric's rate — but 4.4 of the launch's 28 us went by BEFORE the loop: the workgroup's plan came
// through a chain of dependent loads (round range -> stream offsets -> windows) and the first window (4 500 columns) was filled by a
// loop of ONE load per thread and iteration, each drained before its LDS store: eighteen round trips.  Now: ONE 64-byte record per
// workgroup (a single scalar load) names everything the prologue needs, and the whole first window is in flight at once (24 loads
// per thread, issued in front of the stream's first D steps so that the window lands first and is written to LDS while the stream
// arrives).  The same round: a ROW SHIFT (a row-mapped piece whose rows go to y[r + odd offset] is planned one row down, so that its
// row pairs stay 16-byte aligned: the interior rows of a partition's rank), workgroups dealt their rounds so that the ones
// dispatched last get the shorter share, and the FUSED form: a rank's whole step of the peer-push exchange (push_exchange.hpp) in
// one launch of THIS kernel, as spmv_csr_ring<..., FUSED> has been since round 2.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "spmv_ring.hpp" // RingComm, ring_ldx, ring_push_link, ring_push_gate: the protocol pieces of the fused multi-GPU step

namespace mi355 {

typedef double ss_v2d __attribute__((ext_vector_type(2)));
constexpr int kSsRing = 8192;            // x entries in the LDS ring (64 KB)
constexpr int kSsSliceRows = 128;        // rows per slice = two per lane
constexpr int kSsRound = 512;            // rows per round of a workgroup: four waves' slices
constexpr int kSsNewMax = 1024;          // new window columns per round (four per thread, a round ahead in registers)
constexpr int kSsPark = 20;              // slices of y parked per wave: 4 x 20 KiB beside the 64 KiB ring
#ifndef MI355_SS_TAIL
#define MI355_SS_TAIL 2
#endif
constexpr int kSsTail = MI355_SS_TAIL;   // rounds of a workgroup whose sums stay parked until its end (the others are stored earlier; tools/ A/B it with -DMI355_SS_TAIL=n)
constexpr unsigned kSsPad = 0x8000u;     // slot flag (either half): padding place, not multiplied
constexpr unsigned kSsFirst = 0x4000u;   // slot flag (low half): first step of a slice
constexpr int kSsPadSteps = 64;          // steps of padding behind the last one: the stream's loads run ahead unclamped (D <= 16)
constexpr int kSsMaxWgs = 256;           // one workgroup per CU
constexpr int kSsFill = 24;              // first-window columns per thread loaded in ONE batch (6144 columns; wider windows take further batches of 8)
constexpr int kSsGhostSlack = 3;         // rounds a ghost-reading workgroup of the fused step gets less than its share.  Its push (write-through stores, drain,
                                         // flag), its wait (poll + system-scope acquire) and the uncached window loads come first: 3.8 us in front of the loop
                                         // without push duty, 6.5-7.2 with (a plain workgroup: 2.3), and its rounds take 3.4 us instead of 2.7 (the window's
                                         // loads return in order with the stream's) — sim_rank 8 1's trace, profiles/r05_sim_rank.txt: with a slack of two rounds
                                         // the pushers were the launch's last workgroups in one launch of ten (20.1 us against 17.8)

// everything a workgroup's prologue needs, as ONE 64-byte record (one scalar load)
struct SsWg {
    int r_begin, r_end; // its rounds
    int w0_lo, w0_n;    // first window fill: columns [w0_lo, w0_lo + w0_n)
    int w1_lo, w1_n;    // the new columns of its second round (its last round's if it has one round only)
    int t[5];           // first step of each wave's stream, and the end of the last wave's
    int halo;           // fused multi-GPU step: its windows hold a ghost column (it waits for the neighbours' entries first)
    int link;           // fused multi-GPU step: first push link it serves (then every npush_runs-th), or -1 (written at connect time: capi_part.hip)
    int pad[3];
};
static_assert(sizeof(SsWg) == 64, "one record = one 64-byte scalar load");

struct SsView {
    const ss_v2d* val;     // [steps + kSsPadSteps][64]
    const unsigned* slot;  // [steps + kSsPadSteps][64]: low half = row 2l, high half = row 2l + 1
    const SsWg* wg;        // [nwg]
    const int2* win;       // [rounds] {first new column, count} the window takes in before the round (a workgroup's first round: its first fill)
    int nwg, n, ncols;     // n = rows of the VIEW (the handle's rows + shift)
    const int* rowmap;     // nullptr, or the handle's row i writes y[rowmap[i]] (partition pieces, the relabelled twin of reorder.hpp)
    int shift;             // 0 | 1: view row v is the handle's row v - shift (view row 0 does not exist when shift = 1); unmapped: y points `shift` in front
    unsigned long long* trace = nullptr; // tools/sstream_trace.hip only (ABL & 8): four s_memrealtime stamps per workgroup
};

// ---- host: the plan --------------------------------------------------------------------------------------------------------------
struct SsPlanHost {
    bool eligible = false;
    const char* why = "";            // first reason the matrix is not eligible
    int nwg = 0, rounds = 0, shift = 0;
    long long steps = 0, pad_places = 0;
    int max_slice_nnz = 0;           // the longest slice's CSR segment (sstream_fill_values picks its LDS buffer by it)
    std::vector<int> wptr, rptr, slice_step, slice_len; // slice_step / slice_len[4 * round + wave] = first step / steps of that slice (value refills)
    std::vector<int2> win;
    std::vector<SsWg> wg;
    std::vector<int> wg_halo;        // ghost columns given: per workgroup, its windows hold a ghost column (the fused step: it waits for the neighbours)
    bool fusable = false;            // ghost columns given, and every such workgroup takes its whole column range in with its first fill (spmv_sstream_fused can run the plan)
    std::vector<unsigned> slot;      // [steps + kSsPadSteps][64]
};

// build the plan of an n x ncols pattern (columns ascending or not: a row's nonzeros keep their CSR order); max_padding = padded places per
// nonzero allowed; shift: plan the rows one down (see SsView); columns outside [ghost_lo, ghost_hi) are ghosts when ghost_lo < ghost_hi
inline void build_sstream_plan(int n, int ncols, const int* ptrow, const int* indcol, double max_padding, SsPlanHost& P, bool want_slots = true, int shift = 0,
                               int ghost_lo = 0, int ghost_hi = 0)
{
    P = SsPlanHost();
    const long long nnz = n > 0 ? ptrow[n] : 0;
    if (n <= 0 || nnz <= 0) { P.why = "empty matrix"; return; }
    if (shift != 0 && shift != 1) { P.why = "bad shift"; return; }
    P.shift = shift;
    const int nv = n + shift; // rows of the view
    auto PT = [&](int v) { const int i = v - shift; return ptrow[i < 0 ? 0 : (i > n ? n : i)]; }; // first nonzero of view row v
    const int rounds = (nv + kSsRound - 1) / kSsRound;
    int nwg = std::min(kSsMaxWgs, rounds);
    if (nwg >= 8) nwg = nwg / 8 * 8; // a multiple of the XCD count keeps the workgroup -> XCD dealing of the kernel
    P.nwg = nwg;
    P.rounds = rounds;
    const bool ghosts = ghost_lo < ghost_hi;
    // column extent of every round
    std::vector<int> cmin((size_t)rounds, 0x7fffffff), cmax((size_t)rounds, -1);
    for (int r = 0; r < rounds; r++) {
        int lo = 0x7fffffff, hi = -1;
        for (int k = PT(r * kSsRound); k < PT(std::min(nv, (r + 1) * kSsRound)); k++) {
            lo = std::min(lo, indcol[k]);
            hi = std::max(hi, indcol[k]);
        }
        cmin[r] = lo;
        cmax[r] = hi;
    }
    // Rounds per workgroup.  The kernel takes block b as logical workgroup (b % 8) * (nwg / 8) + b / 8 (blocks are dealt round-robin to the
    // XCDs: the workgroups of one XCD stream neighbouring rows), and a launch's blocks start in block order over ~1 us: inside each XCD's
    // chunk of nwg / 8 consecutive logical workgroups the FIRST ones (dispatched first) take the rounds that do not divide evenly.
    // less[g] > 0: workgroup g gets that many rounds less (never below one) and they go to the least loaded others.
    auto deal = [&](const std::vector<int>& less) {
        std::vector<int> cnt((size_t)nwg, rounds / nwg);
        const int extra = rounds % nwg;
        const int per = nwg % 8 == 0 ? nwg / 8 : nwg, chunks = nwg / per;
        for (int xcd = 0; xcd < chunks; xcd++) {
            const int e = (int)((long long)extra * (xcd + 1) / chunks - (long long)extra * xcd / chunks);
            for (int j = 0; j < e; j++) cnt[(size_t)xcd * per + j]++;
        }
        int pool = 0;
        for (int g = 0; g < nwg; g++) {
            const int take = std::min(less[g], cnt[g] - 1);
            if (take > 0) { cnt[g] -= take; pool += take; }
        }
        while (pool > 0) { // to the least loaded workgroup that gives nothing itself (ties: the one dispatched earliest)
            int best = -1;
            for (int g = 0; g < nwg; g++)
                if (!less[g] && (best < 0 || cnt[g] < cnt[best] || (cnt[g] == cnt[best] && g % per < best % per))) best = g;
            if (best < 0) { // everybody gives: hand the rounds back in order
                for (int g = 0; g < nwg && pool > 0; g++) { cnt[g]++; pool--; }
                break;
            }
            cnt[best]++;
            pool--;
        }
        P.rptr.assign((size_t)nwg + 1, 0);
        for (int g = 0; g < nwg; g++) P.rptr[g + 1] = P.rptr[g] + cnt[g];
    };
    // windows: per workgroup a monotone upper end; everything a round names must lie within kSsRing below it.  A workgroup whose
    // windows hold a ghost column (the fused multi-GPU step) takes its WHOLE column range in with its first fill — ghosts are then
    // read (from the receive window: uncached memory, a select per load) in its prologue only, and its loop is the plain
    // kernel's, load for load — which needs that range to fit the ring: too_wide[g] says it does not (yet).
    std::vector<char> too_wide((size_t)nwg, 0);
    auto windows = [&]() -> bool {
        P.win.assign((size_t)rounds, make_int2(0, 0));
        P.wg_halo.assign((size_t)nwg, 0);
        std::fill(too_wide.begin(), too_wide.end(), 0);
        for (int g = 0; g < nwg; g++) {
            int allmin = 0x7fffffff;
            for (int r = P.rptr[g]; r < P.rptr[g + 1]; r++) allmin = std::min(allmin, cmin[r]);
            int whi = 0, wlo = 0;
            for (int r = P.rptr[g]; r < P.rptr[g + 1]; r++) {
                const int nhi = std::max(whi, cmax[r] + 1);
                if (r == P.rptr[g]) {
                    const int lo = std::max(std::max(0, nhi - kSsRing), std::min(allmin, nhi));
                    P.win[r] = make_int2(lo, nhi - lo);
                    wlo = lo;
                } else {
                    P.win[r] = make_int2(whi, nhi - whi);
                    if (nhi - whi > kSsNewMax) { P.why = "a round brings more new columns than the window takes in at once"; return false; }
                }
                whi = nhi;
                if (cmin[r] != 0x7fffffff && cmin[r] < whi - kSsRing) { P.why = "a round's rows reach further apart than the LDS ring holds"; return false; }
            }
            if (ghosts && whi > wlo && (wlo < ghost_lo || whi > ghost_hi)) {
                P.wg_halo[g] = 1;
                if (whi - wlo > kSsRing) too_wide[g] = 1;
                else { // everything up front: later rounds bring nothing
                    P.win[P.rptr[g]] = make_int2(wlo, whi - wlo);
                    for (int r = P.rptr[g] + 1; r < P.rptr[g + 1]; r++) P.win[r] = make_int2(whi, 0);
                }
            }
        }
        return true;
    };
    std::vector<int> less((size_t)nwg, 0);
    deal(less);
    if (!windows()) return;
    if (ghosts)
        for (int it = 0; it < 64; it++) { // who reads ghosts depends on the dealing and the dealing on who reads ghosts (a workgroup that
            // gives rounds away moves its neighbour's rows towards the cut, never away from it): a few passes settle it
            std::vector<int> want = less;
            bool changed = false;
            for (int g = 0; g < nwg; g++) {
                if (!P.wg_halo[g]) continue;
                const int need = too_wide[g] ? std::max(less[g] + 1, kSsGhostSlack) : std::max(less[g], kSsGhostSlack);
                if (need != less[g]) { want[g] = need; changed = true; }
            }
            if (!changed) break;
            less = want;
            deal(less);
            if (!windows()) return; // (the marks always describe the dealing in force, whatever the loop's last `less` was)
        }
    P.fusable = ghosts;
    for (int g = 0; g < nwg; g++)
        if (P.wg_halo[g] && too_wide[g]) P.fusable = false; // (a ghost-reading workgroup's columns do not fit the ring even with one round: the ring kernel's FUSED form serves such a piece)
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
    P.wg.assign((size_t)nwg, SsWg());
    for (int g = 0; g < nwg; g++) {
        SsWg& W = P.wg[g];
        W.r_begin = P.rptr[g];
        W.r_end = P.rptr[g + 1];
        if (W.r_begin < W.r_end) {
            const int2 w0 = P.win[W.r_begin], w1 = P.win[std::min(W.r_begin + 1, W.r_end - 1)];
            W.w0_lo = w0.x;
            W.w0_n = w0.y;
            W.w1_lo = w1.x;
            W.w1_n = w1.y;
        }
        for (int k = 0; k < 5; k++) W.t[k] = P.wptr[(size_t)g * 4 + k];
        W.halo = P.wg_halo[g];
        W.link = -1;
    }
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
                            const int v = row0 + 2 * l + h;
                            unsigned sh = kSsPad;
                            if (v < nv && j < PT(v + 1) - PT(v)) sh = (unsigned)(indcol[PT(v) + j] & (kSsRing - 1));
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
// the window when its round runs; the padding places are flagged; the workgroup records say what the tables say; a workgroup whose
// windows hold a ghost column is marked; returns nullptr or the first violation
inline const char* check_sstream_plan(const SsPlanHost& P, int n, const int* ptrow, const int* indcol, int ghost_lo = 0, int ghost_hi = 0)
{
    if (!P.eligible) return nullptr;
    const int shift = P.shift, nv = n + shift;
    auto PT = [&](int v) { const int i = v - shift; return ptrow[i < 0 ? 0 : (i > n ? n : i)]; };
    if (P.rptr[0] != 0 || P.rptr[P.nwg] != P.rounds) return "the workgroups' rounds do not cover the matrix";
    for (int g = 0; g < P.nwg; g++) {
        int wlo = 0, whi = 0;
        const SsWg& W = P.wg[g];
        if (W.r_begin != P.rptr[g] || W.r_end != P.rptr[g + 1] || W.r_end < W.r_begin) return "a workgroup record disagrees with the round table";
        for (int k = 0; k < 5; k++)
            if (W.t[k] != P.wptr[(size_t)g * 4 + k]) return "a workgroup record disagrees with the stream table";
        if (W.r_begin < W.r_end) {
            const int2 w0 = P.win[W.r_begin], w1 = P.win[std::min(W.r_begin + 1, W.r_end - 1)];
            if (W.w0_lo != w0.x || W.w0_n != w0.y || W.w1_lo != w1.x || W.w1_n != w1.y) return "a workgroup record disagrees with the window table";
        }
        bool names_ghost = false;
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
                for (int v = row0; v < std::min(nv, row0 + kSsSliceRows); v++) L = std::max(L, PT(v + 1) - PT(v));
                if (L != P.slice_len[(size_t)r * 4 + wv]) return "slice length disagrees with the rows";
                for (int j = 0; j < L; j++)
                    for (int l = 0; l < 64; l++) {
                        const unsigned s = P.slot[base + (size_t)j * 64 + l];
                        if (l == 0 && ((s & kSsFirst) != 0) != (j == 0)) return "slice-begin flag misplaced";
                        for (int h = 0; h < 2; h++) {
                            const int v = row0 + 2 * l + h;
                            const unsigned sh = (s >> (16 * h)) & 0xffffu;
                            const bool real = v < nv && j < PT(v + 1) - PT(v);
                            if (!real) {
                                if (!(sh & kSsPad)) return "a padding place is not flagged";
                                continue;
                            }
                            const int c = indcol[PT(v) + j];
                            if (sh & kSsPad) return "a nonzero is flagged as padding";
                            if ((sh & (kSsRing - 1)) != (unsigned)(c & (kSsRing - 1))) return "slot is not the column's ring slot";
                            if (c < wlo || c >= whi) return "a column lies outside the window when its round runs";
                            names_ghost = names_ghost || (ghost_lo < ghost_hi && (c < ghost_lo || c >= ghost_hi));
                        }
                    }
            }
        }
        if (names_ghost && !P.wg_halo[g]) return "a workgroup names a ghost column and is not marked";
        if (P.fusable && P.wg_halo[g])
            for (int r = P.rptr[g] + 1; r < P.rptr[g + 1]; r++)
                if (P.win[r].y != 0) return "a ghost-reading workgroup takes columns in after its first fill";
        if (W.halo != P.wg_halo[g]) return "a workgroup record's ghost mark disagrees with the plan's";
    }
    return nullptr;
}

// ---- device ------------------------------------------------------------------------------------------------------------------------
// (Re)fills the sliced values from CSR values (setup and value refreshes — a Newton loop's Jacobian, src/solve_newton.c:1245-1247 —, never
// per product): one workgroup per slice.  The slice's CSR segment (128 consecutive rows: contiguous) is read ONCE, coalesced, into LDS —
// and, when csr_out is given, written on to the handle's CSR value array in the same pass (mi_csr_update_values_dev: one read of the
// caller's values instead of a device-to-device copy followed by a strided re-read) — and leaves as whole 1-KiB steps.  CAP = LDS doubles
// per workgroup (0: slices longer than any buffer read their rows straight from src).
// Round 4's form (one wave per slice, every lane striding through its own row) took 1.39 ms at C4 — ten products.
template <int CAP>
__global__ __launch_bounds__(256) void sstream_fill_kernel(int nslices, int n, int shift, const int* __restrict__ ptrow, const double* __restrict__ src,
                                                           double* __restrict__ csr_out, const int* __restrict__ slice_step, const int* __restrict__ slice_len,
                                                           ss_v2d* __restrict__ val, int pair64)
{
    __shared__ double buf[CAP > 0 ? CAP : 1];
    __shared__ int rp[kSsSliceRows + 1];
    const int tid = threadIdx.x;
    for (int sidx = blockIdx.x; sidx < nslices; sidx += gridDim.x) { // sidx = 4 * round + wave: view rows [128 * sidx, 128 * sidx + 128)
        const int row0 = kSsSliceRows * sidx - shift;
        if (tid <= kSsSliceRows) rp[tid] = ptrow[max(0, min(row0 + tid, n))];
        __syncthreads();
        const int base = rp[0], seg = rp[kSsSliceRows] - base;
        const bool staged = CAP > 0 && seg <= CAP; // (workgroup-uniform)
        if (staged) {
            for (int k = tid; k < seg; k += 256) {
                const double v = src[base + k];
                buf[k] = v;
                if (csr_out) csr_out[base + k] = v;
            }
            __syncthreads();
        } else if (csr_out) {
            for (int k = tid; k < seg; k += 256) csr_out[base + k] = src[base + k];
        }
        const int t0 = slice_step[sidx], L = slice_len[sidx];
        for (int q = tid; q < L * 64; q += 256) {
            const int j = q >> 6, lane = q & 63;
            // the lane's two rows: 2 l and 2 l + 1, or (pair64: the cut-ring form, spmv_sstream_mw.hpp) l and l + 64
            const int ra = pair64 ? lane : 2 * lane, rb = pair64 ? lane + 64 : 2 * lane + 1;
            const int a0 = rp[ra], a1 = rp[ra + 1], b0 = rp[rb], b1 = rp[rb + 1];
            ss_v2d v = {0.0, 0.0};
            if (j < a1 - a0) v.x = staged ? buf[a0 - base + j] : src[a0 + j];
            if (j < b1 - b0) v.y = staged ? buf[b0 - base + j] : src[b0 + j];
            val[(size_t)(t0 + j) * 64 + lane] = v;
        }
        __syncthreads(); // rp / buf are rewritten for the next slice
    }
}

// host: enqueue the fill of all 4 * rounds slices on stream s; max_slice_nnz = the longest slice's CSR segment (SsPlanHost)
inline void sstream_fill_values(int rounds, int n, int shift, const int* d_ptrow, const double* d_src, double* d_csr_out, const int* d_slice_step,
                                const int* d_slice_len, ss_v2d* d_val, int max_slice_nnz, hipStream_t s, int pair64 = 0)
{
    const int nslices = 4 * rounds;
    if (nslices <= 0) return;
    if (max_slice_nnz <= 2048) // (S15: 1920 values per slice) 16 KB of LDS: eight workgroups per CU
        hipLaunchKernelGGL((sstream_fill_kernel<2048>), dim3((unsigned)std::min(nslices, 2048)), dim3(256), 0, s, nslices, n, shift, d_ptrow, d_src, d_csr_out, d_slice_step, d_slice_len, d_val, pair64);
    else if (max_slice_nnz <= 8192) // (the FE rows of 56: 7168) 64 KB: two per CU
        hipLaunchKernelGGL((sstream_fill_kernel<8192>), dim3((unsigned)std::min(nslices, 1024)), dim3(256), 0, s, nslices, n, shift, d_ptrow, d_src, d_csr_out, d_slice_step, d_slice_len, d_val, pair64);
    else
        hipLaunchKernelGGL((sstream_fill_kernel<0>), dim3((unsigned)std::min(nslices, 2048)), dim3(256), 0, s, nslices, n, shift, d_ptrow, d_src, d_csr_out, d_slice_step, d_slice_len, d_val, pair64);
}

// device copy of a plan (library and tools/ alike): allocate + upload; on failure everything is released and the error returned
struct SsDevice {
    ss_v2d* val = nullptr;
    unsigned* slot = nullptr;
    SsWg* wg = nullptr;
    int2* win = nullptr;
    int* slice_step = nullptr;
    int* slice_len = nullptr;
    int* wg_halo = nullptr; // plans with ghost columns only
    int2* winK = nullptr;   // the cut-ring form only (spmv_sstream_mw.hpp): [4 * rounds] intakes per sub-ring
};
inline void ss_free(SsDevice& Dv)
{
    (void)hipFree(Dv.winK);
    (void)hipFree(Dv.val); (void)hipFree(Dv.slot); (void)hipFree(Dv.wg); (void)hipFree(Dv.win); (void)hipFree(Dv.slice_step); (void)hipFree(Dv.slice_len); (void)hipFree(Dv.wg_halo);
    Dv = SsDevice();
}
inline hipError_t ss_upload(const SsPlanHost& P, SsDevice& Dv, bool ghosts)
{
    hipError_t e;
    const size_t vbytes = sizeof(ss_v2d) * (size_t)(P.steps + kSsPadSteps) * 64;
    if ((e = hipMalloc(&Dv.val, vbytes)) != hipSuccess || (e = hipMalloc(&Dv.slot, sizeof(unsigned) * P.slot.size())) != hipSuccess ||
        (e = hipMalloc(&Dv.wg, sizeof(SsWg) * P.wg.size())) != hipSuccess || (e = hipMalloc(&Dv.win, sizeof(int2) * P.win.size())) != hipSuccess ||
        (e = hipMalloc(&Dv.slice_step, sizeof(int) * P.slice_step.size())) != hipSuccess || (e = hipMalloc(&Dv.slice_len, sizeof(int) * P.slice_len.size())) != hipSuccess ||
        (ghosts && (e = hipMalloc(&Dv.wg_halo, sizeof(int) * P.wg_halo.size())) != hipSuccess) ||
        (e = hipMemset((char*)Dv.val + sizeof(ss_v2d) * (size_t)P.steps * 64, 0, sizeof(ss_v2d) * (size_t)kSsPadSteps * 64)) != hipSuccess ||
        (e = hipMemcpy(Dv.slot, P.slot.data(), sizeof(unsigned) * P.slot.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(Dv.wg, P.wg.data(), sizeof(SsWg) * P.wg.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(Dv.win, P.win.data(), sizeof(int2) * P.win.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(Dv.slice_step, P.slice_step.data(), sizeof(int) * P.slice_step.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(Dv.slice_len, P.slice_len.data(), sizeof(int) * P.slice_len.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (ghosts && (e = hipMemcpy(Dv.wg_halo, P.wg_halo.data(), sizeof(int) * P.wg_halo.size(), hipMemcpyHostToDevice)) != hipSuccess)) {
        ss_free(Dv);
        return e;
    }
    return hipSuccess;
}

// D steps of the stream in flight per lane; ONE workgroup of four waves per CU.  Workgroup b is taken as logical workgroup
// (b % 8) * (G / 8) + b / 8 so that the workgroups that share an XCD stream neighbouring rows (their x lines meet in that XCD's L2).
// ABL (tools/sstream_ablate.hip only; invalid results): 1 no LDS gather, 2 no y stores, 4 no new-column loads / window refills at the round boundaries;
// 8 (tools/sstream_trace.hip; valid results): s_memrealtime at the workgroup's start, in front of its loop, behind it and at its end.
// FUSED (mi_part_spmv_push_dev): a rank's whole step of the peer-push exchange in this launch — the columns are numbered [ghosts of
// lower ranks | owned | ghosts of higher ranks] (partition.hpp: build_combined), ghosts are read from this rank's receive window
// (C.halo), a workgroup whose windows hold a ghost column (SsWg::halo) first does its share of the push (SsWg::link) and then
// waits — bounded, loud — for every neighbour's flag of this step, its stream's first loads already in flight.
// MODE 0: the plain product.  MODE 1 / 2: a workgroup of the fused multi-GPU step that reads no ghost column (x is the owned part of
// x_ext, columns offset by C.n_left: the plain kernel's code, nothing more — the first fused form ran EVERY workgroup through the
// ghost-column selects, push and wait code: 21 KB of ISA against 13.7, 20.3 us per step against 18.7 for the same rows unfused) /
// one that does (push duty, wait, ghost columns from the receive window).
template <int D, bool NT, int ABL, int MODE>
__device__ __forceinline__ void ss_body(const SsView& S, const SsWg& W, int g, const double* __restrict__ x_in, double* __restrict__ y, const RingComm& C,
                                        double* ring /* [kSsRing], LDS */, ss_v2d* s_park /* [4 * kSsPark * 64], LDS */)
{
    constexpr bool GHOSTS = MODE == 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* __restrict__ x = MODE != 0 ? x_in - C.n_left : x_in; // (fused step: column c is owned entry c - n_left; ghost columns are read in MODE 2's first fill only)
    const int r_begin = W.r_begin, r_end = W.r_end;
    if ((ABL & 8) && tid == 0) S.trace[4 * g] = __builtin_amdgcn_s_memrealtime();
    // (ghost mark and push link ride in the workgroup's record: two further dependent scalar loads in front of every workgroup's
    // first vector load cost the whole launch ~1.5 us — first form of this kernel, sim_rank 8 1)
    if (GHOSTS && W.link >= 0 && C.npush_runs > 0) // push duty of this workgroup, before anything that could wait
        for (int l = W.link; l < C.n_links; l += C.npush_runs) ring_push_link<256>(C, x_in, l);
    ss_v2d* park = s_park + wv * kSsPark * 64 + lane;
    int parked = 0, park_first = 0; // (wave-uniform) the slices of rounds park_first .. park_first + parked - 1 are parked
    const int t0 = wv == 0 ? W.t[0] : (wv == 1 ? W.t[1] : (wv == 2 ? W.t[2] : W.t[3]));
    const int t_end = wv == 0 ? W.t[1] : (wv == 1 ? W.t[2] : (wv == 2 ? W.t[3] : W.t[4]));
    const int clast = S.ncols - 1;
    // (fused step: the prefetch of a window's new columns reaches neither in front of nor behind the OWNED entries — whatever lies
    // beyond is never used (a plain workgroup names no ghost; a ghost reader's later windows are empty), and the caller's x need not
    // extend there)
    const int cfirst = MODE != 0 ? C.n_left : 0;
    const int cowned = MODE != 0 ? C.n_left + C.n_local - 1 : clast;
    const ss_v2d* vb = S.val + lane;
    const unsigned* sb = S.slot + lane;
    ss_v2d a[D];
    unsigned sl[D];
    auto first_steps = [&]() {
#pragma unroll
        for (int d = 0; d < D; d++) {
            a[d] = NT ? __builtin_nontemporal_load(vb + (size_t)(t0 + d) * 64) : vb[(size_t)(t0 + d) * 64];
            sl[d] = sb[(size_t)(t0 + d) * 64];
        }
    };
    if (GHOSTS) { // the stream's first steps travel while this workgroup waits for its neighbours' entries
        first_steps();
        push_wait_flags(C.flags, C.nb, C.n_nb, C.step, 0u, C.timeouts, C.spin_max, tid, 256);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    // The whole first window in flight at once, IN FRONT of the stream's first D steps: loads return in order, so the window lands
    // first and is written to LDS while the stream's steps arrive (one memory latency in front of the loop).
    int r = r_begin; // the round this wave's current slice belongs to
    double nx[kSsNewMax / 256]; // the NEXT round's new columns, a round ahead in registers
    int2 wn = make_int2(W.w1_lo, W.w1_n);
    {
        // (MODE 2: the first fill is the workgroup's WHOLE column range — the planner saw to it — so that this is the only place where
        // a column can be a ghost; the next-round prefetch below and in the loop loads owned entries like the plain kernel's, into
        // windows of zero new columns)
        double fx[kSsFill];
#pragma unroll
        for (int u = 0; u < kSsFill; u++) fx[u] = GHOSTS ? ring_ldx<true>(x_in, C, min(W.w0_lo + tid + 256 * u, clast)) : x[min(W.w0_lo + tid + 256 * u, cowned)];
#pragma unroll
        for (int u = 0; u < kSsNewMax / 256; u++) nx[u] = (ABL & 4) ? 0.0 : x[min(max(wn.x + tid + 256 * u, cfirst), cowned)];
        if (!GHOSTS) first_steps();
#pragma unroll
        for (int u = 0; u < kSsFill; u++) {
            const int c = W.w0_lo + tid + 256 * u;
            if (c < W.w0_lo + W.w0_n) ring[c & (kSsRing - 1)] = fx[u];
        }
        for (int c0 = W.w0_lo + 256 * kSsFill + tid; c0 < W.w0_lo + W.w0_n; c0 += 256 * 8) { // (a first window wider than 6144 columns)
            double f8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) f8[u] = GHOSTS ? ring_ldx<true>(x_in, C, min(c0 + 256 * u, clast)) : x[min(c0 + 256 * u, cowned)];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (c0 + 256 * u < W.w0_lo + W.w0_n) ring[(c0 + 256 * u) & (kSsRing - 1)] = f8[u];
        }
    }
    __syncthreads();
    if ((ABL & 8) && tid == 0) S.trace[4 * g + 1] = __builtin_amdgcn_s_memrealtime();
    double acc0 = 0.0, acc1 = 0.0;
    auto store = [&](int round, ss_v2d v) {
        const int v0 = round * kSsRound + wv * kSsSliceRows + 2 * lane; // view rows v0, v0 + 1
        if ((ABL & 2) && v.x != 123.456) return;
        if (S.rowmap) { // (wave-uniform) mapped rows: two 8-byte stores wherever the map sends them
            const int i = v0 - S.shift;
            if (i >= 0 && v0 < S.n) y[S.rowmap[i]] = v.x;
            if (v0 + 1 < S.n) y[S.rowmap[i + 1]] = v.y;
        } else if (v0 >= S.shift && v0 + 1 < S.n) *reinterpret_cast<ss_v2d*>(y + v0) = v;
        else {
            if (v0 >= S.shift && v0 < S.n) y[v0] = v.x;
            if (v0 + 1 < S.n) y[v0 + 1] = v.y;
        }
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
                        for (int u = 0; u < kSsNewMax / 256; u++) nx[u] = x[min(max(wn.x + tid + 256 * u, cfirst), cowned)];
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
    if ((ABL & 8) && tid == 0) S.trace[4 * g + 2] = __builtin_amdgcn_s_memrealtime();
    emit();
    flush();
    if (ABL & 8) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) S.trace[4 * g + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// block b -> logical workgroup (b % 8) * (G / 8) + b / 8 (the workgroups of one XCD stream neighbouring rows), and its record
__device__ __forceinline__ int ss_logical_wg(const SsView& S, int bid)
{
    const int per = S.nwg >> 3;
    return per > 0 && (S.nwg & 7) == 0 ? (bid & 7) * per + (bid >> 3) : bid;
}

template <int D, bool NT, int ABL = 0>
__global__ __launch_bounds__(256) void spmv_sstream(SsView S, const double* __restrict__ x, double* __restrict__ y)
{
    const int g = ss_logical_wg(S, (int)blockIdx.x);
    const SsWg W = S.wg[g]; // (uniform address: one scalar load)
    if (W.r_begin >= W.r_end) return;
    __shared__ double ring[kSsRing];
    __shared__ ss_v2d s_park[4 * kSsPark * 64];
    ss_body<D, NT, ABL, 0>(S, W, g, x, y, RingComm{}, ring, s_park);
}

template <int D, bool NT, int ABL = 0>
__global__ __launch_bounds__(256) void spmv_sstream_fused(SsView S, const double* __restrict__ x, double* __restrict__ y, RingComm C)
{
    if ((int)blockIdx.x < C.push_wgs) { // fallback (a rank that sends and reads no ghost): dedicated push workgroups in front of the grid
        if ((int)blockIdx.x < C.n_links) ring_push_gate<256>(C);
        for (int l = blockIdx.x; l < C.n_links; l += C.push_wgs) ring_push_link<256>(C, x, l);
        return;
    }
    const int g = ss_logical_wg(S, (int)blockIdx.x - C.push_wgs);
    const SsWg W = S.wg[g];
    if (W.r_begin >= W.r_end) return;
    __shared__ double ring[kSsRing];
    __shared__ ss_v2d s_park[4 * kSsPark * 64];
    if (W.halo) ss_body<D, NT, ABL, 2>(S, W, g, x, y, C, ring, s_park); // (workgroup-uniform: each workgroup runs ONE of the two bodies)
    else ss_body<D, NT, ABL, 1>(S, W, g, x, y, C, ring, s_park);
}

} // namespace mi355
