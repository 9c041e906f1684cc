// mring_plan.hpp — host-side (no HIP) plan of the multi-window ring kernel (spmv_mring.hpp).
//
// The ring kernel (spmv_ring.hpp) keeps ONE sliding window of x in LDS: right for a band, useless for a 3-D mesh, whose
// rows reach into the mesh plane (or Cuthill-McKee level) below, their own, and the one above — three narrow clusters of
// columns, two whole planes apart (P1 pressure operator on 100^3 cells, natural or relabelled order: one window would
// have to span 16-21 k columns, THREE windows 0.7-1.0 k in total).  And the clusters slide: along a plane each moves forward
// with the rows, and where a level ends the next one begins right behind it in the numbering, so the windows just keep
// sliding.  So: K = 5 independent sliding windows, each a ring of RING / K entries of the same LDS array; every nonzero's
// LDS slot is precomputed as a 16-bit number exactly as for the single ring.  The kernel is the ring kernel with a K-way
// refill — same pipeline, same row chains, same bits.
// (Relabelled 100^3-cell mesh, blocks of 2048 nonzeros: 3 clusters in 92 %, 5 where the rows cross from one Cuthill-McKee
// level into the next, in 7 %; widest cluster 265 columns median, 461 at the 99th percentile: K = 5 windows of 960 serve
// 99.7 %, K = 4 of 1280 only 93 %.)
//
// Refill without decoding.  New columns enter a window in GROUPS of 64 consecutive columns (windows start on multiples of 64
// and simply run a little ahead of what a block needs), and the plan spells a block's groups out: up to kMringGroups = 8 per
// block, each as {first column, first LDS slot}.  A wave of the kernel loads one group per prefetch register — one uniform
// LDS read for the address, nothing to decode (decoding per-window ranges in the kernel, per lane or per wave, cost 20-50 %
// of the whole kernel: with two waves per SIMD even scalar instructions are not free).  Groups never straddle a window's
// wrap-around (window size and bases are multiples of 64).  Spare groups of a block are spent on running further ahead, so
// that the three windows of a mesh — 137 new columns each per block, i.e. two or three groups — never all need three at once.
//
// Plan, per run (a workgroup's consecutive blocks), per block:
//   * the block's distinct columns are cut into clusters wherever two neighbours lie kMringGap or more apart;
//   * a cluster continues the window it starts in (or just above: within kMringGap of its upper end), sliding it; clusters
//     that continue nothing take windows no cluster of this block uses (a restart of that window alone);
//   * more than K clusters, or a cluster wider than a window: the block is PLAIN (computed behind the loop, like the single
//     ring's), and all windows start afresh behind it;
//   * a block that needs more than kMringGroups groups starts a new run: a run's first block has its windows filled whole
//     by the kernel's prologue (per-run record `first`), so the steady-state loop has no unpipelined refill (LEAN).
// Pure integer work; checked by replay in mi_mring_plan_probe (every nonzero's slot holds its column when its block runs).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ring_plan.hpp"

namespace mi355 {

constexpr int kMringK = 5;          // windows
constexpr int kMringW = 960;        // entries per window (a multiple of 64)
constexpr int kMringRing = kMringK * kMringW; // doubles of LDS for all windows together: with staging and the plan records
                                              // two workgroups fit a CU's LDS, like the single ring's
constexpr int kMringGap = 256;      // neighbouring distinct columns this far apart belong to different clusters
constexpr int kMringLead = 4;       // groups a window runs ahead of what its cluster needs, where spare groups allow
constexpr int kMringGroups = 8;     // groups of 64 new columns a block inside a run may bring (2 per wave of the kernel)
constexpr int kMringNnzb = 2048, kMringThreads = 256, kMringWgUnit = 512;
constexpr int kMringXcds = 8;       // workgroup i of a grid goes to XCD i % 8 (= kNXCD of the kernels; this header has no HIP in it)
constexpr int kMringMaxB = 96;      // blocks per run (the LDS copy of a run's records)
constexpr int kMringRec = 20;       // ints per block record: {r0, p0, rows, nnz} {flags, groups, plain rows, 0} {gcol[8]} {gslot[8] as 16-bit pairs}
constexpr int kMringFirst = 16;     // ints per run: first block's windows {lo[5]} {count[5]} {offset in its ring [5]} {0}
static_assert(kMringW % 64 == 0, "windows are refilled in groups of 64 columns");

struct MringPlanHost {
    int nblk = 0, wgs = 0, nruns = 0, bpw = 0, bad_runs = 0; // wgs: length of the per-run tables = the kernel's grid (8 XCD shares, the shorter ones padded with empty entries); nruns of them hold a run
    long long bad_nnz = 0;
    std::vector<int> plan;   // kMringRec ints per block; flags 1 = window-served, 2 = PLAIN {r0, p0, 0, nnz} {2, 0, rows, 0}, 0 = empty
    std::vector<int> first;  // kMringFirst ints per run
    std::vector<int> run_ok; // per run: 0 plain path, 1 loop, 3 loop + PLAIN blocks behind it
    std::vector<int> run_rng;
    std::vector<unsigned short> slots; // per block kMringNnzb entries in thread order (as build_ring_slots)
    long long restarts = 0;            // runs started because a block needed more groups than the loop refills (diagnostic)
};

inline void build_mring_plan(int n, const int* ptrow, const int* indcol, MringPlanHost& out, int row_align_arg = 0, int skew_pct = 0)
{
    out = MringPlanHost();
    const int K = kMringK, W = kMringW, T = kMringThreads, nnzb = kMringNnzb, per = nnzb / T, G = kMringGroups;
    std::vector<int> rows, ptrs;
    // <= T rows per block (the kernel makes one pass over a block's rows), whole waves of rows where that keeps 7/8 of the block
    // (ring_plan.hpp: the row-chain phase is the LDS-bound part)
    int row_align = row_align_arg > 0 ? row_align_arg : 64; // (large matrices: mi_csr_create times both shapes, ring_plan.hpp)
    if (const char* e = getenv("MI355_RING_ROW_ALIGN")) row_align = std::max(1, atoi(e)); // (A/B: tools/rowalign_ab.py)
    build_row_blocks(n, ptrow, nnzb, T, rows, ptrs, row_align, 7);
    const int nblk = (int)rows.size() - 1;
    out.nblk = nblk;
    if (nblk <= 0) return;
    // clusters of every block: distinct columns sorted, cut at gaps >= kMringGap
    struct Cl { int lo, hi; }; // [lo, hi]
    std::vector<int> cl_ptr((size_t)nblk + 1, 0);
    std::vector<Cl> cls;
    std::vector<char> wide((size_t)nblk, 0);
    {
        std::vector<int> u;
        for (int b = 0; b < nblk; b++) {
            const int p0 = ptrs[b], nn = ptrs[b + 1] - p0;
            cl_ptr[b] = (int)cls.size();
            if (nn <= 0) continue;
            if (nn > nnzb) { wide[b] = 1; continue; }
            u.assign(indcol + p0, indcol + p0 + nn);
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            const size_t first = cls.size();
            Cl cur{u[0], u[0]};
            for (size_t i = 1; i < u.size(); i++) {
                if (u[i] - u[i - 1] >= kMringGap) { cls.push_back(cur); cur.lo = u[i]; }
                cur.hi = u[i];
            }
            cls.push_back(cur);
            bool bad = (int)(cls.size() - first) > K;
            for (size_t i = first; i < cls.size() && !bad; i++) bad = cls[i].hi - cls[i].lo + 1 > W - 64; // (room to start on a multiple of 64)
            if (bad) { cls.resize(first); wide[b] = 1; }
        }
        cl_ptr[nblk] = (int)cls.size();
    }
    long long weight = 0;
    for (int b = 0; b < nblk; b++) weight += wide[b] ? kRingPlainWeight : 1;
    out.plan.assign((size_t)kMringRec * nblk, 0);
    out.slots.assign((size_t)nblk * nnzb, 0);

    // window w holds columns [lo, hi), both a multiple of 64 away from base; slot of column c = w * W + (c - base) mod W
    struct Win { int lo[kMringK], hi[kMringK], base[kMringK]; bool live[kMringK]; };
    auto reset = [&](Win& S) { for (int w = 0; w < K; w++) { S.lo[w] = S.hi[w] = S.base[w] = 0; S.live[w] = false; } };
    auto slot_of = [&](const Win& S, int w, int c) {
        int s = c - S.base[w];
        if (s >= W) s -= W;
        return w * W + s;
    };
    struct Step { int nlo[kMringK], from[kMringK], groups[kMringK]; bool restart[kMringK]; int win_of[kMringK], total; };
    // what block b needs of window state S (S is not changed): per window the groups to load, the new lower end
    auto need = [&](int b, const Win& S, Step& R) {
        const Cl* C = &cls[cl_ptr[b]];
        const int nc = cl_ptr[b + 1] - cl_ptr[b];
        bool used[kMringK] = {};
        for (int w = 0; w < K; w++) { R.groups[w] = 0; R.restart[w] = false; R.nlo[w] = S.lo[w]; R.from[w] = S.hi[w]; }
        for (int j = 0; j < K; j++) R.win_of[j] = -1;
        for (int j = 0; j < nc; j++) // which window does each cluster continue?
            for (int w = 0; w < K; w++)
                if (S.live[w] && !used[w] && C[j].lo >= S.lo[w] && C[j].lo < S.hi[w] + kMringGap) { R.win_of[j] = w; used[w] = true; break; }
        for (int j = 0; j < nc; j++) { // the others: a window nobody uses in this block (prefer one that holds nothing)
            if (R.win_of[j] >= 0) continue;
            int pick = -1;
            for (int w = 0; w < K && pick < 0; w++)
                if (!used[w] && !S.live[w]) pick = w;
            for (int w = 0; w < K && pick < 0; w++)
                if (!used[w]) pick = w;
            R.win_of[j] = pick; // nc <= K: there is one
            used[pick] = true;
            R.restart[pick] = true;
        }
        R.total = 0;
        for (int j = 0; j < nc; j++) {
            const int w = R.win_of[j], cmin = C[j].lo, cmax = C[j].hi;
            const int lo = S.lo[w], hi = S.hi[w];
            if (!R.restart[w]) {
                const int nhi = hi + ((std::max(hi, cmax + 1) - hi + 63) & ~63);
                if (cmin < std::max(lo, nhi - W)) R.restart[w] = true; // cannot keep the upper end and reach down to cmin
                else { R.from[w] = hi; R.groups[w] = (nhi - hi) / 64; R.nlo[w] = std::max(lo, nhi - W); }
            }
            if (R.restart[w]) {
                const int l0 = cmin & ~63; // a multiple of 64 at or below cmin; the cluster is at most W - 64 wide: it fits
                R.from[w] = l0;
                R.groups[w] = (cmax + 1 - l0 + 63) / 64; // <= W / 64
                R.nlo[w] = l0;
            }
            R.total += R.groups[w];
        }
    };
    // commit a step: advance S, spend `spare` further groups on running ahead (the window with the least lead first)
    auto commit = [&](int b, Win& S, Step& R, int spare) {
        const Cl* C = &cls[cl_ptr[b]];
        const int nc = cl_ptr[b + 1] - cl_ptr[b];
        for (int j = 0; j < nc; j++) {
            const int w = R.win_of[j];
            if (R.restart[w]) S.base[w] = (R.from[w] / W) * W;
            S.lo[w] = R.nlo[w];
            S.hi[w] = R.from[w] + 64 * R.groups[w];
            S.live[w] = true;
        }
        while (spare > 0) {
            int pick = -1, lead = 0x7fffffff;
            for (int j = 0; j < nc; j++) {
                const int w = R.win_of[j];
                if (S.hi[w] + 64 - C[j].lo > W || S.hi[w] >= n) continue; // must keep the cluster's lower end; nothing left to run ahead into
                const int l = S.hi[w] - (C[j].hi + 1);
                if (l < lead) { lead = l; pick = w; }
            }
            if (pick < 0 || lead >= kMringLead * 64) break; // far enough ahead everywhere
            R.groups[pick]++;
            S.hi[pick] += 64;
            S.lo[pick] = std::max(S.lo[pick], S.hi[pick] - W);
            spare--;
        }
        for (int w = 0; w < K; w++)
            while (S.live[w] && S.lo[w] - S.base[w] >= W) S.base[w] += W;
    };

    std::vector<int> cuts, first, forced_at;
    int forced = 0;
    // One pass over the blocks: runs are cut where their weight would pass `target` — or, when `planned` is given, exactly at the
    // listed blocks — and, on top, wherever a block needs more groups than the loop refills (a forced cut; where these fall depends
    // a little on where the run began, since that decides what the windows hold).  emit = false: cuts only.
    auto pass = [&](long long target, const std::vector<int>* planned, bool emit) {
        cuts.assign(1, 0);
        first.assign(kMringFirst, 0);
        forced_at.clear();
        forced = 0;
        if (emit) std::fill(out.plan.begin(), out.plan.end(), 0);
    Win S;
    reset(S);
    long long cum = 0;
    int count = 0;
    size_t pi = 0;
    for (int b = 0; b < nblk; b++) {
        const int nn = ptrs[b + 1] - ptrs[b], nrows = rows[b + 1] - rows[b], wb = wide[b] ? kRingPlainWeight : 1;
        auto fresh = [&]() {
            if (b > cuts.back()) { cuts.push_back(b); first.resize(first.size() + kMringFirst, 0); }
            cum = 0; count = 0;
            reset(S);
        };
        while (planned && pi < planned->size() && (*planned)[pi] < b) pi++;
        const bool cut_here = planned ? (pi < planned->size() && (*planned)[pi] == b) : cum + wb > target;
        if (count > 0 && (count >= kMringMaxB || cut_here)) fresh();
        if (!emit) { // cuts only: the windows' state decides where a block forces a new run, nothing is written
            if (nn > 0 && wide[b]) reset(S);
            else if (nn > 0) {
                Step R;
                need(b, S, R);
                if (count > 0 && R.total > G) { fresh(); forced++; forced_at.push_back(b); need(b, S, R); }
                commit(b, S, R, count == 0 ? kMringLead * K : G - R.total);
            }
            cum += wb;
            count++;
            continue;
        }
        int* P = &out.plan[(size_t)kMringRec * b];
        P[0] = rows[b]; P[1] = ptrs[b]; P[2] = nrows; P[3] = nn;
        for (int g = 0; g < G; g++) P[8 + g] = std::min(std::max(0, n - 64), rows[b]); // unused groups: a harmless load near the rows
        P[16] = P[17] = P[18] = P[19] = -1;                                            // slot 0xFFFF: no group
        if (nn > 0 && wide[b]) {
            P[2] = 0; P[4] = 2; P[6] = nrows;
            reset(S);
        } else if (nn > 0) {
            Step R;
            need(b, S, R);
            if (count > 0 && R.total > G) { fresh(); forced++; forced_at.push_back(b); need(b, S, R); } // this block starts a run: its windows are filled whole
            const bool first_now = count == 0;
            commit(b, S, R, first_now ? kMringLead * K : G - R.total); // a run's first windows come in with the lead the loop keeps (the prologue loads whole windows anyway)
            P[4] = 1;
            if (first_now) {
                int* F = &first[first.size() - kMringFirst];
                for (int w = 0; w < K; w++) {
                    F[w] = R.groups[w] ? R.from[w] : 0;
                    F[5 + w] = 64 * R.groups[w];
                    F[10 + w] = R.groups[w] ? (R.from[w] - S.base[w]) % W : 0;
                }
                P[5] = 0;
            } else {
                int g = 0;
                for (int w = 0; w < K; w++)
                    for (int q = 0; q < R.groups[w]; q++, g++) {
                        const int c0 = R.from[w] + 64 * q;
                        P[8 + g] = c0;
                        const unsigned sl = (unsigned)slot_of(S, w, c0);
                        unsigned pk = (unsigned)P[16 + g / 2];
                        pk = (g & 1) ? ((pk & 0x0000ffffu) | (sl << 16)) : ((pk & 0xffff0000u) | sl);
                        P[16 + g / 2] = (int)pk;
                    }
                P[5] = g;
            }
            // slots of this block's nonzeros
            const Cl* C = &cls[cl_ptr[b]];
            const int nc = cl_ptr[b + 1] - cl_ptr[b], p0 = ptrs[b];
            unsigned short* o = &out.slots[(size_t)b * nnzb];
            const bool pair = ring_pairs(T); // (ring_pair.hpp: the thread owns nonzero pairs, 16-byte value loads)
            for (int t = 0; t < T; t++)
                for (int i = 0; i < per; i++) {
                    const int k = std::min(pair ? 2 * (t + (i >> 1) * T) + (i & 1) : t + i * T, nn - 1);
                    const int c = indcol[p0 + k];
                    int j = 0;
                    while (j + 1 < nc && c > C[j].hi) j++;
                    o[t * per + i] = (unsigned short)slot_of(S, R.win_of[j], c);
                }
        }
        cum += wb;
        count++;
    }
        return (int)cuts.size();
    };
    // Where to cut, and who runs what.  How a grid of these workgroups is dispatched (tools/mring_timeline.py, the kernel's TRACE
    // instantiation): workgroup i goes to XCD i % 8; an XCD takes its workgroups strictly in order, each bound to one of its four
    // shader engines in turn, two per CU (LDS) — so a workgroup beyond the 64 resident ones starts only when a slot of ITS engine
    // frees, and everything behind it waits with it: short runs among the first 64 free their slots for nobody until the long
    // runs end (seen: late-comers starting at 145 us of 165, whatever had finished at 5).  The two workgroups of a CU do not share it
    // evenly either: one finishes after ~90 % of the launch, and the late-comers then run beside the other.  Hence:
    //   * a run is LONG or SHORT (at most kMringShort blocks' weight); at most 64 long runs per XCD, all in its first round, in
    //     matrix order (neighbouring runs share the XCD's L2); the short ones behind them — what is left of the first round, then
    //     the second, where a short run ends before the launch's last long run does;
    //   * forced cuts (a block that needs more groups than the loop refills: ~90 on a relabelled 5 M-row mesh) split the matrix into
    //     segments; a segment is cut into equal pieces of at most `target`, so that no remainder falls between short and long, and
    //     `target` is the smallest for which the long runs fit the first round (91-block runs beside stubs -> 78-block runs).
    const int slots_per_xcd = kMringWgUnit / kMringXcds;
    const long long ideal = std::max<long long>(1, (weight + kMringWgUnit - 1) / kMringWgUnit);
    const long long kMringShort = std::min<long long>(5, ideal / 8); // (a second-round run has the launch's last tenth to itself)
    std::vector<long long> cumw((size_t)nblk + 1, 0);
    for (int b = 0; b < nblk; b++) cumw[b + 1] = cumw[b] + (wide[b] ? kRingPlainWeight : 1);
    auto run_weight = [&](int r) { return cumw[r + 1 < (int)cuts.size() ? cuts[r + 1] : nblk] - cumw[cuts[r]]; };
    auto count_long = [&]() {
        int nl = 0;
        for (int r = 0; r < (int)cuts.size(); r++) nl += run_weight(r) > kMringShort;
        return nl;
    };
    const char* e1 = getenv("MI355_MRING_DEAL"); // A/B: 0 = runs by weight only, no more runs than resident workgroups (the round-2 rule)
    const bool by_weight = e1 && atoi(e1) == 0;
    std::vector<int> planned;
    if (by_weight || ideal >= kMringMaxB) {
        long long target = ideal;
        pass(target, nullptr, false);
        while (by_weight && (int)cuts.size() > kMringWgUnit && target < kMringMaxB) pass(++target, nullptr, false);
        pass(target, nullptr, true);
    } else {
        pass(ideal, nullptr, false);
        std::vector<int> seg(forced_at); // segment boundaries: where blocks force a cut whatever the weight
        seg.insert(seg.begin(), 0);
        seg.push_back(nblk);
        for (long long target = ideal;; target++) {
            planned.clear();
            for (size_t i = 0; i + 1 < seg.size(); i++) {
                const int b0 = seg[i], b1 = seg[i + 1];
                const long long w = cumw[b1] - cumw[b0];
                if (i > 0) planned.push_back(b0);
                if (w <= kMringShort) continue;
                const long long pieces = (w + target - 1) / target;
                int b = b0;
                for (long long q = 1; q < pieces; q++) { // piece q ends where the segment's weight passes q / pieces of the whole
                    // (skew_pct: pieces alternately that many per cent longer and shorter; the longer go to the older workgroup of a CU)
                    const long long num = 100 * q + ((q & 1) ? skew_pct : 0);
                    while (b < b1 && (cumw[b] - cumw[b0]) * pieces * 100 < w * num) b++;
                    if (b > b0 && b < b1 && (planned.empty() || planned.back() < b)) planned.push_back(b);
                }
            }
            pass(0, &planned, false);
            if (count_long() <= kMringWgUnit || target >= kMringMaxB) break;
        }
        pass(0, &planned, true);
    }
    const int nruns = (int)cuts.size();
    // dispatch order: per XCD its long runs, then its short ones
    std::vector<std::vector<int>> share(kMringXcds);
    {
        const int nl = count_long();
        std::vector<std::vector<int>> shorts(kMringXcds);
        int il = 0, x = 0;
        for (int r = 0; r < nruns; r++) {
            if (run_weight(r) > kMringShort || by_weight) {
                x = nl <= kMringWgUnit && !by_weight ? (int)((long long)il * kMringXcds / std::max(nl, 1)) : (int)((long long)r * kMringXcds / nruns);
                share[x].push_back(r);
                il++;
            } else shorts[x].push_back(r);
        }
        for (int q = 0; q < kMringXcds; q++) {
            // workgroups j and j + 32 of an XCD share a CU, which is done when both are: its longest run beside its shortest, and so on
            std::vector<int>& L = share[q];
            const int m = (int)L.size(), half = slots_per_xcd / 2;
            if (!by_weight && m > half && m <= slots_per_xcd) {
                std::vector<int> byw(L);
                std::stable_sort(byw.begin(), byw.end(), [&](int a, int b) { return run_weight(a) > run_weight(b); });
                for (int r = 0; r < m; r++) L[r < half ? r : half + (m - 1 - r)] = byw[r];
            }
            L.insert(L.end(), shorts[q].begin(), shorts[q].end());
        }
    }
    int per_xcd = 1;
    for (int q = 0; q < kMringXcds; q++) per_xcd = std::max(per_xcd, (int)share[q].size());
    const int ntab = kMringXcds * per_xcd;
    out.wgs = ntab;
    out.nruns = nruns;
    out.bpw = (nblk + nruns - 1) / nruns;
    out.restarts = forced;
    out.run_ok.assign(ntab, 1);
    out.run_rng.assign((size_t)2 * ntab, 0);
    out.first.assign((size_t)kMringFirst * ntab, 0);
    for (int g = 0; g < ntab; g++) { // table entry g = what workgroup g % per_xcd of XCD g / per_xcd runs (nothing: an empty range)
        const int q = g / per_xcd, j = g % per_xcd, r = j < (int)share[q].size() ? share[q][j] : -1;
        out.run_rng[2 * g] = r >= 0 ? cuts[r] : nblk;
        out.run_rng[2 * g + 1] = r >= 0 ? (r + 1 < nruns ? cuts[r + 1] : nblk) : nblk;
        if (r >= 0) std::copy(first.begin() + (size_t)kMringFirst * r, first.begin() + (size_t)kMringFirst * (r + 1), out.first.begin() + (size_t)kMringFirst * g);
    }
    for (int g = 0; g < ntab; g++) { // runs with too many PLAIN blocks go down the plain path as a whole
        int nplain = 0;
        long long run_nnz = 0, plain_nnz = 0;
        for (int b = out.run_rng[2 * g]; b < out.run_rng[2 * g + 1]; b++) {
            const int* P = &out.plan[(size_t)kMringRec * b];
            run_nnz += P[3];
            if (P[4] == 2) { nplain++; plain_nnz += P[3]; }
        }
        if (nplain > kRingMaxPlain) {
            out.run_ok[g] = 0;
            out.bad_runs++;
            out.bad_nnz += run_nnz;
            for (int b = out.run_rng[2 * g]; b < out.run_rng[2 * g + 1]; b++) {
                int* P = &out.plan[(size_t)kMringRec * b];
                if (P[4] == 2) { P[2] = P[6]; P[6] = 0; }
                P[4] = 0;
            }
        } else {
            out.bad_nnz += plain_nnz;
            if (nplain > 0) out.run_ok[g] = 3; // tells the kernel to look for PLAIN blocks behind its loop
        }
    }
}

// Replay of a plan against its matrix, as the kernel will execute it: returns nullptr or the first violation.
inline const char* check_mring_plan(const MringPlanHost& P, int n, const int* ptrow, const int* indcol)
{
    const int K = kMringK, W = kMringW, T = kMringThreads, nnzb = kMringNnzb, per = nnzb / T, G = kMringGroups;
    if (P.nblk == 0) return n == 0 ? nullptr : "no blocks for a matrix with rows";
    std::vector<int> run_of((size_t)P.nblk, -1);
    for (int g = 0; g < P.wgs; g++) {
        const int b0 = P.run_rng[2 * g], b1 = P.run_rng[2 * g + 1];
        if (b0 < 0 || b1 < b0 || b1 > P.nblk || b1 - b0 > kMringMaxB) return "run range out of bounds or longer than the kernel's plan";
        for (int b = b0; b < b1; b++) {
            if (run_of[b] >= 0) return "a block belongs to two runs";
            run_of[b] = g;
        }
    }
    std::vector<int> content((size_t)K * W, -1);
    int next_row = 0, cur_run = -1;
    long long next_nz = 0;
    for (int b = 0; b < P.nblk; b++) {
        if (run_of[b] < 0) return "a block belongs to no run";
        const int* Q = &P.plan[(size_t)kMringRec * b];
        const int brows = Q[4] == 2 ? Q[6] : Q[2];
        if (Q[0] != next_row || Q[1] != next_nz) return "plan does not cover rows / nonzeros in order";
        next_row += brows;
        next_nz += Q[3];
        if (Q[3] != ptrow[Q[0] + brows] - ptrow[Q[0]]) return "block nonzero count disagrees with ptrow";
        if (brows > T) return "a block of more than T rows";
        const int run = run_of[b];
        if (run != cur_run) { // a new workgroup: its prologue fills the first block's windows whole from the run's record
            std::fill(content.begin(), content.end(), -1);
            cur_run = run;
            const int* F = &P.first[(size_t)kMringFirst * run];
            for (int w = 0; w < K; w++) {
                if (F[5 + w] % 64 != 0 || F[5 + w] < 0 || F[5 + w] > W || F[10 + w] % 64 != 0 || F[10 + w] < 0 || F[10 + w] >= W) return "first-block record out of range";
                for (int i = 0; i < F[5 + w]; i++) content[(size_t)w * W + (F[10 + w] + i) % W] = F[w] + i;
            }
        }
        for (int g = 0; g < G; g++)
            if (Q[8 + g] < 0 || Q[8 + g] > std::max(n, 64)) return "a group's first column is out of range";
        if (!P.run_ok[run]) {
            if (Q[4] != 0) return "flags of a block in a plain run";
            continue;
        }
        if (Q[3] == 0) continue;
        if (Q[4] == 2) {
            if (Q[2] != 0 || Q[5] != 0) return "a PLAIN block is visible to the loop";
            if (P.run_ok[run] != 3) return "a run with a PLAIN block does not tell the kernel to look behind its loop";
            continue;
        }
        if (Q[4] != 1 || Q[3] > nnzb) return "a served run holds a block the kernel cannot take";
        if (Q[5] < 0 || Q[5] > G) return "more groups than the loop refills";
        if (Q[5] > 0 && b == P.run_rng[2 * run]) return "a run's first block must come in through the run's record";
        for (int g = 0; g < G; g++) {
            const unsigned sl = ((unsigned)Q[16 + g / 2] >> (16 * (g & 1))) & 0xffffu;
            if (g >= Q[5]) {
                if (sl != 0xffffu) return "an unused group has a slot";
                continue;
            }
            if (sl % 64 != 0 || sl >= (unsigned)(K * W)) return "a group's slot is not a multiple of 64 inside the LDS array";
            if ((sl % W) + 64 > (unsigned)W) return "a group straddles its window's wrap-around";
            for (int i = 0; i < 64; i++) content[sl + i] = Q[8 + g] + i;
        }
        for (int k = 0; k < Q[3]; k++) {
            const int slot = P.slots[(size_t)b * nnzb + (size_t)ring_slot_pos(T, per, k)];
            if (slot < 0 || slot >= K * W) return "slot outside the LDS array";
            if (content[slot] != indcol[Q[1] + k]) return "a nonzero's slot does not hold its column when its block runs";
        }
    }
    if (next_row != n || next_nz != ptrow[n]) return "plan does not cover the matrix";
    return nullptr;
}

} // namespace mi355
