// ring_plan.hpp — host-side (no HIP) construction of the per-block window plan that
// the ring kernel (spmv_ring.hpp) replays.  Pure integer work, done once per
// matrix at mi_csr_create; testable on a CPU-only machine.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "partition.hpp"
#include "ring_pair.hpp"

namespace mi355 {

struct RingConfig {
    int id;      // 1, 2, 3 (see kRingConfigs)
    int threads; // T
    int nnzb;    // nonzeros per row block
    int ring;    // doubles of x window in LDS
    int depth;   // D, blocks of prefetch
    int wg_unit; // workgroups per "wave" of the launch: 256 CUs x resident workgroups per CU
};

constexpr int kRingMaxB = 160; // blocks per run (LDS plan capacity)
// A block the window cannot serve (one row reaching over more columns than the ring holds, or longer than a block) does not
// disqualify its run: up to this many per run are flagged PLAIN — the pipelined loop passes over them as over an empty block
// and the workgroup computes them after its loop with direct gathers (spmv_ring.hpp).  More than that and the whole run takes
// the plain path, as before.
constexpr int kRingMaxPlain = 3;
// All runs of the persistent grid start together, so the launch lasts as long as its slowest run: blocks the window cannot
// serve count this many served blocks when the rows are dealt out to the runs (measured cost of the plain path per block)
constexpr int kRingPlainWeight = 4;
// The ring kernel addresses the value stream, ptrow and rowmap with plain tid-strided indices: lanes
// past a block's last nonzero / last row read what lies behind (never used) instead of clamping
// every index.  The device copies therefore carry this much initialised padding at the end:
constexpr int kRingPadNnz = 4096;      // >= the largest NNZB of kRingConfigs
constexpr int kRingPadRows = 1024 + 1; // >= the largest 2*T, plus the row-end entry

// 1: two workgroups per CU (74 KB LDS each); 2: one per CU, bigger blocks (108 KB);
// 3: one per CU with the widest window that still fits 160 KB (for wider bands).
// 4: like 1 with 256 threads (8 nonzeros per thread): two workgroups of 4 waves per CU.  With the
//    16-bit column stream and non-temporal value loads this is the fastest shape measured on C4
//    (tools/kbench "C16S" rows: 146-151 us against 152-159 for 1, 168-184 for 2), so it is tried first.
//    (In-process A/B of the final kernel, tools/cfg_ab.py, C4: 4 = 151.4 us, 1 = 158.1, 2 = 160.0; the same
//    shape with D = 3 or with a 4352-entry ring: 152.4 / 151.3 — no gain, not kept.  C2: all within 2 %.)
constexpr int kNumRingConfigs = 4;
static const RingConfig kRingConfigs[kNumRingConfigs] = {
    {1, 512, 2048, 5120, 2, 512},
    {2, 512, 4096, 5120, 2, 256},
    {3, 512, 4096, 11264, 2, 256},
    {4, 256, 2048, 5120, 2, 512},
};
static const int kRingConfigOrder[kNumRingConfigs] = {4, 1, 2, 3};

// position, inside a block's nnzb entries of the column stream, of the block's k-th nonzero (thread-major: PER entries per thread)
inline int ring_slot_pos(int T, int per, int k)
{
    if (!ring_pairs(T)) return (k % T) * per + k / T;
    const int m = k >> 1; // pair index t + i*T
    return (m % T) * per + 2 * (m / T) + (k & 1);
}

struct RingPlanHost {
    RingConfig cfg{};
    int nblk = 0, wgs = 0, bpw = 0, bad_runs = 0;
    long long bad_nnz = 0;     // nonzeros living in runs that take the plain path
    std::vector<int> plan;     // 8 ints per block: {r0, p0, rows, nnz, new_lo, new_cnt, base, flags}; flags 1 = window-served,
                               // 2 = PLAIN: {r0, p0, 0, nnz, rows, 0, base, 2} (rows moved out of the loop's sight), 0 = empty
    std::vector<int> run_ok;   // per run: 0 plain path, 1 ring loop, 3 ring loop + PLAIN blocks behind it
    std::vector<int> run_rng;  // per run: {first block, end block} — a run is a contiguous block range, runs need not be in order
    std::vector<int> run_halo; // per run: touches a ghost column (fused multi-GPU step only)
    bool lean = true;          // no served block but a run's first brings > T new columns, no block holds > T rows (spmv_ring.hpp: LEAN)
};

// The fused multi-GPU step (spmv_ring.hpp, FUSED): columns outside [ghost_lo, ghost_hi) are ghosts that arrive from the
// neighbours a few microseconds into the kernel, and a run that touches one waits for them first.  All runs start together
// (persistent grid), so a waiting run must be SHORTER than the others by the length of that wait or it becomes the tail of
// the launch: the blocks touching ghosts — a prefix and a suffix of the rows for a banded partition — are dealt out in runs
// of bpw - kGhostRunSlack blocks, the rest in runs of bpw as usual.
constexpr int kGhostRunSlack = 2; // blocks (~2.5 us each at 2048 nonzeros) — the push of push_exchange.hpp lands within ~5 us

// Row blocks for the ring: as build_row_blocks (whole rows, <= nnzb nonzeros, <= max_rows rows), and a block also ends
// where one more row would stretch its column span beyond the ring.  A relabelled band (reorder.hpp) has a few places
// where consecutive rows reach far apart; cut there, the two halves each fit a window, where the uncut block would not
// and would send its whole run down the plain path — with all runs of the persistent grid starting together that one run
// is then the tail of the launch (measured on the relabelled 1 M-row S15 matrix: 39 such blocks of 7353, 81-88 us against
// 34 us for the natural order).
inline void build_ring_blocks(int n, const int* ptrow, const int* row_min, const int* row_max, int nnzb, int max_rows, int ring,
                              std::vector<int>& out_rows, std::vector<int>& out_ptr, int row_align = 1)
{
    out_rows.clear();
    out_ptr.clear();
    int r = 0;
    while (r < n) {
        const int start = r, p0 = ptrow[r];
        int cmin = 0x7fffffff, cmax = -1;
        if (row_min[r] <= row_max[r]) { cmin = row_min[r]; cmax = row_max[r]; }
        int e = r + 1; // a block always takes at least one row
        while (e < n && (e - start) < max_rows && (long long)ptrow[e + 1] - p0 <= nnzb) {
            if (row_min[e] <= row_max[e]) {
                const int nmin = std::min(cmin, row_min[e]), nmax = std::max(cmax, row_max[e]);
                if (cmax >= 0 && nmax - nmin + 1 > ring) break;
                cmin = nmin;
                cmax = nmax;
            }
            e++;
        }
        if (row_align > 1 && e < n) { // whole waves of rows (see build_ring_plan)
            const int ea = (e / row_align) * row_align;
            if (ea > start && 8 * (long long)(ptrow[ea] - p0) >= 7 * (long long)(ptrow[e] - p0)) e = ea;
        }
        out_rows.push_back(start);
        out_ptr.push_back(p0);
        r = e;
    }
    out_rows.push_back(n);
    out_ptr.push_back(n > 0 ? ptrow[n] : 0);
}

// row_min/row_max: smallest / largest column of each row (row_min > row_max for an empty row)
// ghost_lo < ghost_hi: shape the runs for the fused step as described above (and fill run_halo)
// row_align: blocks end on multiples of that many rows (1: no alignment; 0: the default — 64 for configuration 4, see below)
inline void build_ring_plan(const RingConfig& cfg, int n, const int* ptrow, const int* row_min, const int* row_max,
                            RingPlanHost& out, int ghost_lo = 0, int ghost_hi = 0, int row_align_arg = 0)
{
    out = RingPlanHost();
    out.cfg = cfg;
    std::vector<int> rows, ptrs;
    // (configuration 4 keeps its blocks to T rows: one pass over the rows per block is a condition of the LEAN kernel)
    const int max_rows = cfg.id == 4 ? cfg.threads : 2 * cfg.threads;
    // Row-chain phase: one thread per row, so a block of 137 rows keeps THREE waves busy for 2.14 waves of work, and the kernel's
    // busiest unit is the LDS.  Blocks are therefore ended on multiples of 64 rows where that gives up less than an eighth of the
    // block's nonzeros (S15: 128 rows / 1920 nonzeros instead of 136 / 2040): 2-5 % faster at 1 M rows on every box tried; at
    // 5 M rows 150 -> 140 us on one box and 151 -> 158 us on another (in-process A/B, profiles/r02_ring_rowalign_ab.txt) — so for
    // large matrices mi_csr_create builds both plans and keeps the one that measures faster on the box at hand.
    // MI355_RING_ROW_ALIGN=1|64 forces one.
    int row_align = row_align_arg > 0 ? row_align_arg : (cfg.id == 4 ? 64 : 1);
    if (const char* e = getenv("MI355_RING_ROW_ALIGN")) row_align = std::max(1, atoi(e));
    build_ring_blocks(n, ptrow, row_min, row_max, cfg.nnzb, max_rows, 0x7fffffff, rows, ptrs, row_align);
    {   // span-limited blocks where that takes only a few cuts; a matrix whose rows themselves outspan the ring would
        // fall apart into one-row blocks — it is not the ring's to serve, keep the plain blocks (its runs go plain)
        std::vector<int> rows2, ptrs2;
        build_ring_blocks(n, ptrow, row_min, row_max, cfg.nnzb, max_rows, cfg.ring, rows2, ptrs2, row_align);
        if (rows2.size() <= rows.size() + rows.size() / 8 + 16) {
            rows.swap(rows2);
            ptrs.swap(ptrs2);
        }
    }
    const int nblk = (int)rows.size() - 1;
    out.nblk = nblk;
    if (nblk <= 0) return;
    // column range of every block; WIDE blocks (a row longer than a block, or columns further apart than the ring holds)
    // cannot be window-served whatever the window did before — everything else can (the window restarts where it must)
    std::vector<int> bmin((size_t)nblk, 0x7fffffff), bmax((size_t)nblk, -1);
    std::vector<char> wide((size_t)nblk, 0);
    long long weight = 0;
    int nwide = 0;
    for (int b = 0; b < nblk; b++) {
        for (int r = rows[b]; r < rows[b + 1]; r++)
            if (row_min[r] <= row_max[r]) { bmin[b] = std::min(bmin[b], row_min[r]); bmax[b] = std::max(bmax[b], row_max[r]); }
        const int nn = ptrs[b + 1] - ptrs[b];
        wide[b] = nn > 0 && (nn > cfg.nnzb || bmax[b] - bmin[b] + 1 > cfg.ring);
        nwide += wide[b];
        weight += wide[b] ? kRingPlainWeight : 1;
    }
    const bool weighted = nwide > 0 && !(ghost_lo < ghost_hi);
    const long long per_unit = (long long)(weighted ? kRingMaxB - kRingPlainWeight : kRingMaxB) * cfg.wg_unit;
    int wgs = cfg.wg_unit * (int)(((weighted ? weight : (long long)nblk) + per_unit - 1) / per_unit);
    if (wgs < cfg.wg_unit) wgs = cfg.wg_unit;
    const int bpw = (nblk + wgs - 1) / wgs;
    out.wgs = wgs;
    out.bpw = bpw;
    out.plan.assign((size_t)8 * nblk, 0);
    out.run_ok.assign(wgs, 1);
    out.run_rng.assign((size_t)2 * wgs, 0);
    out.run_halo.assign(wgs, 0);
    for (int g = 0; g < wgs; g++) { // default: consecutive runs of bpw blocks
        out.run_rng[2 * g] = std::min(nblk, g * bpw);
        out.run_rng[2 * g + 1] = std::min(nblk, (g + 1) * bpw);
    }
    if (weighted) { // runs of equal WEIGHT: run g ends with the block that carries the cumulated weight past (g + 1) / wgs of the total
        long long cum = 0;
        int g = 0, start = 0;
        for (int b = 0; b < nblk; b++) {
            cum += wide[b] ? kRingPlainWeight : 1;
            while (g < wgs - 1 && cum * wgs >= (long long)(g + 1) * weight) {
                out.run_rng[2 * g] = start;
                out.run_rng[2 * g + 1] = b + 1;
                start = b + 1;
                g++;
            }
        }
        out.run_rng[2 * g] = start;
        out.run_rng[2 * g + 1] = nblk;
        for (g++; g < wgs; g++) out.run_rng[2 * g] = out.run_rng[2 * g + 1] = nblk;
    }
    // LEAN runs (spmv_ring.hpp): inside a run no block may bring more than T new columns — only a run's FIRST block may (the
    // kernel fills its whole window up front).  So runs are cut where the window would restart or jump (a relabelled band has a
    // few dozen such places), the other cuts are placed by weight as before.  Replays the window exactly as the plan loop
    // below; gives up (keeps the runs above: general kernel) if that needs more runs than workgroups.
    if (!(ghost_lo < ghost_hi) && cfg.id == 4) {
        const int ring = cfg.ring, T = cfg.threads;
        std::vector<int> cuts; // first block of every run
        auto simulate = [&](long long target) {
            cuts.clear();
            cuts.push_back(0);
            long long cum = 0;
            int count = 0, wlo = 0, whi = 0;
            bool live = false;
            for (int b = 0; b < nblk; b++) {
                const int nn = ptrs[b + 1] - ptrs[b], wb = wide[b] ? kRingPlainWeight : 1;
                auto fresh = [&]() { if (b > cuts.back()) cuts.push_back(b); cum = 0; count = 0; live = false; };
                if (count >= kRingMaxB || (count > 0 && cum + wb > target)) fresh();
                if (nn > 0 && !wide[b]) {
                    const int cmin = bmin[b], cmax = bmax[b];
                    for (int pass = 0; pass < 2; pass++) {
                        int lo = live ? wlo : cmin, hi = live ? whi : cmin;
                        bool restart = !live || cmin < lo || cmin > hi;
                        if (restart) { lo = std::max(0, std::min(cmin, cmax + 1 - ring)); hi = lo; }
                        int nhi = std::max(hi, cmax + 1), nlo = std::max(lo, nhi - ring);
                        if (cmin < nlo) { lo = std::max(0, std::min(cmin, cmax + 1 - ring)); hi = lo; nhi = cmax + 1; nlo = std::max(lo, nhi - ring); }
                        if (nhi - hi > T && count > 0 && pass == 0) { fresh(); continue; } // must start a run
                        wlo = nlo; whi = nhi; live = true;
                        break;
                    }
                } else if (wide[b]) {
                    live = false;
                }
                cum += wb;
                count++;
            }
            return (int)cuts.size();
        };
        long long target = (weight + wgs - 1) / wgs;
        int nruns = simulate(target);
        for (int it = 0; it < 4 && nruns > wgs; it++) { // forced cuts cost runs: make the weight cuts rarer
            const long long spare = (long long)wgs - (nruns - (weight + target - 1) / target);
            if (spare <= 0) break;
            target = (weight + spare - 1) / spare;
            nruns = simulate(target);
        }
        if (nruns <= wgs) {
            for (int g = 0; g < wgs; g++) {
                out.run_rng[2 * g] = g < nruns ? cuts[g] : nblk;
                out.run_rng[2 * g + 1] = g + 1 < nruns ? cuts[g + 1] : nblk;
            }
        }
    }
    if (ghost_lo < ghost_hi) {
        std::vector<char> gh((size_t)nblk, 0);
        for (int b = 0; b < nblk; b++)
            for (int r = rows[b]; r < rows[b + 1] && !gh[b]; r++)
                gh[b] = row_min[r] <= row_max[r] && (row_min[r] < ghost_lo || row_max[r] >= ghost_hi);
        int F = 0, T = 0;
        while (F < nblk && gh[F]) F++;
        while (T < nblk - F && gh[nblk - 1 - T]) T++;
        bool middle_clean = true;
        for (int b = F; b < nblk - T; b++) middle_clean = middle_clean && !gh[b];
        const int m = std::max(1, bpw - kGhostRunSlack);
        const int rf = (F + m - 1) / m, rt = (T + m - 1) / m, rmid = wgs - rf - rt;
        const long long M = (long long)nblk - F - T;
        if (middle_clean && (F || T) && rmid > 0 && M <= (long long)rmid * bpw) {
            int g = 0;
            for (int b = F; b < nblk - T; b += bpw, g++) { out.run_rng[2 * g] = b; out.run_rng[2 * g + 1] = std::min(nblk - T, b + bpw); }
            for (; g < rmid; g++) out.run_rng[2 * g] = out.run_rng[2 * g + 1] = 0; // idle runs
            for (int b = 0; b < F; b += m, g++) { out.run_rng[2 * g] = b; out.run_rng[2 * g + 1] = std::min(F, b + m); }
            for (int b = nblk - T; b < nblk; b += m, g++) { out.run_rng[2 * g] = b; out.run_rng[2 * g + 1] = std::min(nblk, b + m); }
            for (; g < wgs; g++) out.run_rng[2 * g] = out.run_rng[2 * g + 1] = 0;
        }
        for (int g = 0; g < wgs; g++)
            for (int b = out.run_rng[2 * g]; b < out.run_rng[2 * g + 1] && !out.run_halo[g]; b++) out.run_halo[g] = gh[b];
    }
    const int ring = cfg.ring;
    for (int g = 0; g < wgs; g++) {
        int wlo = 0, whi = 0, base = 0;
        bool live = false;
        long long run_nnz = 0, plain_nnz = 0;
        int nplain = 0;
        bool ok = true;
        for (int b = out.run_rng[2 * g]; b < out.run_rng[2 * g + 1]; b++) {
            const int nn = ptrs[b + 1] - ptrs[b], nrows = rows[b + 1] - rows[b];
            int* P = &out.plan[(size_t)8 * b];
            P[0] = rows[b]; P[1] = ptrs[b]; P[2] = nrows; P[3] = nn;
            // (a block without nonzeros brings no column; its lanes' loads — values never used — go where the last served block's
            // went, so that every load of a run stays inside the column range build_run_deps derives from these records)
            P[4] = live ? std::max(0, whi - 1) : 0; P[5] = 0; P[6] = base; P[7] = 0;
            run_nnz += nn;
            if (nrows > cfg.threads) out.lean = false; // (empty rows included: their zeros are stored by the loop's second pass)
            if (nn == 0) continue;
            const int cmin = bmin[b], cmax = bmax[b];
            bool use = !wide[b];
            if (use) {
                int lo = live ? wlo : cmin, hi = live ? whi : cmin;
                bool restart = !live;
                // a block that reaches back behind the window (block minima are not monotone, e.g.
                // FE rows) or jumps ahead of it restarts the window: its whole span is reloaded
                if (cmin < lo || cmin > hi) restart = true;
                if (restart) { // start as low as the ring allows: later blocks may dip below this one's cmin
                    lo = std::max(0, std::min(cmin, cmax + 1 - ring));
                    hi = lo;
                }
                int nhi = std::max(hi, cmax + 1), nlo = std::max(lo, nhi - ring);
                if (cmin < nlo) { // the window cannot keep its upper end AND reach down to cmin: give the upper end up —
                                  // restart on this block's own span (which fits), later blocks reload what they need
                    restart = true;
                    lo = std::max(0, std::min(cmin, cmax + 1 - ring));
                    hi = lo;
                    nhi = cmax + 1;
                    nlo = std::max(lo, nhi - ring);
                }
                if (cmin < nlo) use = false; // cannot hold [cmin, cmax] at once
                else {
                    if (restart) base = (lo / ring) * ring;
                    while (nlo - base >= ring) base += ring;
                    P[4] = hi; P[5] = nhi - hi; P[6] = base; P[7] = 1;
                    if ((nhi - hi > cfg.threads && b != out.run_rng[2 * g]) || nrows > cfg.threads) out.lean = false;
                    wlo = nlo; whi = nhi; live = true;
                }
            }
            if (!use) { // PLAIN: computed behind the loop; the window starts afresh on the next block
                P[2] = 0; P[4] = nrows; P[5] = 0; P[6] = base; P[7] = 2;
                live = false;
                nplain++;
                plain_nnz += nn;
            }
        }
        if (nplain > kRingMaxPlain) ok = false;
        if (!ok) {
            out.run_ok[g] = 0;
            out.bad_runs++;
            out.bad_nnz += run_nnz;
            for (int b = out.run_rng[2 * g]; b < out.run_rng[2 * g + 1]; b++) { // the plain path reads plain records
                int* P = &out.plan[(size_t)8 * b];
                if (P[7] == 2) { P[2] = P[4]; P[4] = 0; P[7] = 0; }
            }
        } else {
            out.bad_nnz += plain_nnz;
            if (nplain > 0) out.run_ok[g] = 3; // tells the kernel to look for PLAIN blocks behind its loop
        }
    }
}

// The one-launch matrix-powers step (spmk_ring.hpp): run g's power p + 1 reads power p of other runs' rows.  For a SQUARE
// matrix: dep_run[dep_ptr[g] .. dep_ptr[g + 1]) = the runs (itself included) whose row range meets the range of columns run g
// LOADS — not only the columns its rows name: the window's first fill may start below the first block's smallest column, and
// lanes past a block's new columns load up to T - 1 entries beyond them (values never used).  With every load inside the
// published range no cache of the consumer's XCD can ever hold a line of y_p from before its publication, which is what lets
// the kernel do without an acquire between powers.  Taken from the plan records ({new_lo, new_cnt} per block) exactly as the
// kernel replays them.  Runs are contiguous block ranges but need not be in row order; runs without rows depend on nothing and
// nobody depends on them.  Only for plans whose every row is computed inside the ring loop (no PLAIN block, no plain run).
inline void build_run_deps(const RingPlanHost& P, int n, std::vector<int>& dep_ptr, std::vector<int>& dep_run)
{
    const int W = P.wgs, T = P.cfg.threads;
    std::vector<int> r0((size_t)W, 0), r1((size_t)W, 0), order;
    for (int g = 0; g < W; g++) {
        const int b0 = P.run_rng[2 * g], b1 = P.run_rng[2 * g + 1];
        if (b0 >= b1) continue;
        r0[g] = P.plan[(size_t)8 * b0];
        r1[g] = P.plan[(size_t)8 * (b1 - 1)] + P.plan[(size_t)8 * (b1 - 1) + 2];
        if (r1[g] > r0[g]) order.push_back(g);
    }
    std::sort(order.begin(), order.end(), [&](int a, int b) { return r0[a] < r0[b]; });
    dep_ptr.assign((size_t)W + 1, 0);
    dep_run.clear();
    for (int g = 0; g < W; g++) {
        dep_ptr[g] = (int)dep_run.size();
        if (r1[g] <= r0[g]) continue;
        int cmin = 0x7fffffff, cmax = -1;
        for (int b = P.run_rng[2 * g]; b < P.run_rng[2 * g + 1]; b++) {
            const int* Q = &P.plan[(size_t)8 * b];
            cmin = std::min(cmin, Q[4]);
            cmax = std::max(cmax, std::min(n - 1, Q[4] + std::max(Q[5], T) - 1)); // new columns, or the T lanes' over-read
        }
        size_t lo = 0, hi = order.size();
        while (lo < hi) { // first run (in row order) whose rows end beyond cmin
            const size_t mid = (lo + hi) / 2;
            if (r1[order[mid]] > cmin) hi = mid; else lo = mid + 1;
        }
        for (size_t i = lo; i < order.size() && r0[order[i]] <= cmax; i++) dep_run.push_back(order[i]);
    }
    dep_ptr[W] = (int)dep_run.size();
}

// The 16-bit column stream of the ring kernel: for block b, thread t, i < PER the
// ring slot of nonzero k = t + i*T of the block (the last nonzero again for k >= nnz of the
// block) at out[(b*T + t)*PER + i] — configuration 4: of nonzero k = 2(t + (i/2)T) + (i & 1), see ring_pairs().  Blocks the
// ring does not serve get zeros (their runs take the plain path, which reads indcol).
inline void build_ring_slots(const RingPlanHost& P, const int* indcol, std::vector<unsigned short>& out)
{
    const int T = P.cfg.threads, per = P.cfg.nnzb / T, ring = P.cfg.ring;
    out.assign((size_t)P.nblk * P.cfg.nnzb, 0);
    for (int b = 0; b < P.nblk; b++) {
        const int* Q = &P.plan[(size_t)8 * b];
        const int p0 = Q[1], nn = Q[3], base = Q[6];
        if (Q[7] != 1 || nn <= 0 || nn > P.cfg.nnzb) continue;
        unsigned short* o = &out[(size_t)b * P.cfg.nnzb];
        const bool pair = ring_pairs(T);
        for (int t = 0; t < T; t++)
            for (int i = 0; i < per; i++) {
                const int k = std::min(pair ? 2 * (t + (i >> 1) * T) + (i & 1) : t + i * T, nn - 1);
                int p = indcol[p0 + k] - base;
                if (p >= ring) p -= ring;
                o[t * per + i] = (unsigned short)p;
            }
    }
}

} // namespace mi355
