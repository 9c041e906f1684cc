// mring_plan.hpp — host-side (no HIP) plan of the multi-window ring kernel (spmv_mring.hpp).
//
// The ring kernel (spmv_ring.hpp) keeps ONE sliding window of x in LDS: right for a band, useless for a 3-D mesh, whose
// rows reach into the mesh plane (or Cuthill-McKee level) below, their own, and the one above — three narrow clusters of
// columns, two whole planes apart (P1 pressure operator on 100^3 cells, natural or relabelled order: one window would
// have to span 16-21 k columns, THREE windows 0.7-1.0 k in total; tile_plan.hpp's neighbours' measurement).  And the
// clusters slide: along a plane each moves forward with the rows, and where a level ends the next one begins right behind
// it in the numbering, so the windows just keep sliding.  So: K = 5 independent sliding windows, each a ring of
// RING / K entries of the same LDS array; per block the host plan says, per window, which columns enter; every nonzero's
// LDS slot is precomputed as a 16-bit number exactly as for the single ring.  The kernel is the ring kernel with a
// K-way refill — same pipeline, same row chains, same bits.
//
// Plan, per run (a workgroup's consecutive blocks), per block:
//   * the block's distinct columns are cut into clusters wherever two neighbours lie kMringGap or more apart;
//   * a cluster continues the window it starts in (or just above: within kMringGap of its upper end), sliding it; clusters
//     that continue nothing take windows no cluster of this block uses (their old content is dropped: a restart of that
//     window alone);
//   * more than K clusters, or a cluster wider than a window: the block is PLAIN (computed behind the loop, like the single
//     ring's), and all windows start afresh behind it.
// Pure integer work; checked by replay in mi_mring_plan_probe (every nonzero's slot holds its column when its block runs).
#pragma once
#include <algorithm>
#include <vector>

#include "ring_plan.hpp"

namespace mi355 {

constexpr int kMringK = 5;          // windows
constexpr int kMringRing = 4800;    // doubles of LDS for all windows together (with staging and the plan records 78 992 B per
                                    // workgroup: two per CU, like the single ring's 80 224)
constexpr int kMringW = kMringRing / kMringK; // entries per window
constexpr int kMringGap = 256;      // neighbouring distinct columns this far apart belong to different clusters
// (relabelled 100^3-cell mesh, blocks of 2048 nonzeros: 3 clusters in 92 %, 5 where the rows cross from one Cuthill-McKee level
// into the next, in 7 %; widest cluster 265 columns median, 461 at the 99th percentile: K = 5 windows of 1024 (960: the same) serve 99.7 %,
// K = 4 of 1280 only 93 %.)
// record slots of window w: first new column at plan[kMringLoAt(w)], count | base index << 11 at plan[kMringPkAt(w)]
constexpr int kMringLoAt(int w) { return w < 4 ? 8 + w : 6; }
constexpr int kMringPkAt(int w) { return w < 4 ? 12 + w : 7; }
static_assert(kMringW % 64 == 0, "windows are refilled in groups of 64 columns");
constexpr int kMringNnzb = 2048, kMringThreads = 256, kMringWgUnit = 512;
constexpr int kMringFast = 4 * kMringThreads; // new columns a block inside a run may bring (the kernel prefetches 4 per thread:
                                              // three windows advancing by a block's <= 256 rows, padded to 64, stay below)
constexpr int kMringMaxB = 96;      // blocks per run: the plan records are 64 bytes each and two workgroups must fit a CU's LDS

struct MringPlanHost {
    int nblk = 0, wgs = 0, bpw = 0, bad_runs = 0;
    long long bad_nnz = 0;
    // 16 ints per block: {r0, p0, rows, nnz} {flags, total new, new_lo[4], pk[4]} {new_lo[0..3]} {pk[0..3]}, pk = count | base index << 11
    // flags 1 = window-served, 2 = PLAIN {r0, p0, 0, nnz} {2, 0, 0, 0} {rows, 0, 0, 0}, 0 = empty
    std::vector<int> plan;
    std::vector<int> run_ok;
    std::vector<int> run_rng;
    std::vector<unsigned short> slots; // per block kMringNnzb entries in thread order (as build_ring_slots)
    long long restarts = 0;            // window restarts in the middle of a run (diagnostic)
};

inline void build_mring_plan(int n, const int* ptrow, const int* indcol, MringPlanHost& out)
{
    out = MringPlanHost();
    const int K = kMringK, W = kMringW, T = kMringThreads, nnzb = kMringNnzb, per = nnzb / T;
    std::vector<int> rows, ptrs;
    build_row_blocks(n, ptrow, nnzb, T, rows, ptrs); // <= T rows per block: the kernel makes one pass over a block's rows
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
            for (size_t i = first; i < cls.size() && !bad; i++) bad = cls[i].hi - cls[i].lo + 1 > W;
            if (bad) { cls.resize(first); wide[b] = 1; }
        }
        cl_ptr[nblk] = (int)cls.size();
    }
    long long weight = 0;
    for (int b = 0; b < nblk; b++) weight += wide[b] ? kRingPlainWeight : 1;
    const long long per_unit = (long long)(kMringMaxB - kRingPlainWeight) * kMringWgUnit;
    int wgs = kMringWgUnit * (int)((weight + per_unit - 1) / per_unit);
    if (wgs < kMringWgUnit) wgs = kMringWgUnit;
    out.plan.assign((size_t)16 * nblk, 0);
    out.slots.assign((size_t)nblk * nnzb, 0);

    struct Win { int lo[kMringK], hi[kMringK], base[kMringK]; bool live[kMringK]; };
    auto reset = [&](Win& S) { for (int w = 0; w < K; w++) { S.lo[w] = S.hi[w] = S.base[w] = 0; S.live[w] = false; } };
    std::vector<int> win_of;
    long long restarts = 0;
    // one served block against window state S: fills its record P (and, if o != nullptr, its slots); returns the new columns
    auto plan_block = [&](int b, Win& S, int* P, unsigned short* o) {
        const Cl* C = &cls[cl_ptr[b]];
        const int nc = cl_ptr[b + 1] - cl_ptr[b], nn = ptrs[b + 1] - ptrs[b], p0 = ptrs[b];
        win_of.assign(nc, -1);
        bool used[kMringK] = {};
        for (int j = 0; j < nc; j++) // which window does each cluster continue?
            for (int w = 0; w < K; w++)
                if (S.live[w] && !used[w] && C[j].lo >= S.lo[w] && C[j].lo < S.hi[w] + kMringGap) { win_of[j] = w; used[w] = true; break; }
        for (int j = 0; j < nc; j++) { // the others: a window nobody uses in this block (prefer one that holds nothing)
            if (win_of[j] >= 0) continue;
            int pick = -1;
            for (int w = 0; w < K && pick < 0; w++)
                if (!used[w] && !S.live[w]) pick = w;
            for (int w = 0; w < K && pick < 0; w++)
                if (!used[w]) pick = w;
            win_of[j] = pick; // nc <= K: there is one
            used[pick] = true;
            S.live[pick] = false;
        }
        int total_new = 0;
        int nlo_[kMringK] = {}, ncnt_[kMringK] = {};
        // New columns come in whole groups of 64 (the window simply runs a little ahead of what the block needs): a wave of
        // the kernel then refills ONE window and decodes its share of the record with scalar instructions.
        auto pad64 = [](int from, int to) { return from + ((to - from + 63) & ~63); };
        for (int j = 0; j < nc; j++) {
            const int w = win_of[j], cmin = C[j].lo, cmax = C[j].hi;
            int lo = S.live[w] ? S.lo[w] : cmin, hi = S.live[w] ? S.hi[w] : cmin;
            bool restart = !S.live[w];
            int nhi = pad64(hi, std::max(hi, cmax + 1)), nlo = std::max(lo, nhi - W);
            if (!restart && cmin < nlo) restart = true; // cannot keep the upper end and reach down: start afresh on this cluster
            if (restart) {
                lo = std::max(0, std::min(cmin, cmax + 1 - W));
                hi = lo;
                nhi = pad64(lo, cmax + 1); // <= lo + W: W is a multiple of 64
                nlo = std::max(lo, nhi - W);
                S.base[w] = (lo / W) * W;
            }
            while (nlo - S.base[w] >= W) S.base[w] += W;
            nlo_[w] = hi;
            ncnt_[w] = nhi - hi;
            total_new += nhi - hi;
            S.lo[w] = nlo; S.hi[w] = nhi; S.live[w] = true;
        }
        P[4] = 1; P[5] = total_new;
        for (int w = 0; w < K; w++) {
            P[kMringLoAt(w)] = nlo_[w];
            P[kMringPkAt(w)] = ncnt_[w] | ((S.base[w] / W) << 11);
        }
        if (o)
            for (int t = 0; t < T; t++)
                for (int i = 0; i < per; i++) {
                    const int k = std::min(t + i * T, nn - 1);
                    const int c = indcol[p0 + k];
                    int j = 0;
                    while (j + 1 < nc && c > C[j].hi) j++;
                    const int w = win_of[j];
                    int sl = c - S.base[w];
                    if (sl >= W) sl -= W;
                    o[t * per + i] = (unsigned short)(w * W + sl);
                }
        return total_new;
    };
    // Runs are formed on the way: a run ends where its weight is used up, and — the kernel's loop has no unpipelined refill —
    // in front of every block that would bring more than kMringFast new columns at once (several windows starting afresh):
    // such a block starts a run, whose first block's windows the kernel fills whole.
    std::vector<int> cuts;
    // (forced cuts ADD runs — short ones — instead of lengthening the others: the grid grows by whole rounds of workgroups, and
    // the hardware deals the later rounds out as the first workgroups finish)
    const long long target = (weight + wgs - 1) / wgs;
    {
        cuts.assign(1, 0);
        restarts = 0;
        Win S;
        reset(S);
        long long cum = 0;
        int count = 0, forced = 0;
        for (int b = 0; b < nblk; b++) {
            const int nn = ptrs[b + 1] - ptrs[b], nrows = rows[b + 1] - rows[b], wb = wide[b] ? kRingPlainWeight : 1;
            auto fresh = [&]() { if (b > cuts.back()) cuts.push_back(b); cum = 0; count = 0; reset(S); };
            if (count > 0 && (count >= kMringMaxB || cum + wb > target)) fresh();
            int* P = &out.plan[(size_t)16 * b];
            for (int q = 0; q < 16; q++) P[q] = 0;
            P[0] = rows[b]; P[1] = ptrs[b]; P[2] = nrows; P[3] = nn;
            if (nn > 0 && wide[b]) {
                P[2] = 0; P[4] = 2; P[8] = nrows;
                reset(S);
            } else if (nn > 0) {
                Win trial = S;
                int tn = plan_block(b, trial, P, nullptr);
                if (tn > kMringFast && count > 0) { fresh(); forced++; } // this block starts a run: its windows are filled whole
                plan_block(b, S, P, &out.slots[(size_t)b * nnzb]);        // commit: advances S
            }
            cum += wb;
            count++;
        }
        restarts = forced;
    }
    while ((int)cuts.size() > wgs) wgs += kMringWgUnit;
    const int nruns = (int)cuts.size();
    out.wgs = wgs;
    out.bpw = (nblk + wgs - 1) / wgs;
    out.restarts = restarts;
    out.run_ok.assign(wgs, 1);
    out.run_rng.assign((size_t)2 * wgs, 0);
    for (int g = 0; g < wgs; g++) {
        out.run_rng[2 * g] = g < nruns ? cuts[g] : nblk;
        out.run_rng[2 * g + 1] = g + 1 < nruns ? cuts[g + 1] : nblk;
    }
    for (int g = 0; g < nruns; g++) { // runs with too many PLAIN blocks go down the plain path as a whole
        int nplain = 0;
        long long run_nnz = 0, plain_nnz = 0;
        for (int b = out.run_rng[2 * g]; b < out.run_rng[2 * g + 1]; b++) {
            const int* P = &out.plan[(size_t)16 * b];
            run_nnz += P[3];
            if (P[4] == 2) { nplain++; plain_nnz += P[3]; }
        }
        if (nplain > kRingMaxPlain) {
            out.run_ok[g] = 0;
            out.bad_runs++;
            out.bad_nnz += run_nnz;
            for (int b = out.run_rng[2 * g]; b < out.run_rng[2 * g + 1]; b++) {
                int* P = &out.plan[(size_t)16 * b];
                if (P[4] == 2) { P[2] = P[8]; P[8] = 0; }
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
    const int K = kMringK, W = kMringW, T = kMringThreads, nnzb = kMringNnzb, per = nnzb / T;
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
        const int* Q = &P.plan[(size_t)16 * b];
        const int brows = Q[4] == 2 ? Q[8] : Q[2];
        if (Q[0] != next_row || Q[1] != next_nz) return "plan does not cover rows / nonzeros in order";
        next_row += brows;
        next_nz += Q[3];
        if (Q[3] != ptrow[Q[0] + brows] - ptrow[Q[0]]) return "block nonzero count disagrees with ptrow";
        const int run = run_of[b];
        if (run != cur_run) {
            std::fill(content.begin(), content.end(), -1);
            cur_run = run;
        }
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
        if (Q[4] != 1 || Q[3] > nnzb || Q[2] > T) return "a served run holds a block the kernel cannot take";
        int total = 0;
        for (int w = 0; w < K; w++) {
            const int lo = Q[kMringLoAt(w)], cnt = Q[kMringPkAt(w)] & 2047;
            const long long base = (long long)((unsigned)Q[kMringPkAt(w)] >> 11) * W;
            total += cnt;
            if (cnt % 64 != 0 || cnt > W) return "a window's new columns are not whole groups of 64";
            for (int c = lo; c < lo + cnt; c++) {
                long long sl = c - base;
                if (sl >= W) sl -= W;
                if (sl < 0 || sl >= W) return "a new column falls outside its window";
                content[(size_t)w * W + sl] = c;
            }
        }
        if (total != Q[5]) return "total of new columns disagrees with the windows";
        if (total > kMringFast && b != P.run_rng[2 * run]) return "a block inside a run brings more new columns than the kernel prefetches";
        for (int k = 0; k < Q[3]; k++) {
            const int slot = P.slots[(size_t)b * nnzb + (size_t)(k % T) * per + k / T];
            if (slot < 0 || slot >= K * W) return "slot outside the LDS array";
            if (content[slot] != indcol[Q[1] + k]) return "a nonzero's slot does not hold its column when its block runs";
        }
    }
    if (next_row != n || next_nz != ptrow[n]) return "plan does not cover the matrix";
    return nullptr;
}

} // namespace mi355
