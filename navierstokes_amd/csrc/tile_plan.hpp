// tile_plan.hpp — host-side (no HIP) plan of the "tile" CSR kernel (spmv_tile.hpp): per row block the sorted list of
// the DISTINCT columns its nonzeros touch, and per nonzero the 16-bit position of its column in that list.
//
// Why: the reference's scalar matrices are P1 operators on unstructured tetrahedral meshes (pressure Poisson, ~15
// nonzeros per row; src/solve_newton.c:91-197).  Neighbouring rows of such a matrix share most of their columns — a
// block of 2048 nonzeros (~137 rows) of a 3-D mesh in Cuthill-McKee order touches ~570 distinct columns (measured:
// 0.28 per nonzero; a random band like S15: 0.79) — but their span grows like n^(2/3) (10 k columns at 330 k nodes,
// 60 k at 5 M), far beyond any contiguous LDS window, so the ring kernel cannot serve them and the stream kernel pays
// one 8-byte L1 gather per NONZERO (its bound: 3.3-3.4 TB/s, profiles/NOTES.md §4.2).  With the block's distinct columns listed,
// the kernel gathers each ONCE (sorted, so neighbouring lanes share lines) into an LDS tile and every per-nonzero
// access is a ds_read through a 16-bit index: 3-4x fewer global gathers, and the column stream shrinks from 4 to
// 2 + 4u bytes per nonzero (u = distinct columns per nonzero).
// Row terms keep their CSR order — the kernel's fma chain, and so every bit of y, is that of the other kernels.
// Pure integer work, threaded over blocks; tested on CPU-only machines (mi_tile_plan_probe).
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "partition.hpp"

namespace mi355 {

constexpr int kTileNnzb = 2048;   // nonzeros per row block (= capacity of the LDS tile: a block has at most that many distinct columns)
constexpr int kTileThreads = 256; // threads per workgroup of the kernel
constexpr int kTilePadNnz = 2048; // initialised padding behind the value / slot streams (unclamped tid-strided loads)

struct TilePlanHost {
    int nnzb = kTileNnzb;
    int nblk = 0;
    // per block b (nblk + 1 entries, the last one closes the arrays): {first row, first nonzero, first list entry, first slot}
    std::vector<int> desc;
    std::vector<unsigned> ulist;       // distinct columns of each block, ascending
    std::vector<unsigned short> slots; // per nonzero k of block b (at desc.slot0 + k - p0): index of its column in the block's list
    long long listed = 0;              // nonzeros living in listed blocks (a single row longer than nnzb is not listed)
    int max_unique = 0;
};

inline void build_tile_plan(int n, const int* ptrow, const int* indcol, TilePlanHost& out, int nnzb = kTileNnzb, int threads = 0,
                            int row_align = 64)
{
    out = TilePlanHost();
    out.nnzb = nnzb;
    std::vector<int> rows, ptrs;
    build_row_blocks(n, ptrow, nnzb, 4 * kTileThreads, rows, ptrs, row_align, 7); // whole waves of rows for the row-chain phase
                                                                                   // (ring_plan.hpp; the caller decides by matrix size)
    const int nblk = (int)rows.size() - 1;
    out.nblk = nblk;
    out.desc.assign((size_t)4 * (nblk + 1), 0);
    if (nblk <= 0) return;
    // slot segments start at multiples of 8 entries (16 bytes): one 16-byte load hands a thread eight slots
    std::vector<long long> slot0((size_t)nblk + 1, 0);
    for (int b = 0; b < nblk; b++) {
        const int nn = ptrs[b + 1] - ptrs[b];
        slot0[b + 1] = slot0[b] + (nn <= nnzb ? (nn + 7) / 8 * 8 : 0);
    }
    out.slots.assign((size_t)slot0[nblk] + kTilePadNnz, 0);
    std::vector<int> ucount((size_t)nblk, 0);
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        threads = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
        if ((long long)ptrow[n] < 2000000) threads = 1;
    }
    threads = std::min(threads, nblk);
    // pass 1 (threaded over contiguous block ranges): distinct columns per block into per-thread lists, slots in place
    std::vector<std::vector<unsigned>> part((size_t)threads);
    auto work = [&](int t) {
        const int b0 = (int)((long long)nblk * t / threads), b1 = (int)((long long)nblk * (t + 1) / threads);
        std::vector<unsigned> u;
        std::vector<unsigned>& mine = part[t];
        for (int b = b0; b < b1; b++) {
            const int p0 = ptrs[b], nn = ptrs[b + 1] - p0;
            if (nn <= 0 || nn > nnzb) continue;
            u.assign(indcol + p0, indcol + p0 + nn);
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            ucount[b] = (int)u.size();
            unsigned short* s = out.slots.data() + slot0[b];
            for (int k = 0; k < nn; k++)
                s[k] = (unsigned short)(std::lower_bound(u.begin(), u.end(), (unsigned)indcol[p0 + k]) - u.begin());
            for (int k = nn; k < (nn + 7) / 8 * 8; k++) s[k] = s[nn - 1];
            mine.insert(mine.end(), u.begin(), u.end());
        }
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    size_t total = 0;
    for (auto& v : part) total += v.size();
    out.ulist.reserve(total + kTileThreads);
    for (auto& v : part) {
        out.ulist.insert(out.ulist.end(), v.begin(), v.end());
        std::vector<unsigned>().swap(v);
    }
    out.ulist.resize(total + kTileThreads, 0); // padding: the kernel's clamped list loads never leave the array
    long long u0 = 0;
    for (int b = 0; b <= nblk; b++) {
        int* D = &out.desc[(size_t)4 * b];
        D[0] = rows[b];
        D[1] = ptrs[b];
        D[2] = (int)u0;
        D[3] = (int)slot0[b];
        if (b < nblk) {
            u0 += ucount[b];
            out.max_unique = std::max(out.max_unique, ucount[b]);
            if (ucount[b] > 0) out.listed += ptrs[b + 1] - ptrs[b];
        }
    }
}

// invariants of a plan against the matrix it was built from; returns nullptr or the first violation
inline const char* check_tile_plan(const TilePlanHost& P, int n, const int* ptrow, const int* indcol)
{
    if ((int)P.desc.size() != 4 * (P.nblk + 1)) return "descriptor table size";
    if (P.nblk == 0) return n == 0 ? nullptr : "no blocks for a matrix with rows";
    if (P.desc[0] != 0 || P.desc[1] != 0) return "first block does not start at row 0";
    for (int b = 0; b < P.nblk; b++) {
        const int* D = &P.desc[(size_t)4 * b];
        const int r0 = D[0], p0 = D[1], u0 = D[2], s0 = D[3], r1 = D[4], p1 = D[5], u1 = D[6];
        if (r1 <= r0 || r1 > n) return "blocks must take at least one row and stay inside the matrix";
        if (ptrow[r0] != p0 || ptrow[r1] != p1) return "block boundaries are not row boundaries";
        const int nn = p1 - p0, U = u1 - u0;
        if (nn > P.nnzb) {
            if (r1 != r0 + 1) return "a block over the nonzero limit must be a single row";
            if (U != 0) return "an over-long row must not be listed";
            continue;
        }
        if (s0 % 8 != 0) return "slot segment not 16-byte aligned";
        if (U > nn || (nn > 0 && U == 0)) return "distinct-column count out of range";
        for (int j = 1; j < U; j++)
            if (P.ulist[(size_t)u0 + j] <= P.ulist[(size_t)u0 + j - 1]) return "column list not strictly ascending";
        for (int k = 0; k < nn; k++) {
            const unsigned s = P.slots[(size_t)s0 + k];
            if ((int)s >= U) return "slot outside the block's list";
            if (P.ulist[(size_t)u0 + s] != (unsigned)indcol[p0 + k]) return "slot does not name the nonzero's column";
        }
    }
    if (P.desc[(size_t)4 * P.nblk] != n) return "last descriptor does not close the rows";
    return nullptr;
}

} // namespace mi355
