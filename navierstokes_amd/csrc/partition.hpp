// partition.hpp — host-side planner for a row-range partition of one CSR matrix
// over the ranks (GPUs) of a node.  Pure C++ integer work, no HIP: it runs and is
// tested on CPU-only machines (tests/test_partition*.py).
//
// New design: the reference is strictly single-process (all PETSc
// objects are Seq, src/solve_newton.c:972,981).  What it must reproduce is only
// the arithmetic: every row of y = A x is still the same CSR-ordered fma chain,
// because the column relabelling below is monotone inside each of the two
// classes (owned, ghost) but NOT across them — so each local row keeps its
// nonzeros in the ORIGINAL global-column order and only the x index changes.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

namespace mi355 {

struct LocalPiece {           // interior or boundary rows of one rank
    std::vector<int> ptrow;   // [nrows+1]
    std::vector<int> indcol;  // local column ids
    std::vector<double> coef;
    std::vector<int> rowmap;  // local row index of each piece row
    std::vector<int> src;     // position of each nonzero in the caller's (rank-local) CSR arrays: refreshes coef in place
};

struct PartPlan {
    int nranks = 0, rank = 0;
    std::vector<long long> row_starts; // [nranks+1]
    int n_local = 0, n_halo = 0;
    std::vector<long long> halo_ids;   // ascending global ids of ghost columns
    std::vector<int> recv_counts, recv_offsets; // per peer, into halo_ids
    std::vector<int> send_counts, send_offsets; // per peer, into send_idx
    std::vector<std::vector<int>> send_lists;   // local ids per peer
    std::vector<int> send_idx;                  // concatenated by peer
    LocalPiece piece[2];               // 0 interior, 1 boundary
    LocalPiece all;                    // interior rows then boundary rows in ONE piece (the fused step); built on demand
    bool sends_set = false;
    bool sends_contiguous = false;     // every non-empty send list is a run of consecutive local ids: no pack needed

    // returns "" or an error message
    std::string build(int nranks_, int rank_, const long long* rs, const int* ptrow, const int* indcol,
                      const double* coef)
    {
        if (nranks_ < 1 || rank_ < 0 || rank_ >= nranks_ || !rs || !ptrow) return "bad rank/row_starts";
        nranks = nranks_;
        rank = rank_;
        row_starts.assign(rs, rs + nranks + 1);
        for (int p = 0; p < nranks; p++)
            if (row_starts[p] > row_starts[p + 1]) return "row_starts must be non-decreasing";
        if (row_starts[0] != 0) return "row_starts[0] must be 0";
        if (row_starts[nranks] > INT32_MAX) return "global row count exceeds int32 (mpk/SpMV.h:20-22 uses int)";
        const long long lo = row_starts[rank], hi = row_starts[rank + 1];
        n_local = (int)(hi - lo);
        const long long nglob = row_starts[nranks];
        if (ptrow[0] != 0) return "ptrow must be relative to the rank's first nonzero (ptrow[0] == 0)";
        const int nnz = ptrow[n_local];
        if (nnz > 0 && (!indcol || !coef)) return "null indcol/coef";

        // ghost columns
        std::vector<long long> ghosts;
        for (int k = 0; k < nnz; k++) {
            const long long c = indcol[k];
            if (c < 0 || c >= nglob) return "column index out of range";
            if (c < lo || c >= hi) ghosts.push_back(c);
        }
        std::sort(ghosts.begin(), ghosts.end());
        ghosts.erase(std::unique(ghosts.begin(), ghosts.end()), ghosts.end());
        // Where the ghosts of one owner nearly fill a range (banded matrices: the rows just across the
        // partition boundary), take the WHOLE range: the owner can then send a contiguous slice of its x
        // without a pack kernel, at the price of a few unused entries.  Column relabelling below only
        // needs the list to be ascending, so nothing else changes.
        const char* dense_env = getenv("MI355_PART_DENSE_HALO"); // 0: keep the exact ghost sets (tests of the packed path)
        if (!(dense_env && dense_env[0] == '0')) {
            std::vector<long long> dense;
            size_t i = 0;
            while (i < ghosts.size()) {
                int p = 0;
                while (ghosts[i] >= rs[p + 1]) p++;
                size_t j = i;
                while (j < ghosts.size() && ghosts[j] < rs[p + 1]) j++;
                const long long gmin = ghosts[i], gmax = ghosts[j - 1], cnt = (long long)(j - i);
                if (gmax - gmin + 1 <= cnt + cnt / 2 + 64)
                    for (long long g = gmin; g <= gmax; g++) dense.push_back(g);
                else
                    dense.insert(dense.end(), ghosts.begin() + i, ghosts.begin() + j);
                i = j;
            }
            ghosts.swap(dense);
        }
        halo_ids.swap(ghosts);
        n_halo = (int)halo_ids.size();
        if ((long long)n_local + n_halo > INT32_MAX) return "n_local + n_halo exceeds int32";

        recv_counts.assign(nranks, 0);
        recv_offsets.assign(nranks + 1, 0);
        {
            int p = 0;
            for (int h = 0; h < n_halo; h++) {
                while (halo_ids[h] >= row_starts[p + 1]) p++;
                recv_counts[p]++;
            }
            for (int q = 0; q < nranks; q++) recv_offsets[q + 1] = recv_offsets[q] + recv_counts[q];
        }
        if (recv_counts[rank] != 0) return "internal: ghost owned by self";

        // relabel + split
        for (int w = 0; w < 2; w++) {
            piece[w] = LocalPiece();
            piece[w].ptrow.push_back(0);
        }
        // Which rows wait for the halo.  A row that names a ghost column must; a banded rank's first and last ~w rows do so one here, one
        // not (the first ghost-free row lies far in front of the last ghost-naming one), which would leave the interior piece with a
        // scattered row map — two 8-byte mapped stores per lane in the sliced kernel, a gather of row ids in the others (sim_rank 8 1:
        // 20.5 us against 18.8 for the same rows behind a plain offset).  So where ONE run of consecutive ghost-free rows holds at least
        // 90 % of all ghost-free rows, that run is the interior piece (row r -> y[r + offset]) and every other row joins the boundary
        // piece, ghost-free or not: a few hundred rows computed behind the exchange instead of beside it.  MI355_PART_CONTIGUOUS_INTERIOR=0
        // keeps the exact split.
        std::vector<char> is_boundary((size_t)n_local, 0);
        long long free_rows = 0;
        for (int r = 0; r < n_local; r++) {
            for (int k = ptrow[r]; k < ptrow[r + 1]; k++) {
                const long long c = indcol[k];
                if (c < lo || c >= hi) { is_boundary[r] = 1; break; }
            }
            free_rows += !is_boundary[r];
        }
        {
            const char* ce = getenv("MI355_PART_CONTIGUOUS_INTERIOR");
            int best_lo = 0, best_len = 0;
            for (int r = 0; r < n_local;) {
                if (is_boundary[r]) { r++; continue; }
                int e = r;
                while (e < n_local && !is_boundary[e]) e++;
                if (e - r > best_len) { best_len = e - r; best_lo = r; }
                r = e;
            }
            if (!(ce && ce[0] == '0') && best_len < free_rows && 10LL * best_len >= 9LL * free_rows)
                for (int r = 0; r < n_local; r++)
                    if (r < best_lo || r >= best_lo + best_len) is_boundary[r] = 1;
        }
        for (int r = 0; r < n_local; r++) {
            const bool boundary = is_boundary[r] != 0;
            LocalPiece& P = piece[boundary ? 1 : 0];
            for (int k = ptrow[r]; k < ptrow[r + 1]; k++) {
                const long long c = indcol[k];
                int lc;
                if (c >= lo && c < hi) lc = (int)(c - lo);
                else lc = n_local + (int)(std::lower_bound(halo_ids.begin(), halo_ids.end(), c) - halo_ids.begin());
                P.indcol.push_back(lc);
                P.coef.push_back(coef[k]);
                P.src.push_back(k);
            }
            P.ptrow.push_back((int)P.indcol.size());
            P.rowmap.push_back(r);
        }
        send_counts.assign(nranks, 0);
        send_offsets.assign(nranks + 1, 0);
        send_lists.assign(nranks, std::vector<int>());
        send_idx.clear();
        sends_set = (nranks == 1);
        return "";
    }

    // The one-launch step works on ALL local rows at once, in their natural order (no row map), with the columns in a
    // numbering of its own: [ghosts owned by lower ranks | owned | ghosts owned by higher ranks] — ascending global id
    // throughout, so a banded matrix stays banded across the partition boundary and the ring kernel serves the boundary
    // rows like any others.  n_left = number of ghosts in front.  Ghost g of the halo (x_ext order) is column g if
    // g < n_left, else g + n_local; owned entry i is column n_left + i.
    int n_left = 0;
    void build_combined()
    {
        if (!all.ptrow.empty()) return;
        n_left = recv_offsets[rank];
        std::vector<int> where((size_t)n_local), which((size_t)n_local);
        for (int w = 0; w < 2; w++)
            for (size_t r = 0; r < piece[w].rowmap.size(); r++) {
                where[piece[w].rowmap[r]] = (int)r;
                which[piece[w].rowmap[r]] = w;
            }
        all.ptrow.push_back(0);
        for (int r = 0; r < n_local; r++) {
            const LocalPiece& L = piece[which[r]];
            for (int k = L.ptrow[where[r]]; k < L.ptrow[where[r] + 1]; k++) {
                const int c = L.indcol[k];
                const int g = c - n_local;
                all.indcol.push_back(c < n_local ? n_left + c : (g < n_left ? g : g + n_local));
                all.coef.push_back(L.coef[k]);
            }
            all.ptrow.push_back((int)all.indcol.size());
        }
    }

    // All local rows in natural order with the columns as the pieces have them — [owned | halo], the layout of x_ext itself: the piece of
    // the one-launch step that first copies its ghosts out of the receive window into x_ext's halo part (capi_part.hip, round 5).
    LocalPiece all_ext;
    void build_all_ext()
    {
        if (!all_ext.ptrow.empty()) return;
        std::vector<int> where((size_t)n_local), which((size_t)n_local);
        for (int w = 0; w < 2; w++)
            for (size_t r = 0; r < piece[w].rowmap.size(); r++) {
                where[piece[w].rowmap[r]] = (int)r;
                which[piece[w].rowmap[r]] = w;
            }
        all_ext.ptrow.push_back(0);
        for (int r = 0; r < n_local; r++) {
            const LocalPiece& L = piece[which[r]];
            for (int k = L.ptrow[where[r]]; k < L.ptrow[where[r] + 1]; k++) {
                all_ext.indcol.push_back(L.indcol[k]);
                all_ext.coef.push_back(L.coef[k]);
            }
            all_ext.ptrow.push_back((int)all_ext.indcol.size());
        }
    }

    std::string set_send(int peer, int count, const long long* ids)
    {
        if (peer < 0 || peer >= nranks || count < 0 || (count > 0 && !ids)) return "bad peer/count";
        if (peer == rank && count != 0) return "a rank never sends to itself";
        const long long lo = row_starts[rank], hi = row_starts[rank + 1];
        std::vector<int> l(count);
        for (int i = 0; i < count; i++) {
            if (ids[i] < lo || ids[i] >= hi) return "peer asked for a row this rank does not own";
            l[i] = (int)(ids[i] - lo);
        }
        send_lists[peer].swap(l);
        // rebuild the concatenation
        send_idx.clear();
        for (int p = 0; p < nranks; p++) {
            send_counts[p] = (int)send_lists[p].size();
            send_offsets[p + 1] = send_offsets[p] + send_counts[p];
            send_idx.insert(send_idx.end(), send_lists[p].begin(), send_lists[p].end());
        }
        sends_set = true;
        sends_contiguous = true;
        for (int p = 0; p < nranks && sends_contiguous; p++)
            for (size_t i = 1; i < send_lists[p].size(); i++)
                if (send_lists[p][i] != send_lists[p][0] + (int)i) { sends_contiguous = false; break; }
        return "";
    }
};

// Row-block table of the stream kernels: consecutive rows with <= nnzb nonzeros
// (and <= max_rows rows); a row longer than nnzb is a block of its own.
// out: {first row, first nnz} per block plus the terminator {n, nnz}.
// row_align > 1: block boundaries are pulled back to multiples of row_align rows where possible,
// so that the y segment a block writes starts on a 128-byte line (row_align = 16).
// row_align > 1: blocks end on multiples of row_align rows; keep_eighths > 0: only where the shortened block keeps at least that
// many eighths of its nonzeros (the one-thread-per-row kernels want whole waves of rows, but not half-empty blocks)
inline void build_row_blocks(int n, const int* ptrow, int nnzb, int max_rows, std::vector<int>& out_rows,
                             std::vector<int>& out_ptr, int row_align = 1, int keep_eighths = 0)
{
    out_rows.clear();
    out_ptr.clear();
    int r = 0;
    while (r < n) {
        const int start = r;
        const int p0 = ptrow[r];
        int e = r + 1; // a block always takes at least one row
        while (e < n && (e - start) < max_rows && (long long)ptrow[e + 1] - p0 <= nnzb) e++;
        if (row_align > 1 && e < n) {
            const int ea = (e / row_align) * row_align;
            if (ea > start && 8 * (long long)(ptrow[ea] - p0) >= (long long)keep_eighths * (ptrow[e] - p0)) e = ea;
        }
        out_rows.push_back(start);
        out_ptr.push_back(p0);
        r = e;
    }
    out_rows.push_back(n);
    out_ptr.push_back(n > 0 ? ptrow[n] : 0);
}

// CSR with 4x4 node-block structure (what the reference's FE assembly produces:
// src/benchmark_spmv.c:104-118 inserts whole 4x4 blocks) -> BCSR 4x4 with row-major blocks
// (mpk/SpMV.h:26-33).  Succeeds only if the conversion loses nothing and keeps every row's
// nonzeros in the same order: n % 4 == 0, the four rows of a block row hold the same columns, and
// these come in aligned groups {4j, 4j+1, 4j+2, 4j+3}.  Then a BCSR row chain visits exactly the
// CSR row's terms in CSR order, so the two kernels return the same bits.
// pattern-only form of the test in csr_to_bcsr4_exact below (no values copied)
inline bool csr_has_block4_pattern(int n, const int* ptrow, const int* indcol)
{
    if (n <= 0 || n % 4 != 0) return false;
    for (int b = 0; b < n / 4; b++) {
        const int p0 = ptrow[4 * b], len = ptrow[4 * b + 1] - p0;
        if (len % 4 != 0) return false;
        for (int r = 1; r < 4; r++)
            if (ptrow[4 * b + r + 1] - ptrow[4 * b + r] != len) return false;
        for (int g = 0; g < len; g += 4) {
            const int c0 = indcol[p0 + g];
            if (c0 % 4 != 0) return false;
            for (int k = 1; k < 4; k++)
                if (indcol[p0 + g + k] != c0 + k) return false;
        }
        for (int r = 1; r < 4; r++) {
            const int pr = ptrow[4 * b + r];
            for (int k = 0; k < len; k++)
                if (indcol[pr + k] != indcol[p0 + k]) return false;
        }
    }
    return true;
}

inline bool csr_to_bcsr4_exact(int n, const int* ptrow, const int* indcol, const double* coef, std::vector<int>& bptr,
                               std::vector<int>& bcol, std::vector<double>& bval)
{
    bptr.clear();
    bcol.clear();
    bval.clear();
    if (n <= 0 || n % 4 != 0) return false;
    const int nb = n / 4;
    bptr.resize((size_t)nb + 1);
    bptr[0] = 0;
    for (int b = 0; b < nb; b++) {
        const int p0 = ptrow[4 * b], len = ptrow[4 * b + 1] - p0;
        if (len % 4 != 0) return false;
        for (int r = 1; r < 4; r++)
            if (ptrow[4 * b + r + 1] - ptrow[4 * b + r] != len) return false;
        for (int g = 0; g < len; g += 4) {
            const int c0 = indcol[p0 + g];
            if (c0 % 4 != 0) return false;
            for (int k = 1; k < 4; k++)
                if (indcol[p0 + g + k] != c0 + k) return false;
        }
        for (int r = 1; r < 4; r++) {
            const int pr = ptrow[4 * b + r];
            for (int k = 0; k < len; k++)
                if (indcol[pr + k] != indcol[p0 + k]) return false;
        }
        bptr[b + 1] = bptr[b] + len / 4;
    }
    const size_t nblk = (size_t)bptr[nb];
    bcol.resize(nblk);
    bval.resize(16 * nblk);
    for (int b = 0; b < nb; b++) {
        const int p0 = ptrow[4 * b], nbk = bptr[b + 1] - bptr[b];
        for (int g = 0; g < nbk; g++) {
            const size_t blk = (size_t)bptr[b] + g;
            bcol[blk] = indcol[p0 + 4 * g] / 4;
            for (int r = 0; r < 4; r++) {
                const int pr = ptrow[4 * b + r] + 4 * g;
                for (int k = 0; k < 4; k++) bval[16 * blk + 4 * r + k] = coef[pr + k];
            }
        }
    }
    return true;
}

} // namespace mi355
