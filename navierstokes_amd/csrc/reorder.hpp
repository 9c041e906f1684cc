// reorder.hpp — host-side (no HIP) locality reordering of a square CSR matrix at mi_csr_create.
//
// Why: the reference's real input is an unstructured gmsh mesh (src/solve_newton.c:91-197; matrix
// producer src/benchmark_spmv.c:76-123).  Its node numbering is whatever the mesher emitted, so the
// columns of a row can lie anywhere in x.  On the GPU that costs twice: the ring kernel's LDS window
// cannot hold a row block's column span, and the kernels that gather x through L2 touch one 32-byte
// node per 128-byte line.  A symmetric relabelling A' = P A P^T with a bandwidth-reducing order
// restores both — and it can be done without touching a single bit of the result:
//   * row r moves to position perm[r] and its columns are renamed perm[c], but its nonzeros STAY IN
//     THE ORDER THE CALLER GAVE THEM, so the row's fma chain visits the same (coefficient, x value)
//     pairs in the same order;
//   * the library gathers x into the new numbering before the product and writes y straight back
//     to the caller's numbering (row map), so the caller never sees the permutation.
//
// The order is reverse Cuthill-McKee on the symmetrised pattern: per connected component a
// pseudo-peripheral start node (George & Liu: repeated BFS to the farthest minimum-degree node),
// breadth-first levels with neighbours taken by increasing degree, the whole order reversed.
// For matrices with exact 4x4 node-block structure (FE matrices, partition.hpp: csr_to_bcsr4_exact)
// the graph is the NODE graph and the four dofs of a node stay together, so the reordered matrix
// keeps its block structure and the BCSR kernel stays eligible.
// Pure integer work, O(nnz log d); tested on CPU-only machines (tests/test_reorder.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace mi355 {

struct Reorder {
    int block = 1;          // 1: rows are the graph's nodes; 4: node = four consecutive rows
    std::vector<int> perm;  // [n] new index of old row/column
    std::vector<int> iperm; // [n] old index of new row/column
    double spread_before = 0.0, spread_after = 0.0; // mean |column - row| over the nonzeros, in nodes
};

// mean distance of a nonzero from the diagonal, in nodes (block = 4: node = row / 4)
inline double mean_column_distance(int n, const int* ptrow, const int* indcol, int block, const int* perm_nodes = nullptr)
{
    const long long nnz = ptrow[n];
    if (nnz == 0) return 0.0;
    long double s = 0;
    for (int i = 0; i < n; i += block) { // block = 4: the four rows of a node hold the same columns
        const int ri = perm_nodes ? perm_nodes[i / block] : i / block;
        for (int k = ptrow[i]; k < ptrow[i + 1]; k += block) {
            const int cj = perm_nodes ? perm_nodes[indcol[k] / block] : indcol[k] / block;
            s += std::abs(cj - ri);
        }
    }
    return (double)(s * block * block / nnz);
}

// symmetrised node graph without self loops: adjacency lists sorted ascending, duplicates removed
inline void build_node_graph(int n, const int* ptrow, const int* indcol, int block, std::vector<long long>& gptr,
                             std::vector<int>& gadj)
{
    const int nn = n / block;
    std::vector<long long> cnt((size_t)nn + 1, 0);
    for (int b = 0; b < nn; b++) {
        const int r = b * block;
        for (int k = ptrow[r]; k < ptrow[r + 1]; k += block) {
            const int c = indcol[k] / block;
            if (c == b || c >= nn) continue;
            cnt[b + 1]++;
            cnt[c + 1]++;
        }
    }
    for (int b = 0; b < nn; b++) cnt[b + 1] += cnt[b];
    std::vector<int> raw((size_t)cnt[nn]);
    std::vector<long long> fill(cnt.begin(), cnt.end() - 1);
    for (int b = 0; b < nn; b++) {
        const int r = b * block;
        for (int k = ptrow[r]; k < ptrow[r + 1]; k += block) {
            const int c = indcol[k] / block;
            if (c == b || c >= nn) continue;
            raw[fill[b]++] = c;
            raw[fill[c]++] = b;
        }
    }
    gptr.assign((size_t)nn + 1, 0);
    gadj.clear();
    gadj.reserve(raw.size() / 2 + 16);
    for (int b = 0; b < nn; b++) {
        int* lo = raw.data() + cnt[b];
        int* hi = raw.data() + cnt[b + 1];
        std::sort(lo, hi);
        hi = std::unique(lo, hi);
        gadj.insert(gadj.end(), lo, hi);
        gptr[b + 1] = (long long)gadj.size();
    }
}

// reverse Cuthill-McKee order of the graph: order[k] = node visited k-th (before the reversal is applied by the caller)
inline void cuthill_mckee(int nn, const std::vector<long long>& gptr, const std::vector<int>& gadj, std::vector<int>& order)
{
    order.clear();
    order.reserve(nn);
    std::vector<int> level((size_t)nn, -1); // BFS scratch: -1 = unseen in the current search
    std::vector<char> placed((size_t)nn, 0);
    std::vector<int> q, touched;
    auto degree = [&](int v) { return (int)(gptr[v + 1] - gptr[v]); };
    // BFS from s over the not-yet-placed nodes; returns the farthest minimum-degree node and the depth
    auto bfs_far = [&](int s, int& depth) {
        q.clear();
        touched.clear();
        q.push_back(s);
        level[s] = 0;
        touched.push_back(s);
        size_t head = 0;
        int far = s;
        while (head < q.size()) {
            const int v = q[head++];
            if (level[v] > level[far] || (level[v] == level[far] && degree(v) < degree(far))) far = v;
            for (long long e = gptr[v]; e < gptr[v + 1]; e++) {
                const int w = gadj[e];
                if (placed[w] || level[w] >= 0) continue;
                level[w] = level[v] + 1;
                touched.push_back(w);
                q.push_back(w);
            }
        }
        depth = level[far];
        for (int v : touched) level[v] = -1;
        return far;
    };
    // candidate starts in order of increasing degree (a low-degree node is a good first guess per component)
    std::vector<int> by_degree((size_t)nn);
    std::iota(by_degree.begin(), by_degree.end(), 0);
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](int a, int b) { return degree(a) < degree(b); });
    std::vector<int> nb;
    for (int cand : by_degree) {
        if (placed[cand]) continue;
        // pseudo-peripheral node of cand's component
        int s = cand, depth = -1;
        for (int it = 0; it < 4; it++) {
            int d = 0;
            const int f = bfs_far(s, d);
            if (d <= depth) break;
            depth = d;
            s = f;
        }
        // Cuthill-McKee sweep of the component
        const size_t first = order.size();
        order.push_back(s);
        placed[s] = 1;
        for (size_t head = first; head < order.size(); head++) {
            const int v = order[head];
            nb.clear();
            for (long long e = gptr[v]; e < gptr[v + 1]; e++)
                if (!placed[gadj[e]]) {
                    placed[gadj[e]] = 1;
                    nb.push_back(gadj[e]);
                }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return degree(a) != degree(b) ? degree(a) < degree(b) : a < b; });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
}

// RCM relabelling of a square matrix.  block = 4 requires exact 4x4 node-block structure (the caller checked).
inline void rcm_reorder(int n, const int* ptrow, const int* indcol, int block, Reorder& out)
{
    out = Reorder();
    out.block = block;
    const int nn = n / block;
    std::vector<long long> gptr;
    std::vector<int> gadj, order;
    build_node_graph(n, ptrow, indcol, block, gptr, gadj);
    cuthill_mckee(nn, gptr, gadj, order);
    std::vector<int> pnode((size_t)nn);
    for (int k = 0; k < nn; k++) pnode[order[k]] = nn - 1 - k; // reversed
    out.perm.resize((size_t)n);
    out.iperm.resize((size_t)n);
    for (int i = 0; i < nn * block; i++) out.perm[i] = pnode[i / block] * block + i % block;
    for (int i = nn * block; i < n; i++) out.perm[i] = i; // (block = 1 only: never happens; kept for safety)
    for (int i = 0; i < n; i++) out.iperm[out.perm[i]] = i;
    out.spread_before = mean_column_distance(n, ptrow, indcol, block);
    out.spread_after = mean_column_distance(n, ptrow, indcol, block, pnode.data());
}

// A' = P A P^T with every row's nonzeros in their ORIGINAL order.  src_start[r'] = offset of new row r' in the
// caller's arrays (for refreshing the values later without redoing any of this).
inline void permute_csr(int n, const int* ptrow, const int* indcol, const double* coef, const Reorder& R,
                        std::vector<int>& p2, std::vector<int>& c2, std::vector<double>& v2, std::vector<int>& src_start)
{
    p2.assign((size_t)n + 1, 0);
    src_start.assign((size_t)n, 0);
    for (int rn = 0; rn < n; rn++) {
        const int ro = R.iperm[rn];
        p2[rn + 1] = p2[rn] + (ptrow[ro + 1] - ptrow[ro]);
        src_start[rn] = ptrow[ro];
    }
    const size_t nnz = (size_t)ptrow[n];
    c2.resize(nnz);
    v2.resize(nnz);
    for (int rn = 0; rn < n; rn++) {
        const int a = src_start[rn], len = p2[rn + 1] - p2[rn], b = p2[rn];
        for (int k = 0; k < len; k++) {
            c2[(size_t)b + k] = R.perm[indcol[a + k]];
            v2[(size_t)b + k] = coef[a + k];
        }
    }
}

} // namespace mi355
