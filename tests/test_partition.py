"""Row-range partition planner (C++ behind mi_part_*), checked on the CPU: the
partitioned SpMV — with halos moved by hand here, by gloo in
test_dist_gloo.py — reproduces the global oracle SpMV BIT FOR BIT."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_bit_equal
from navierstokes_amd import dist as D
from navierstokes_amd import synth
from oracle import oracle as O


def build_plans(kind, n, w, nranks, balanced=True, matrix=None):
    P, C, V = matrix if matrix is not None else synth.rows(kind, n, w=w)
    rs = D.balanced_row_starts(n, nranks, np.diff(P) if balanced else None)
    plans = []
    for r in range(nranks):
        lo, hi = int(rs[r]), int(rs[r + 1])
        p = (P[lo:hi + 1] - P[lo]).astype(np.int32)
        plans.append(_Plan(rs, r, nranks, p, C[P[lo]:P[hi]], V[P[lo]:P[hi]]))
    return (P, C, V), rs, plans


class _Plan(D.DistCSR):
    """DistCSR without torch.distributed: ids are exchanged by the test itself."""

    def __init__(self, rs, rank, nranks, p, c, v):
        import ctypes
        from navierstokes_amd import mpk
        self.rank, self.nranks = rank, nranks
        self.row_starts = np.ascontiguousarray(rs, dtype=np.int64)
        L = mpk.lib()
        h = ctypes.c_void_p()
        p = np.ascontiguousarray(p, np.int32)
        c = np.ascontiguousarray(c, np.int32)
        v = np.ascontiguousarray(v, np.float64)
        mpk.check(L.mi_part_create(nranks, rank, self.row_starts.ctypes.data, p.ctypes.data, c.ctypes.data,
                                   v.ctypes.data, ctypes.byref(h)))
        self._h = h
        nl, nh, ni, nb = (ctypes.c_int() for _ in range(4))
        mpk.check(L.mi_part_sizes(h, *(ctypes.byref(t) for t in (nl, nh, ni, nb))))
        self.n_local, self.n_halo, self.n_interior, self.n_boundary = nl.value, nh.value, ni.value, nb.value
        rc = np.zeros(nranks, np.int32)
        mpk.check(L.mi_part_recv_counts(h, rc.ctypes.data))
        self.recv_counts = [int(t) for t in rc]
        self.recv_ids = []
        for q in range(nranks):
            ids = np.empty(rc[q], np.int64)
            if rc[q]:
                mpk.check(L.mi_part_recv_ids(h, q, ids.ctypes.data))
            self.recv_ids.append(ids)

    def set_send(self, peer, ids):
        from navierstokes_amd import mpk
        ids = np.ascontiguousarray(ids, np.int64)
        mpk.check(mpk.lib().mi_part_set_send_ids(self._h, peer, len(ids), ids.ctypes.data if len(ids) else None))

    def send_index(self):
        import ctypes
        from navierstokes_amd import mpk
        tot, ptr = ctypes.c_int(), ctypes.c_void_p()
        mpk.check(mpk.lib().mi_part_send_index(self._h, ctypes.byref(tot), ctypes.byref(ptr)))
        if not tot.value:
            return np.zeros(0, np.int32)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int)), shape=(tot.value,)).copy()

    def send_counts_(self):
        from navierstokes_amd import mpk
        sc = np.zeros(self.nranks, np.int32)
        mpk.check(mpk.lib().mi_part_send_counts(self._h, sc.ctypes.data))
        return sc


def _band_plus_far_couplings(n, w, seed):
    """A banded matrix (dense ghost ranges towards the neighbours) with a few far couplings (sparse ghost sets towards
    ranks that are not neighbours): both halo forms of the planner in one partition."""
    rng = np.random.default_rng(seed)
    P, C, V = synth.rows("s15", n, w=w)
    rows = [C[P[i]:P[i + 1]].astype(np.int64) for i in range(n)]
    vals = [V[P[i]:P[i + 1]] for i in range(n)]
    for i in rng.choice(n, 40, replace=False):
        far = np.setdiff1d(rng.integers(0, n, 3), rows[i])
        rows[i] = np.concatenate([rows[i], far])
        vals[i] = np.concatenate([vals[i], rng.uniform(-1, 1, len(far))])
    lens = np.array([len(r) for r in rows])
    return (np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), np.concatenate(rows).astype(np.int32), np.concatenate(vals))


@pytest.mark.parametrize("kind,n,w,nranks", [("s15", 6000, 300, 1), ("s15", 6000, 300, 2), ("svar", 5000, 200, 3),
                                              ("sfe", 4000, 240, 4), ("s15", 3000, 2000, 8), ("far", 6000, 150, 5),
                                              ("far-exact", 6000, 150, 5), ("s15-exact-split", 6000, 300, 3)])
def test_partitioned_spmv_equals_global_bitwise(kind, n, w, nranks, monkeypatch):
    matrix = None
    if kind == "s15-exact-split":  # the row-by-row split of rounds 1-4: every boundary row names a ghost column
        monkeypatch.setenv("MI355_PART_CONTIGUOUS_INTERIOR", "0")
        kind = "s15"
    if kind.startswith("far"):
        matrix = _band_plus_far_couplings(n, w, 5)
        monkeypatch.setenv("MI355_PART_DENSE_HALO", "0" if kind == "far-exact" else "1")
    (P, C, V), rs, plans = build_plans(kind, n, w, nranks, matrix=matrix)
    if kind == "far":      # dense ranges towards neighbours made the halo larger than the exact ghost set ...
        monkeypatch.setenv("MI355_PART_DENSE_HALO", "0")
        exact = build_plans(kind, n, w, nranks, matrix=matrix)[2]
        assert all(pl.n_halo >= ex.n_halo for pl, ex in zip(plans, exact))
    x = synth.x_sin(0, n)
    y_ref = O.spmv(P, C, V, x)
    # exchange the id lists by hand: what q needs from r becomes r's send list to q
    for r in range(nranks):
        for q in range(nranks):
            if q != r:
                plans[r].set_send(q, plans[q].recv_ids[r])
    for r, pl in enumerate(plans):
        lo, hi = int(rs[r]), int(rs[r + 1])
        assert pl.n_local == hi - lo and pl.n_interior + pl.n_boundary == pl.n_local
        # halo values as the peers would pack them
        halo = np.empty(pl.n_halo)
        off = 0
        for q in range(nranks):
            cnt = pl.recv_counts[q]
            if not cnt:
                continue
            sidx = plans[q].send_index()
            sc = plans[q].send_counts_()
            s_off = int(sc[:r].sum())
            qlo = int(rs[q])
            packed = x[qlo:int(rs[q + 1])][sidx[s_off:s_off + sc[r]]]
            assert sc[r] == cnt
            halo[off:off + cnt] = packed
            off += cnt
        assert off == pl.n_halo
        x_ext = np.concatenate([x[lo:hi], halo])
        y_loc = np.full(pl.n_local, np.nan)
        for which in (0, 1):
            p, c, v, rmap = pl.local_piece(which)
            assert (c < pl.n_local).all() if which == 0 else True
            if which == 1 and len(rmap):
                # (round 5) a row of the boundary piece names a ghost column — or lies outside the ONE run of consecutive ghost-free rows
                # that became the interior piece (then the interior piece's row map is a plain offset: rows r0, r0 + 1, ...)
                names = np.array([(c[p[i]:p[i + 1]] >= pl.n_local).any() for i in range(len(rmap))])
                if not names.all():
                    imap = pl.local_piece(0)[3]
                    assert len(imap) and (np.diff(imap) == 1).all(), "ghost-free rows in the boundary piece although the interior piece is scattered"
                    free = rmap[~names]
                    assert ((free < imap[0]) | (free > imap[-1])).all(), "a ghost-free row INSIDE the interior run was sent to the boundary piece"
            y_loc[rmap] = O.spmv(p, c, v, x_ext)
        assert_bit_equal(y_loc, y_ref[lo:hi], f"rank {r}/{nranks}")
        if kind == "s15" and nranks > 1 and os.environ.get("MI355_PART_CONTIGUOUS_INTERIOR") != "0" and pl.n_interior > 0:
            assert (np.diff(pl.local_piece(0)[3]) == 1).all(), "a band's interior piece must be one run of rows (a plain offset into y)"
        if nranks == 1:
            assert pl.n_halo == 0 and pl.n_boundary == 0


def test_dense_halo_ranges(monkeypatch):
    """Banded partition: the ghosts of a neighbour nearly fill a range, the planner takes the whole range so that the
    neighbour's send list is one contiguous slice of its x (sent in place, no pack kernel)."""
    sizes = {}
    for dense in ("1", "0"):
        monkeypatch.setenv("MI355_PART_DENSE_HALO", dense)
        _, rs, plans = build_plans("s15", 30000, 2000, 2)
        for r in range(2):
            plans[r].set_send(1 - r, plans[1 - r].recv_ids[r])
        sizes[dense] = [pl.n_halo for pl in plans]
        for pl in plans:
            idx = pl.send_index()
            contiguous = bool((np.diff(idx) == 1).all())
            assert contiguous == (dense == "1")
            for ids in pl.recv_ids:
                assert (np.diff(ids) > 0).all()                      # ascending, unique
                if dense == "1" and len(ids):
                    assert ids[-1] - ids[0] + 1 == len(ids)          # a full range
    assert all(d > e and d <= 1.5 * e + 64 for d, e in zip(sizes["1"], sizes["0"]))


def test_balanced_row_starts():
    lens = np.array([1, 1, 1, 1, 100, 1, 1, 1, 1, 100], np.int64)
    rs = D.balanced_row_starts(10, 2, lens)
    assert rs[0] == 0 and rs[-1] == 10 and 4 <= rs[1] <= 6
    rs = D.balanced_row_starts(1000, 8)
    assert (np.diff(rs) == 125).all()


def test_planner_rejects_bad_input():
    import ctypes
    from navierstokes_amd import mpk
    L = mpk.lib()
    h = ctypes.c_void_p()
    rs = np.array([0, 2, 4], np.int64)
    p = np.array([0, 1, 2], np.int32)
    c = np.array([0, 7], np.int32)  # column 7 >= 4
    v = np.ones(2)
    assert L.mi_part_create(2, 0, rs.ctypes.data, p.ctypes.data, c.ctypes.data, v.ctypes.data, ctypes.byref(h)) == 1
    c = np.array([0, 3], np.int32)
    assert L.mi_part_create(2, 0, rs.ctypes.data, p.ctypes.data, c.ctypes.data, v.ctypes.data, ctypes.byref(h)) == 0
    bad = np.array([3], np.int64)  # rank 0 owns rows 0..1, cannot send row 3
    assert L.mi_part_set_send_ids(h, 1, 1, bad.ctypes.data) == 1
    assert L.mi_part_finalize(h) in (2, 6)  # no GPU here / sends never set
    L.mi_part_destroy(h)


@pytest.mark.parametrize("kind,n,w,nranks", [("s15", 6000, 150, 4), ("svar", 5000, 400, 3), ("sfe", 2000, 120, 5), ("s15", 3000, 2900, 2)])
def test_combined_piece_of_the_one_launch_step(kind, n, w, nranks):
    """The fused multi-GPU step computes on ONE piece per rank: all local rows in natural order, columns numbered
    [ghosts of lower ranks | owned | ghosts of higher ranks] (partition.hpp: build_combined).  With x laid out the same way
    the oracle on that piece reproduces the global product bit for bit, the numbering is monotone in the global column id
    (a band stays a band), and the ring planner serves boundary rows too."""
    import ctypes
    from navierstokes_amd import mpk
    (P, C, V), rs, plans = build_plans(kind, n, w, nranks)
    x = synth.x_sin(0, n)
    yg = O.spmv(P, C, V, x)
    L = mpk.lib()
    for pl in plans:
        lo, hi = int(rs[pl.rank]), int(rs[pl.rank + 1])
        nr, pp, pc, pv, pm = ctypes.c_int(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        mpk.check(L.mi_part_local_csr(pl._h, 2, ctypes.byref(nr), ctypes.byref(pp), ctypes.byref(pc), ctypes.byref(pv), ctypes.byref(pm)))
        assert nr.value == pl.n_local and not pm.value
        nleft = ctypes.c_int()
        mpk.check(L.mi_part_combined_info(pl._h, ctypes.byref(nleft)))
        nleft = nleft.value
        ptrow = np.ctypeslib.as_array(ctypes.cast(pp, ctypes.POINTER(ctypes.c_int)), shape=(nr.value + 1,)).copy()
        nnz = int(ptrow[-1])
        col = np.ctypeslib.as_array(ctypes.cast(pc, ctypes.POINTER(ctypes.c_int)), shape=(max(nnz, 1),))[:nnz].copy()
        val = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_double)), shape=(max(nnz, 1),))[:nnz].copy()
        halo = np.concatenate([pl.recv_ids[q] for q in range(nranks)]) if pl.n_halo else np.zeros(0, np.int64)
        assert nleft == int((halo < lo).sum())
        gid = np.concatenate([halo[:nleft], np.arange(lo, hi), halo[nleft:]])  # global id of every combined column
        assert np.all(np.diff(gid) > 0)                                        # monotone: local bands are global bands
        assert np.array_equal(gid[col], C[P[lo]:P[hi]])                        # same terms, same order, per row
        xv = x[gid]
        assert_bit_equal(O.spmv(ptrow, col, val, xv), yg[lo:hi], f"rank {pl.rank}")
        if kind == "s15" and w < 1000 and pl.n_local > 600:  # banded: every row block of the piece fits the ring, boundary rows included
            nb_, runs, bad, frac, ms = (ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double(), ctypes.c_int())
            mpk.check(L.mi_ring_plan_probe(nr.value, ptrow.ctypes.data, col.ctypes.data, 4, ctypes.byref(nb_), ctypes.byref(runs),
                                           ctypes.byref(bad), ctypes.byref(frac), ctypes.byref(ms)))
            assert bad.value == 0 and frac.value == 1.0
