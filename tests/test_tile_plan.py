"""The tile kernel's host-side planner (navierstokes_amd/csrc/tile_plan.hpp) without a GPU: mi_tile_plan_probe builds
the per-block lists of distinct columns and the 16-bit slot stream exactly as mi_csr_create does and checks their
invariants in C++ (every slot names its nonzero's column, lists strictly ascending, block boundaries on rows, 16-byte
aligned slot segments, over-long rows unlisted); here: its sizes against a numpy count on the matrix families."""
import ctypes

import numpy as np
import pytest

from navierstokes_amd import mpk, synth

NNZB = 2048


def probe(p, c, threads=0):
    L = mpk.lib()
    p = np.ascontiguousarray(p, np.int32)
    c = np.ascontiguousarray(c, np.int32)
    nblk, mx = ctypes.c_int(), ctypes.c_int()
    tot, listed = ctypes.c_longlong(), ctypes.c_longlong()
    mpk.check(L.mi_tile_plan_probe(len(p) - 1, p.ctypes.data, c.ctypes.data, threads, ctypes.byref(nblk), ctypes.byref(tot),
                                   ctypes.byref(mx), ctypes.byref(listed)))
    return nblk.value, tot.value, mx.value, listed.value


def numpy_count(p, c, max_rows=1024):
    """blocks of whole rows with <= NNZB nonzeros (at least one row, at most max_rows), distinct columns of each"""
    n = len(p) - 1
    r, nblk, tot, mx, listed = 0, 0, 0, 0, 0
    while r < n:
        e = r + 1
        while e < n and e - r < max_rows and p[e + 1] - p[r] <= NNZB:
            e += 1
        ea = e // 64 * 64                                   # whole waves of rows where that keeps 7/8 of the block
        if e < n and ea > r and 8 * (p[ea] - p[r]) >= 7 * (p[e] - p[r]):
            e = ea
        nn = int(p[e] - p[r])
        if 0 < nn <= NNZB:
            u = len(np.unique(c[p[r]:p[e]]))
            tot, mx, listed = tot + u, max(mx, u), listed + nn
        nblk += 1
        r = e
    return nblk, tot, mx, listed


@pytest.mark.parametrize("kind,n,w", [("s15", 40_000, 2000), ("svar", 30_000, 500), ("sfe", 12_000, 400), ("s15", 9_001, 30_000)])
@pytest.mark.parametrize("threads", [1, 3])
def test_plan_matches_a_numpy_count(kind, n, w, threads):
    p, c, _ = synth.rows(kind, n, w=w)
    assert probe(p, c, threads) == numpy_count(p, c)


def test_mesh_rows_share_columns_and_a_random_numbering_does_not():
    p, c, v = synth.fe_matrix(10)                                      # 3-D Kuhn mesh, 4 dofs per node, natural order
    nblk, tot, mx, listed = probe(p, c)
    assert listed == p[-1] and tot < 0.2 * p[-1]                       # four rows of a node alone share every column
    ps, cs, _ = synth.permute_nodes(p, c, v, block=4)[:3]
    assert probe(ps, cs)[1] > 1.5 * tot                                # scrambled: neighbours in the numbering are strangers


def test_degenerate_shapes():
    assert probe(np.zeros(1, np.int32), np.zeros(0, np.int32)) == (0, 0, 0, 0)
    assert probe(np.zeros(9, np.int32), np.zeros(0, np.int32)) == (1, 0, 0, 0)       # one block of empty rows
    # empty rows, a row longer than a block (not listed), a short tail
    p = np.array([0, 0, 0, 5000, 5000, 5003], np.int32)
    c = np.concatenate([np.arange(5000) % 4000, [1, 0, 2]]).astype(np.int32)
    assert probe(p, c) == (3, 3, 3, 3)
    # 3000 one-entry rows on one column: blocks are cut at 1024 rows
    p = np.arange(3001, dtype=np.int32)
    c = np.full(3000, 7, np.int32)
    assert probe(p, c) == (3, 3, 1, 3000)
    # exactly NNZB distinct columns in one block: the 16-bit slots reach NNZB - 1
    p = np.array([0, NNZB], np.int32)
    c = np.arange(NNZB, dtype=np.int32)[::-1].copy()
    assert probe(p, c) == (1, NNZB, NNZB, NNZB)
