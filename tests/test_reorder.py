"""Create-time locality reordering (navierstokes_amd/csrc/reorder.hpp): the host-side planner on CPU; the
bitwise GPU checks are in test_gpu_parity.py::test_permuted_*."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

from navierstokes_amd import mpk, synth


def probe(p, c):
    n = len(p) - 1
    L = mpk.lib()
    blk = ctypes.c_int()
    perm = np.empty(n, np.int32)
    sb, sa = ctypes.c_double(), ctypes.c_double()
    mpk.check(L.mi_reorder_probe(n, np.ascontiguousarray(p, np.int32).ctypes.data, np.ascontiguousarray(c, np.int32).ctypes.data,
                                 ctypes.byref(blk), perm.ctypes.data, ctypes.byref(sb), ctypes.byref(sa)))
    return blk.value, perm, sb.value, sa.value


def bandwidth(p, c, perm):
    n = len(p) - 1
    rows = np.repeat(np.arange(n), np.diff(p))
    return int(np.abs(perm[rows] - perm[c]).max())


def test_rcm_recovers_a_scrambled_fe_mesh():
    """An FE matrix under a random NODE numbering (what a mesher delivers): RCM brings the columns back near the diagonal
    and moves whole nodes, so the 4x4 block structure survives."""
    p0, c0, v0 = synth.fe_matrix(10)
    n = len(p0) - 1
    p, c, v, _ = synth.permute_nodes(p0, c0, v0, block=4, seed=3)
    blk, perm, sb, sa = probe(p, c)
    assert blk == 4
    assert sorted(perm) == list(range(n))
    assert np.array_equal(perm[0::4] % 4, np.zeros(n // 4)) and np.array_equal(perm[1::4], perm[0::4] + 1) and np.array_equal(perm[3::4], perm[0::4] + 3)
    nat_blk, _, nat_sb, _ = probe(p0, c0)
    assert nat_blk == 4
    assert sa < 0.2 * sb                 # scrambled: mean distance ~ n/3 nodes; RCM: a few mesh layers
    assert sa < 3.0 * nat_sb             # as good as the generator's natural (lexicographic) numbering, within a small factor
    assert bandwidth(p, c, perm) < 0.25 * bandwidth(p, c, np.arange(n))
    # the relabelled matrix is the same operator: (P A P^T)(P x) = P (A x)
    A = sp.csr_matrix((v, c, p), shape=(n, n))
    Ap = sp.csr_matrix((v, perm[c], p), shape=(n, n))[np.argsort(perm)]
    x = np.sin(np.arange(n))
    xp = np.empty(n)
    xp[perm] = x
    assert np.allclose((Ap @ xp)[perm], A @ x, rtol=1e-13, atol=1e-13)


def test_rcm_on_scalar_rows_and_disconnected_graphs():
    p0, c0, v0 = synth.rows("s15", 6000, w=40)
    p, c, v, _ = synth.permute_nodes(p0, c0, v0, block=1, seed=5)
    blk, perm, sb, sa = probe(p, c)
    assert blk == 1 and sorted(perm) == list(range(6000))
    assert sa < 0.1 * sb and bandwidth(p, c, perm) < 600   # half-bandwidth 40 scrambled over 6000 rows, recovered to a few levels
    # two components + isolated rows + an empty row
    p = np.array([0, 2, 4, 5, 5, 7, 9, 10], np.int32)
    c = np.array([0, 1, 0, 1, 2, 4, 5, 4, 5, 6], np.int32)
    blk, perm, _, _ = probe(p, c)
    assert blk == 1 and sorted(perm) == list(range(7))
    # degenerate sizes
    for n in (0, 1):
        pp = np.zeros(n + 1, np.int32)
        _, perm, _, _ = probe(pp, np.zeros(0, np.int32))
        assert sorted(perm) == list(range(n))


def test_natural_fe_numbering_is_left_alone_by_the_threshold():
    """mi_csr_create only tries RCM when the mean distance exceeds 2 nn^(2/3): the generator's lexicographic numbering stays below."""
    p, c, _ = synth.fe_matrix(16)
    n = len(p) - 1
    _, _, sb, _ = probe(p, c)
    assert sb < 2.0 * (n / 4) ** (2.0 / 3.0)
    ps, cs, _, _ = synth.permute_nodes(p, c, np.zeros(len(c)), block=4, seed=1)
    _, _, sbs, _ = probe(ps, cs)
    assert sbs > 2.0 * (n / 4) ** (2.0 / 3.0)
