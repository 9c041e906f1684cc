"""The multi-window ring kernel's host-side planner (navierstokes_amd/csrc/mring_plan.hpp) without a GPU:
mi_mring_plan_probe builds the plan exactly as mi_csr_create does and REPLAYS it in C++ — windows filled block by block
as the kernel fills them, every nonzero's 16-bit slot must hold its column when its block runs; runs cover every block
once.  Here: what it serves on the matrix families."""
import ctypes

import numpy as np

from navierstokes_amd import mpk, synth
from test_ring_plan import relabelled


def probe(p, c):
    L = mpk.lib()
    p = np.ascontiguousarray(p, np.int32)
    c = np.ascontiguousarray(c, np.int32)
    nblk, runs, bad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    frac = ctypes.c_double()
    restarts = ctypes.c_longlong()
    mpk.check(L.mi_mring_plan_probe(len(p) - 1, p.ctypes.data, c.ctypes.data, ctypes.byref(nblk), ctypes.byref(runs),
                                    ctypes.byref(bad), ctypes.byref(frac), ctypes.byref(restarts)))
    return nblk.value, runs.value, bad.value, frac.value, restarts.value


def test_mesh_operator_is_served_in_natural_and_in_relabelled_order():
    """3-D mesh, scalar P1 operator: three column clusters two mesh planes apart.  The single ring serves none of it
    (test_ring_plan), the five small windows all of it — in the mesher-free natural order and after scramble + relabel."""
    p, c, v = synth.pressure_matrix(48, 44, 40)
    nblk, runs, bad, frac, restarts = probe(p, c)
    assert bad == 0 and frac == 1.0 and restarts <= nblk // 100 and 0 < runs <= 512   # (a handful of blocks at the mesh faces need > 8 groups; one round of workgroups)
    ps, cs, _ = synth.permute_nodes(p, c, v, block=1)[:3]
    assert probe(ps, cs)[3] < 0.05                                      # scrambled: nothing to hold on to
    p2, c2 = relabelled(np.ascontiguousarray(ps, np.int32), np.ascontiguousarray(cs, np.int32))
    nblk, runs, bad, frac, restarts = probe(p2, c2)
    assert bad == 0 and frac > 0.97                                     # level transitions: a few blocks with too many clusters


def test_bands_and_degenerate_shapes():
    p, c, _ = synth.rows("s15", 60_000, w=300)                         # a band narrower than one window: one cluster
    assert probe(p, c)[2:4] == (0, 1.0)
    p, c, _ = synth.rows("s15", 60_000, w=2000)                        # wider than a window: the single ring's case, not this one's
    assert probe(p, c)[3] < 0.05
    p, c, _ = synth.fe_matrix(10)                                      # 4 dofs per node: same clusters, four times as wide
    assert probe(p, c)[3] == 1.0
    assert probe(np.zeros(1, np.int32), np.zeros(0, np.int32))[0] == 0
    assert probe(np.zeros(9, np.int32), np.zeros(0, np.int32))[3] == 0.0
    # empty rows, a row longer than a block (PLAIN), a short tail
    p = np.array([0, 0, 0, 5000, 5000, 5003], np.int32)
    c = np.concatenate([np.arange(5000) % 4000, [1, 0, 2]]).astype(np.int32)
    nblk, runs, bad, frac, _ = probe(p, c)
    assert nblk == 3 and bad == 0 and 0 < frac < 0.01


def test_windows_restart_and_swap_roles():
    """clusters that jump: every block brings two clusters at new places (no window continues) — all windows restart per
    block and the replay still finds every column in its slot; six clusters in a block: PLAIN"""
    n, per = 6000, 15
    rows = np.arange(n)
    blk = rows // 128                                                   # blocks end on multiples of 64 rows: 128 x 15 nonzeros
    a = (blk * 7919) % 50 * 3000                                        # two clusters per block, wandering
    b = a + 150_000
    c = np.where(np.arange(per)[None, :] < 8, a[:, None], b[:, None]) + (rows[:, None] * 13 + np.arange(per)[None, :] * 17) % 600
    p = (np.arange(n + 1) * per).astype(np.int32)
    nblk, runs, bad, frac, restarts = probe(p, c.astype(np.int32).ravel())
    assert bad == 0 and frac == 1.0
    six = (np.arange(per)[None, :] % 6) * 40_000 + (rows[:, None] % 500)
    nblk, runs, bad, frac, _ = probe(p, six.astype(np.int32).ravel())
    assert frac == 0.0


def deal(p, c):
    L = mpk.lib()
    p = np.ascontiguousarray(p, np.int32)
    c = np.ascontiguousarray(c, np.int32)
    tl = ctypes.c_int()
    blocks = np.zeros(8192, np.int32)
    mpk.check(L.mi_mring_plan_deal_probe(len(p) - 1, p.ctypes.data, c.ctypes.data, ctypes.byref(tl), blocks.ctypes.data, len(blocks)))
    assert tl.value % 8 == 0 and tl.value <= len(blocks)
    return blocks[: tl.value].reshape(8, -1)


def test_runs_are_dealt_for_the_dispatch_order_measured_on_the_hardware():
    """mring_plan.hpp: an XCD takes its workgroups in order, 64 resident at a time, and a late-comer waits for a slot of its own
    shader engine — so every XCD's LONG runs must be among its first 64 workgroups (the short ones behind them), the long runs
    must be few enough for that, and nearly equal (a forced cut leaves no stub beside a full-length run).  Relabelled mesh: the
    prologue's lead keeps the forced cuts to a handful."""
    p, c, v = synth.pressure_matrix(100, 96, 92)
    ps, cs, _ = synth.permute_nodes(p, c, v, block=1)[:3]
    p2, c2 = relabelled(np.ascontiguousarray(ps, np.int32), np.ascontiguousarray(cs, np.int32))
    for pp, cc, tag in ((p, c, "natural"), (p2, c2, "relabelled")):
        nblk, runs, bad, frac, restarts = probe(pp, cc)
        T = deal(pp, cc)
        assert int(T.sum()) == nblk and int((T > 0).sum()) == runs, tag
        long_ = T > min(5, -(-nblk // 512) // 8)
        assert long_.sum() <= 512 and (long_[:, 64:].sum() == 0), (tag, long_.sum(axis=1))      # all long runs in the first round
        assert T.max() <= 96, tag
        assert restarts <= max(4, nblk // 400), (tag, restarts, nblk)
        lens = T[long_]
        assert lens.max() <= 1.25 * np.median(lens) + 2, (tag, int(lens.max()), float(np.median(lens)))
        # workgroups j and j + 32 of an XCD share a CU: the sums are balanced (longest beside shortest)
        if T.shape[1] >= 64:
            pair = T[:, :32] + T[:, 32:64]
            assert pair.max() <= 1.2 * np.median(pair) + 4, (tag, pair.max(axis=1))
