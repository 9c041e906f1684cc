"""Seeded fuzz of the three host-side planners (ring, multi-window ring, tile): whatever the pattern, the plan must REPLAY —
the probes rebuild each plan exactly as mi_csr_create does and check, slot by slot, what the kernels will find in LDS when a
block runs (and that no record sends a kernel outside its arrays).  A plan that fails here would be a wrong result or a GPU
fault on the box, so this runs on the CPU, every round."""
import numpy as np
import pytest

from test_mring_plan import probe as mring_probe
from test_ring_plan import is_lean, probe as ring_probe
from test_tile_plan import probe as tile_probe


def random_pattern(rng, n):
    """rows of wildly different lengths (empty, short, one longer than a block), columns in 1-6 clusters that wander, jump or sit
    anywhere; sorted or not; duplicates allowed"""
    kind = rng.integers(0, 5)
    lens = rng.integers(0, [4, 20, 40, 9, 70][kind], n)
    if rng.random() < 0.3:
        lens[rng.integers(0, n, max(1, n // 50))] = 0
    if rng.random() < 0.3:
        lens[rng.integers(0, n)] = int(rng.integers(2049, 5000))
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ncl = int(rng.integers(1, 7))
    width = int(rng.choice([8, 100, 700, 1500, 6000]))
    centres = rng.integers(0, n, ncl)
    speed = rng.choice([0.0, 1.0, 1.0, 3.0], ncl)
    rows = np.repeat(np.arange(n), lens)
    which = rng.integers(0, ncl, len(rows))
    c = (centres[which] + (speed[which] * rows).astype(np.int64) + rng.integers(-width, width + 1, len(rows))) % n
    if rng.random() < 0.2:  # a few entries anywhere
        far = rng.random(len(rows)) < 0.01
        c[far] = rng.integers(0, n, int(far.sum()))
    c = c.astype(np.int32)
    if rng.random() < 0.5:  # sorted rows (what COO2CSR delivers)
        order = np.lexsort((c, rows))
        c = c[order]
    return p, c


@pytest.mark.parametrize("seed", range(24))
def test_plans_replay_on_random_patterns(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 7, 300, 5000, 20000, 60000]))
    p, c = random_pattern(rng, n)
    for cfg in (1, 2, 3, 4):
        nblk, runs, bad, frac, mslot = ring_probe(p, c, cfg)     # raises on the first violated invariant
        assert 0.0 <= frac <= 1.0 and runs % 8 == 0
    is_lean(p, c)
    nblk, runs, bad, frac, restarts = mring_probe(p, c)
    assert 0.0 <= frac <= 1.0 and (runs >= 1 or nblk == 0)
    nblk, tot, mx, listed = tile_probe(p, c, threads=int(rng.integers(1, 4)))
    assert listed <= p[-1] and mx <= 2048


@pytest.mark.parametrize("seed", range(24))
def test_powers_step_dependencies_on_random_patterns(seed):
    """mi_spmk_plan_probe on the same random patterns: wherever the one-launch powers step is eligible, every column a run names
    or loads belongs to a run on its dependency list (the probe raises on the first violation)."""
    import ctypes
    from navierstokes_amd import mpk
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([7, 300, 5000, 20000, 60000, 200000]))
    p, c = random_pattern(rng, n)
    p = np.ascontiguousarray(p, np.int32)
    c = np.ascontiguousarray(c, np.int32)
    e, r, m = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    mpk.check(mpk.lib().mi_spmk_plan_probe(n, p.ctypes.data, c.ctypes.data, ctypes.byref(e), ctypes.byref(r), ctypes.byref(m)))
    assert e.value in (0, 1) and (e.value == 0 or (r.value >= 8 and 1 <= m.value <= 64))
