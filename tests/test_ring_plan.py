"""The ring kernel's host-side planner (navierstokes_amd/csrc/ring_plan.hpp) without a GPU: mi_ring_plan_probe
builds the window plan and the 16-bit column stream exactly as mi_csr_create does and checks their invariants
in C++ (rows / nonzeros covered once and in order, every served block inside one window of <= RING columns,
slots agree with the columns and are < RING); here: what each configuration serves on the benchmark families."""
import ctypes

import numpy as np
import pytest

from navierstokes_amd import mpk, synth


def probe(p, c, cfg):
    L = mpk.lib()
    p = np.ascontiguousarray(p, np.int32)
    c = np.ascontiguousarray(c, np.int32)
    nblk, runs, bad, mslot = (ctypes.c_int() for _ in range(4))
    frac = ctypes.c_double()
    mpk.check(L.mi_ring_plan_probe(len(p) - 1, p.ctypes.data, c.ctypes.data, cfg, ctypes.byref(nblk), ctypes.byref(runs),
                                   ctypes.byref(bad), ctypes.byref(frac), ctypes.byref(mslot)))
    return nblk.value, runs.value, bad.value, frac.value, mslot.value


def is_lean(p, c, cfg=4):
    L = mpk.lib()
    p = np.ascontiguousarray(p, np.int32)
    c = np.ascontiguousarray(c, np.int32)
    out = ctypes.c_int()
    mpk.check(L.mi_ring_plan_lean(len(p) - 1, p.ctypes.data, c.ctypes.data, cfg, ctypes.byref(out)))
    return bool(out.value)


RING = {1: 5120, 2: 5120, 3: 11264, 4: 5120}
NNZB = {1: 2048, 2: 4096, 3: 4096, 4: 2048}


@pytest.mark.parametrize("cfg", [1, 2, 3, 4])
@pytest.mark.parametrize("kind,n,w", [("s15", 60_000, 2000), ("svar", 50_000, 2000), ("s15", 30_001, 300)])
def test_banded_families_are_fully_served(cfg, kind, n, w):
    p, c, _ = synth.rows(kind, n, w=w)
    nblk, runs, bad, frac, mslot = probe(p, c, cfg)
    assert bad == 0 and frac == 1.0
    assert 0 <= mslot < RING[cfg]
    assert nblk >= int(p[-1]) // NNZB[cfg] and runs % 8 == 0      # runs are dealt to the 8 XCDs


def test_wider_band_needs_the_wide_ring():
    p, c, _ = synth.rows("s15", 60_000, w=4500)                   # window ~9000 columns
    assert probe(p, c, 4)[3] < 0.05 and probe(p, c, 2)[3] < 0.05  # 5120-entry rings hold only the truncated corners
    assert probe(p, c, 3)[2:4] == (0, 1.0)                        # the 11264-entry ring can


def test_window_wider_than_any_ring_is_left_to_the_stream_kernel():
    p, c, _ = synth.rows("s15", 80_000, w=30_000)
    for cfg in (1, 2, 3, 4):
        assert probe(p, c, cfg)[3] < 0.6   # only the truncated corners of the band fit a window


def test_fe_matrix_and_degenerate_shapes():
    p, c, _ = synth.fe_matrix(6)                                   # 3-D mesh in natural order: partly served at best
    for cfg in (1, 2, 3, 4):
        nblk, runs, bad, frac, mslot = probe(p, c, cfg)
        assert 0.0 <= frac <= 1.0 and mslot < RING[cfg]
    # empty matrix, all-empty rows, one dense row longer than a block, a single entry
    assert probe(np.zeros(1, np.int32), np.zeros(0, np.int32), 4)[0] == 0
    assert probe(np.zeros(9, np.int32), np.zeros(0, np.int32), 4)[3] == 0.0
    n = 3000
    lens = np.ones(n, np.int64)
    lens[17] = 2500                                                # longer than NNZB of configuration 4
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    c = np.concatenate([np.arange(l) if l > 1 else [i] for i, l in enumerate(lens)]).astype(np.int32)
    nblk, runs, bad, frac, _ = probe(p, c, 4)
    assert bad == 0 and frac < 1.0                                 # the long row is a PLAIN block behind its run's loop, not a plain run
    assert probe(np.array([0, 1], np.int32), np.array([0], np.int32), 4)[2:4] == (0, 1.0)


@pytest.mark.parametrize("seed", range(12))
def test_plan_invariants_hold_on_random_matrices(seed):
    """Whatever the matrix — random row lengths (with empty and over-long rows), random bands, columns unsorted and
    repeated, rectangular — the plan and the 16-bit stream pass the C++ invariant checks for every configuration."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 6000))
    ncols = int(rng.integers(1, 9000))
    band = int(rng.choice([5, 200, 3000, 20000]))
    lens = rng.integers(0, int(rng.choice([4, 20, 70])), n)
    if n > 3:
        lens[rng.integers(0, n, 2)] = rng.integers(2000, 4500, 2)
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    centre = np.linspace(0, ncols - 1, n)
    c = np.concatenate([np.clip(rng.integers(-band, band + 1, l) + int(centre[i]), 0, ncols - 1) for i, l in enumerate(lens)]
                       + [np.zeros(0, np.int64)]).astype(np.int32)
    for cfg in (1, 2, 3, 4):
        nblk, runs, bad, frac, mslot = probe(p, c, cfg)
        assert 0.0 <= frac <= 1.0 and mslot < RING[cfg]


def relabelled(p, c):
    """pattern of P A P^T for the library's own relabelling (mi_reorder_probe), each row's terms in their original order"""
    L = mpk.lib()
    n = len(p) - 1
    perm = np.zeros(n, np.int32)
    blk = ctypes.c_int()
    sb, sa = ctypes.c_double(), ctypes.c_double()
    mpk.check(L.mi_reorder_probe(n, p.ctypes.data, c.ctypes.data, ctypes.byref(blk), perm.ctypes.data, ctypes.byref(sb), ctypes.byref(sa)))
    iperm = np.empty(n, np.int64)
    iperm[perm] = np.arange(n)
    lens = np.diff(p)[iperm]
    p2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.repeat(p[:-1][iperm].astype(np.int64) - p2[:-1], lens) + np.arange(p2[-1])
    return p2, perm[c[idx]].astype(np.int32)


def test_relabelled_band_keeps_every_run_in_the_ring_loop():
    """A scrambled band after the create-time relabelling: where the Cuthill-McKee levels start, ~2 % of the rows reach over
    more columns than the ring holds.  Their blocks are PLAIN blocks behind the loops of runs dealt out by weight — no run
    goes down the plain path as a whole (all runs start together: one slow run is the launch's tail), and where the window
    cannot keep its history it restarts instead of giving up."""
    p, c, v = synth.rows("s15", 300_000, w=2000)
    ps, cs, _ = synth.permute_nodes(p, c, v, block=1)[:3]
    assert probe(ps, cs, 4)[3] < 0.05                      # scrambled: hopeless
    p2, c2 = relabelled(np.ascontiguousarray(ps, np.int32), np.ascontiguousarray(cs, np.int32))
    nblk, runs, bad, frac, mslot = probe(p2, c2, 4)    # the probe also fails if a run carries > 2x the mean weight
    assert bad == 0 and 0.97 < frac < 1.0
    assert is_lean(p2, c2) and is_lean(p, c)           # runs cut at the window restarts: the LEAN kernel runs both
    assert nblk <= int(p2[-1]) // 1920 + 1 + nblk // 50    # only a few extra cuts (blocks end on multiples of 64 rows: 128 x 15 nonzeros)


def test_window_restarts_instead_of_giving_up():
    """blocks whose columns alternate between the two ends of a span wider than the ring: every block fits a window of its
    own, no window holds two consecutive ones — the plan restarts per block and still serves all of them"""
    n, per = 4000, 15
    rows = np.arange(n)
    base = np.where((rows // 128) % 2 == 0, 0, 6000)       # 128 rows of 15 = one block (blocks end on multiples of 64 rows)
    c = (base[:, None] + (rows[:, None] * 7 + np.arange(per)[None, :] * 131) % 3000).astype(np.int32).ravel()
    p = (np.arange(n + 1) * per).astype(np.int32)
    nblk, runs, bad, frac, mslot = probe(p, c, 4)
    assert bad == 0 and frac == 1.0


def test_lean_plans():
    """LEAN = no block inside a run brings > T new columns or holds > T rows (the replay in mi_ring_plan_probe checks the
    invariants of a plan that claims it).  Bands are; a matrix of very short rows is (blocks are cut at T rows); the
    alternating-ends matrix restarts its window at every block and still is — every block starts its own run or the
    planner gives up and says so."""
    p, c, _ = synth.rows("s15", 60_000, w=2000)
    assert is_lean(p, c) and not is_lean(p, c, 1)          # only configuration 4 has the instantiation
    p, c, _ = synth.rows("svar", 50_000, w=2000)
    assert is_lean(p, c)
    n = 40_000
    p = np.arange(n + 1, dtype=np.int32) * 2                # two entries per row: blocks of 256 rows, not 1024
    c = np.stack([np.arange(n), np.minimum(np.arange(n) + 3, n - 1)], 1).astype(np.int32).ravel()
    assert is_lean(p, c) and probe(p, c, 4)[2:4] == (0, 1.0)


def test_one_launch_powers_step_dependencies_cover_every_load():
    """spmk_ring.hpp's hand-off is safe without an acquire only if no load of a run ever touches a column whose owner is not on
    the run's dependency list (ring_plan.hpp: build_run_deps).  mi_spmk_plan_probe derives the lists as mi_csr_create does and
    replays every named and every loaded column against them — bands narrower and wider than a run, variable row lengths, empty
    rows and whole empty blocks, a relabelled band."""
    import ctypes
    L = mpk.lib()

    def probe(p, c, n):
        p = np.ascontiguousarray(p, np.int32)
        c = np.ascontiguousarray(c, np.int32)
        e, r, m = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        mpk.check(L.mi_spmk_plan_probe(n, p.ctypes.data, c.ctypes.data, ctypes.byref(e), ctypes.byref(r), ctypes.byref(m)))
        return e.value, r.value, m.value

    for kind, n, w in (("s15", 300_000, 2000), ("svar", 200_000, 2000), ("s15", 1_000_000, 2000), ("s15", 60_000, 300), ("s15", 400_000, 2400)):
        p, c, v = synth.rows(kind, n, w=w)
        e, runs, md = probe(p, c, n)
        assert e == 1 and runs >= 8 and 1 <= md <= 64, (kind, n, w, e, runs, md)
    p, c, v = synth.rows("s15", 400_000, w=4000)  # rows span more columns than configuration 4's window holds: k launches
    assert probe(p, c, 400_000)[0] == 0
    # rows without nonzeros, and a stretch of 3000 empty rows (whole empty blocks inside a run)
    p, c, v = synth.rows("s15", 200_000)
    lens = np.diff(p).astype(np.int64)
    keep = np.ones(len(c), bool)
    rows = np.repeat(np.arange(200_000), lens)
    keep[(rows % 7 == 3) | ((rows >= 50_000) & (rows < 53_000))] = False
    cnt = np.zeros(200_000, np.int64)
    np.add.at(cnt, rows[keep], 1)
    p2 = np.concatenate([[0], np.cumsum(cnt)])
    e, runs, md = probe(p2, c[keep], 200_000)
    assert e == 1 and md <= 64, (e, runs, md)
    # a band much wider than a run: still checked, not eligible
    p, c, v = synth.rows("s15", 200_000, w=60_000)
    e, runs, md = probe(p, c, 200_000)
    assert e == 0
    # narrow bands: a run's rows plus band span fewer columns than the window's first fill (5120) — the kernel clamps
    # that fill to the first block's new columns (spmk_ring.hpp), which is what the probe's replay of the loads assumes
    for n, hb in ((1_000_000, 1), (300_000, 3), (120_000, 40)):
        i = np.arange(n, dtype=np.int64)
        cols = np.stack([i + d for d in range(-hb, hb + 1)], axis=1)
        ok = (cols >= 0) & (cols < n)
        p = np.concatenate([[0], np.cumsum(ok.sum(axis=1))])
        e, runs, md = probe(p, cols[ok], n)
        assert e == 1 and md <= 64, (n, hb, e, runs, md)


def test_cut_ring_sliced_stream_plan_is_replayed_on_the_host():
    """Round 5: the sliced stream's cut-ring form (spmv_sstream_mw.hpp) for rows that name several column neighbourhoods — a 3-D mesh
    operator in natural node order names the node's plane and the two next to it.  mi_sstream_mw_plan_probe builds the plan as mi_csr_create
    does when the one-window plan is not eligible and replays it with the cut ring simulated: every nonzero must find ITS column at its slot
    when its round runs.  Eligible: the P1 pressure operator on Kuhn meshes (also planned one row down), two far-apart bands; not eligible
    (and saying why): a band wider than a sub-ring, five neighbourhoods per row."""
    import ctypes
    L = mpk.lib()

    def probe(p, c, n, shift=0):
        p = np.ascontiguousarray(p, np.int32)
        c = np.ascontiguousarray(c, np.int32)
        e, r, st, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_double()
        mpk.check(L.mi_sstream_mw_plan_probe(n, n, p.ctypes.data, c.ctypes.data, shift, ctypes.byref(e), ctypes.byref(r), ctypes.byref(st), ctypes.byref(pad)))
        return e.value, r.value, st.value, pad.value, L.mi_last_error().decode()

    for cells in (20, 33, 70):
        p, c, v = synth.pressure_matrix(cells)
        n = len(p) - 1
        for shift in (0, 1):
            e, rounds, steps, pad, why = probe(p, c, n, shift)
            assert e == 1 and rounds == (n + shift + 511) // 512 and pad < 0.12, (cells, shift, e, rounds, pad, why)
        if cells == 70:  # three planes 5041 columns apart + a round's 512 rows: more than the one-window ring holds
            e1, r1, s1, p1 = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_double()
            mpk.check(L.mi_sstream_plan_probe(n, n, p.ctypes.data, c.ctypes.data, ctypes.byref(e1), ctypes.byref(r1), ctypes.byref(s1), ctypes.byref(p1)))
            assert e1.value == 0

    def bands(n, k, gap, width=5):  # k bands of `width` columns, `gap` apart: rows name k neighbourhoods
        offs = np.concatenate([np.arange(width) - width // 2 + (j - k // 2) * gap for j in range(k)])
        cols = np.arange(n)[:, None] + offs[None, :]
        keep = (cols >= 0) & (cols < n)
        p = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int32)
        return p, cols[keep].astype(np.int32)
    p, c = bands(60_000, 3, 9000)
    assert probe(p, c, 60_000)[0] == 1
    p, c = bands(60_000, 4, 5000)
    assert probe(p, c, 60_000)[0] == 1
    p, c = bands(60_000, 5, 5000)
    e, _, _, _, why = probe(p, c, 60_000)
    assert e == 0 and "four" in why, why
    p, c, v = synth.rows("s15", 100_000, w=2000)  # one neighbourhood of 4000 + 512 columns: wider than a sub-ring
    e, _, _, _, why = probe(p, c, 100_000)
    assert e == 0 and "sub-ring" in why, why


def test_sliced_stream_plan_is_replayed_on_the_host():
    """mi_sstream_plan_probe builds the sliced-stream kernel's plan (spmv_sstream.hpp) as mi_csr_create would and replays it: every
    nonzero's slot is its column's ring slot, every column lies inside the sliding window when its round runs, padding places and slice
    beginnings are flagged.  Eligible: S15 bands; not eligible (and saying why): a band wider than the LDS ring, rows whose lengths vary
    too much inside a slice."""
    import ctypes
    L = mpk.lib()

    def probe(p, c, n, ncols=None):
        p = np.ascontiguousarray(p, np.int32)
        c = np.ascontiguousarray(c, np.int32)
        e, r, st, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_double()
        mpk.check(L.mi_sstream_plan_probe(n, n if ncols is None else ncols, p.ctypes.data, c.ctypes.data, ctypes.byref(e), ctypes.byref(r), ctypes.byref(st), ctypes.byref(pad)))
        return e.value, r.value, st.value, pad.value, L.mi_last_error().decode()

    for n, w in ((300_000, 2000), (1_000_000, 2000), (70_001, 900), (1_000, 20), (600_000, 3400)):
        p, c, v = synth.rows("s15", n, w=w)
        e, rounds, steps, pad, _ = probe(p, c, n)
        assert e == 1 and rounds == (n + 511) // 512 and pad < 0.03 and steps >= 15 * ((n + 127) // 128), (n, w, e, rounds, steps, pad)
    p, c, v = synth.rows("s15", 200_000, w=6000)  # 12 000 columns of span: more than the ring's 8192
    e, _, _, _, why = probe(p, c, 200_000)
    assert e == 0 and "ring" in why, why
    p, c, v = synth.rows("svar", 200_000, w=2000)  # 8..22 nonzeros per row: a slice pads to its longest row
    e, _, _, pad, why = probe(p, c, 200_000)
    assert e == 0 and "padding" in why and pad > 0.12, (pad, why)


def test_sliced_stream_plan_with_a_row_shift_and_with_ghost_columns(monkeypatch):
    """Round 5: the two forms a partition's pieces take (mi_sstream_plan_probe_ex builds them as mi_csr_create_mapped / the fused step's
    combined piece would and replays them on the host).  shift = 1: the rows are planned one down (view row 0 does not exist) so that a
    piece whose rows go to y[r + odd offset] keeps 16-byte row pairs.  Ghost columns ([lower ghosts | owned | upper ghosts] numbering of
    partition.hpp: build_combined): every workgroup that names one is marked, the marked ones get a round less than their share, and the
    shares differ by at most two rounds."""
    import ctypes
    L = mpk.lib()
    monkeypatch.setenv("MI355_SSTREAM_MAX_PADDING", "1e9")  # (a few hundred rows: the one row a shift pushes into a slice of its own is all padding)

    def probe(n, ncols, p, c, shift=0, glo=0, ghi=0):
        e, r, gw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        st, pad = ctypes.c_longlong(), ctypes.c_double()
        mm = (ctypes.c_int * 2)()
        mpk.check(L.mi_sstream_plan_probe_ex(n, ncols, p.ctypes.data, c.ctypes.data, shift, glo, ghi, ctypes.byref(e), ctypes.byref(r), ctypes.byref(st),
                                             ctypes.byref(pad), ctypes.byref(gw), mm))
        return e.value, r.value, st.value, pad.value, gw.value, (mm[0], mm[1])

    for n, w in ((70_001, 900), (3_000, 300), (1_001, 20), (511, 30), (512, 30), (513, 30), (300_000, 2000)):
        p, c, _ = synth.rows("s15", n, w=w)
        e0, r0, st0, pad0, _, _ = probe(n, n, p, c)
        e1, r1, st1, pad1, _, _ = probe(n, n, p, c, shift=1)
        assert e0 == 1 and e1 == 1, (n, w)
        assert r1 == (n + 1 + 511) // 512 and r0 == (n + 511) // 512
        assert st1 >= st0 - 15 and st1 <= st0 + 15 * 4, (st0, st1)  # the same rows, at most one slice more
    # one rank's share of a band: rows [lo, hi) of a global band, columns renumbered [lower ghosts | owned | upper ghosts]
    for nglob, lo, hi, w in ((400_000, 100_000, 200_000, 2000), (400_000, 0, 150_000, 2000), (400_000, 250_000, 400_000, 2000), (60_000, 20_000, 40_000, 300),
                             (5_000_000, 2_500_000, 5_000_000, 2000)):  # (the last: half of C4 — twenty rounds per workgroup, a ghost reader's columns capped to the ring)
        p, c, _ = synth.rows("s15", nglob, lo, hi, w=w)
        glo, ghi = max(0, lo - w), min(nglob, hi + w)  # whole ranges, as the partition's dense halo takes them
        cl = (c - glo).astype(np.int32)
        n, ncols = hi - lo, ghi - glo
        e, r, st, pad, gw, (rmin, rmax) = probe(n, ncols, p, cl, 0, lo - glo, lo - glo + n)
        assert e == 1, (nglob, lo, hi)
        expect = (1 if lo > 0 else 0) + (1 if hi < nglob else 0)
        per_side = -(-w // (max(rmin, 1) * 512)) + 1  # workgroups whose rows lie within w of a cut (a share is at least rmin rounds of 512 rows)
        assert gw >= expect and gw <= expect * per_side, (gw, expect, per_side, rmin, rmax)
        assert rmax >= rmin >= 1
        e2, _, _, _, gw2, _ = probe(n, ncols, p, cl)  # no ghost range given: nothing marked
        assert e2 == 1 and gw2 == 0
