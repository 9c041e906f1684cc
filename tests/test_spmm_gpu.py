"""Multi-vector products and the s-step Krylov basis (src/kernels/spmm_avx2.c:7-168) on the GPU.
Oracle: per column, orc_spmv_bcsr4_fma (bit-pinned to SpMV_BCSR_FMA) for MI_ARITH_CHAIN and orc_spmv_bcsr4_blockacc
(bit-pinned to SpM2V_BCSR_OPT, the same association spmm_avx2.c:77-88 uses) for MI_ARITH_BLOCKACC.
The arithmetic of spmm_avx2.c itself cannot be run here (PETSc is absent: SURVEY F6) — for that file parity is
UNPINNED; what is pinned is the association it shares with SpM2V_BCSR_OPT, and its x4 defect is documented, not copied."""
import numpy as np
import pytest
import torch

from conftest import assert_bit_equal
from navierstokes_amd import mpk, synth
from oracle import oracle as O  # checker only

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _vectors(n, s):
    return np.stack([np.sin(0.001 * np.arange(n) + j) for j in range(s)])  # v_i[j] = sin(0.001 j + i), mpk/2SpMV.cpp:110-116


@pytest.mark.parametrize("s", [1, 2, 3, 4, 5, 8, 11])
def test_bcsr4_spmm_columns_bitwise(s):
    p, c, v = synth.fe_matrix(9)
    n = len(p) - 1
    bp, bc, bv = synth.csr_to_bcsr4(p, c, v)
    A = mpk.bcsr4x4_matrix(n // 4, bp, bc, bv, nbcols=n // 4)
    X = _vectors(n, s)
    for arith, orc in (("chain", O.spmv_bcsr4), ("blockacc", O.spmv_bcsr4_blockacc)):
        Y = torch.full((s, n), float("nan"), dtype=torch.float64, device="cuda")
        mpk.MatMatMult_SeqBAIJ_4(A, dev(X), Y, arith)
        Yh = np.full((s, n), np.nan)
        mpk.MatMatMult_SeqBAIJ_4(A, X, Yh, arith)
        for j in range(s):
            ref = orc(bp, bc, bv, X[j])
            assert_bit_equal(Y[j].cpu().numpy(), ref, f"{arith} column {j} of {s} (device)")
            assert_bit_equal(Yh[j], ref, f"{arith} column {j} of {s} (host)")
        if arith == "chain":  # = the CSR fma chain = the single-vector kernels
            assert_bit_equal(Y[0].cpu().numpy(), O.spmv(p, c, v, X[0]))
    # the two associations differ in the last bits only
    assert O.rel_error(O.spmv_bcsr4(bp, bc, bv, X[0]), O.spmv_bcsr4_blockacc(bp, bc, bv, X[0])) <= 1e-15


def test_bcsr4_spmm_padded_leading_dimension_and_empty_rows():
    # block rows 1 and 3 empty; columns strided with ldx > n
    bp = np.array([0, 2, 2, 3, 3], np.int32)
    bc = np.array([0, 3, 2], np.int32)
    rng = np.random.default_rng(3)
    bv = rng.uniform(-1, 1, 3 * 16)
    A = mpk.bcsr4x4_matrix(4, bp, bc, bv, nbcols=4)
    Xbig = torch.zeros((3, 24), dtype=torch.float64, device="cuda")
    X = rng.uniform(-1, 1, (3, 16))
    Xbig[:, :16] = dev(X)
    Ybig = torch.full((3, 20), float("nan"), dtype=torch.float64, device="cuda")
    mpk.MatMatMult_SeqBAIJ_4(A, Xbig, Ybig, "chain")
    for j in range(3):
        assert_bit_equal(Ybig[j, :16].cpu().numpy(), O.spmv_bcsr4(bp, bc, bv, X[j]))
    assert torch.isnan(Ybig[:, 16:]).all()  # nothing written behind a column


def test_csr_spmm_blocked_scrambled_and_unblocked(monkeypatch):
    """mi_spmm_dev on CSR handles: FE matrix (one launch over the blocked copy), the same matrix under a scrambled node
    numbering with the relabelling forced (gathered columns, block-row map), and a matrix without block structure."""
    p, c, v = synth.fe_matrix(10)
    n = len(p) - 1
    s = 4
    X = _vectors(n, s)
    cases = [("fe", p, c, v)]
    ps, cs, vs, _ = synth.permute_nodes(p, c, v, block=4, seed=9)
    cases.append(("fe scrambled", ps, cs, vs))
    p1, c1, v1 = synth.rows("svar", n, w=200)
    cases.append(("svar", p1, c1, v1))
    monkeypatch.setenv("MI355_REORDER", "1")
    for name, pp, cc, vv in cases:
        A = mpk.csrmatrix(n, pp, cc, vv)
        Y = torch.full((s, n), float("nan"), dtype=torch.float64, device="cuda")
        mpk.MatMatMult_SeqBAIJ_4(A, dev(X), Y)
        for j in range(s):
            assert_bit_equal(Y[j].cpu().numpy(), O.spmv(pp, cc, vv, X[j]), f"{name} column {j} ({A.kernel_name()}, {A.reorder_info()['reordered']})")


def test_krylov_basis_monomial():
    """BuildKrylovBasis_AVX2 (spmm_avx2.c:112-168): V[k+1] = A V[k] — bit-equal to the matrix-powers chain."""
    p, c, v = synth.fe_matrix(8)
    n = len(p) - 1
    s = 5
    v0 = synth.x_sin(0, n)
    A = mpk.csrmatrix(n, p, c, v)
    V, H, _ = mpk.BuildKrylovBasis(A, dev(v0), s)
    V = V.cpu().numpy()
    assert H is None
    assert_bit_equal(V[0], v0)
    Y = O.spmk_chain(s, p, c, v, v0)
    for k in range(s):
        assert_bit_equal(V[k + 1], Y[k], f"basis vector {k + 1}")


def test_krylov_basis_orthonormal():
    """orth: product -> orthonormalize_against_basis (mpk/2SpMV.cpp:13-28) -> / norm2 (mpk/utils.cpp:131-136).
    The device's dots and norms come from fixed trees, not the CPU's left-to-right sums, so (a) the recurrence is replayed
    with ITS coefficients and must then match bit for bit, (b) the coefficients are held against the oracle's with a
    rounding-level bound, (c) the result is what it claims to be: orthonormal, spanning the Krylov space (A V_k = V_{k+1} H_k)."""
    p, c, v = synth.rows("s15", 40000, w=300)   # ||A|| < 2: a well-scaled Krylov sequence
    n = len(p) - 1
    s = 6
    v0 = synth.x_sin(0, n) + 0.25
    A = mpk.csrmatrix(n, p, c, v)
    V, H, nrm0 = mpk.BuildKrylovBasis(A, dev(v0), s, orth=True)
    V, H, nrm0 = V.cpu().numpy(), H.cpu().numpy(), float(nrm0)
    assert abs(nrm0 - O.norm2(v0)) <= 1e-13 * nrm0
    assert_bit_equal(V[0], v0 / nrm0, "v0 / ||v0|| (IEEE division)")
    for k in range(s):
        w = O.spmv(p, c, v, V[k])
        wo, dots_o = O.mgs(V[: k + 1], w)
        for j in range(k + 1):
            w = O.ortho_update(H[k, j], V[j], w)
        assert_bit_equal(V[k + 1], w / H[k, k + 1], f"basis vector {k + 1} from the device's own coefficients")
        assert np.abs(H[k, : k + 1] - dots_o).max() <= 1e-12 * max(1.0, np.abs(dots_o).max())
        assert abs(H[k, k + 1] - O.norm2(wo)) <= 1e-11 * O.norm2(wo)
    G = V @ V.T
    assert np.abs(G - np.eye(s + 1)).max() <= 1e-10, np.abs(G - np.eye(s + 1)).max()
    for k in range(s):  # Arnoldi relation
        lhs = O.spmv(p, c, v, V[k])
        rhs = H[k, : k + 1] @ V[: k + 1] + H[k, k + 1] * V[k + 1]
        assert O.rel_error(lhs, rhs) <= 1e-12


@pytest.mark.parametrize("form", ["1", "2", "3", "4"])
@pytest.mark.parametrize("s", [2, 4, 6, 8, 11])
def test_spmm_every_tile_form_bitwise(form, s, monkeypatch):
    """The multi-vector product's tile forms forced on (MI355_SPMM_TILE: 1 = four lanes per block row, 2 / 3 = eight lanes per block
    row with temporal / non-temporal coefficient loads; a form that does not exist for a column count falls back to the gather
    kernels): tiles are breadth-first clusters of the block graph, rows of a tile are not consecutive — every column bit-equal to
    the oracle in both associations, on an FE matrix, on the same matrix under a random node numbering, and on a matrix with empty
    block rows and a block-row count that is no multiple of the tile size.  Form 4 (round 4) is the sliced stream (spmm_bcsr4_sell:
    four and eight columns; eleven = a batch of eight through it and three through the gather kernels)."""
    monkeypatch.setenv("MI355_SPMM_TILE", form)
    if form == "4":
        monkeypatch.setenv("MI355_BCSR_SELL", "1")  # (these matrices are below the size from which the sliced copy is built by itself)
    mats = []
    p, c, v = synth.fe_matrix(13, 11, 9)
    mats.append(("fe", p, c, v))
    p2, c2, v2, _ = synth.permute_nodes(p, c, v, block=4, seed=3)
    mats.append(("fe scrambled", p2, c2, v2))
    for name, p, c, v in mats:
        n = len(p) - 1
        bp, bc, bv = synth.csr_to_bcsr4(p, c, v)
        if name == "fe":  # empty block rows
            keep = np.ones(n // 4, bool)
            keep[[5, 200, 201, n // 4 - 1]] = False
            lens = np.diff(bp) * keep
            sel = np.repeat(keep, np.diff(bp))
            bp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            bc, bv = bc[sel], bv.reshape(-1, 16)[sel].reshape(-1)
        A = mpk.bcsr4x4_matrix(n // 4, bp, bc, bv, nbcols=n // 4)
        X = _vectors(n, s)
        if form == "4" and s in (4, 8):
            import ctypes
            fi = ctypes.c_int(-1)
            mpk.check(mpk.lib().mi_bcsr4_spmm_info(A.handle, s, None, ctypes.byref(fi), None, None))
            assert fi.value == 4, fi.value
        for arith, orc in (("chain", O.spmv_bcsr4), ("blockacc", O.spmv_bcsr4_blockacc)):
            for rep in range(2):
                Y = torch.full((s, n), float("nan"), dtype=torch.float64, device="cuda")
                mpk.MatMatMult_SeqBAIJ_4(A, dev(X), Y, arith)
                Yh = Y.cpu().numpy()
                for j in range(s):
                    assert_bit_equal(Yh[j], orc(bp, bc, bv, X[j]), f"{name} form {form} s={s} {arith} column {j} rep {rep}")
