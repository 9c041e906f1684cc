"""The compute symbols libmpk_mi355.so exports under the reference's names (include/SpMV.h), called as a
reference driver calls them — C++ signatures, host vectors — and compared with the goldens made by the
reference's object code.  Bitwise wherever the reference variant is an fma chain; 1e-15 against its x87 /
lane-interleaved AVX2 variants (tests/test_oracle_golden.py shows those differ from the fma chain by that much)."""
import os

import numpy as np
import pytest

import shim
from conftest import assert_bit_equal
from oracle import oracle as O  # checker only

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(shim.SHIM), reason="libmpk_mi355.so not built")]

CSR_CASES = ["s15_n512", "svar_n400", "sfe_n268"]


@pytest.mark.parametrize("name", CSR_CASES)
def test_SpMV_CSR_symbols(golden, name):
    g = golden(name)
    p, c, v, x = g["ptrow"], g["indcol"], g["coef"], g["x"]
    for fn in shim.CSR_VARIANTS:  # SpMV_CSR, _OPT, _FMA, _AVX2 (mpk/SpMV.cpp:6-85), SpMV (mpk/SpMVmulti0.cpp)
        y = shim.spmv_csr(fn, p, c, v, x)
        assert_bit_equal(y, g["y_fma"], fn + " vs reference SpMV_CSR_FMA")
        assert_bit_equal(y, g["y_opt"], fn + " vs reference SpMV_CSR_OPT")
        assert O.rel_error(g["y_scalar"], y) <= 1e-15
        assert shim.lib().shim_rel_error(len(y), g["y_scalar"], y) <= 1e-15  # rel_error() of the shim itself (GPU)


@pytest.mark.parametrize("name", ["edge_coo_n37", "edge_coo_n40"])
def test_SpMV_BCSR_symbols(golden, name):
    g = golden(name)
    xpad = np.concatenate([g["x"], np.zeros(4)])
    for fn in shim.BCSR_VARIANTS:  # mpk/SpMV.cpp:90-219
        # x padded as in make_golden.py: with nrow = 37 a block column straddles the end and the reference reads x[36..39]
        yb = shim.spmv_bcsr(fn, g["bcsr_ptrow"], g["bcsr_indcol"], g["bcsr_coef"], xpad)
        assert_bit_equal(yb, g["yb_fma"], fn + " vs reference SpMV_BCSR_FMA")
        assert_bit_equal(yb, g["yb_opt"], fn + " vs reference SpMV_BCSR_OPT")
        assert O.rel_error(g["yb_avx2"], yb) <= 1e-15


@pytest.mark.parametrize("name", CSR_CASES)
def test_SpM2V_CSR_symbols(golden, name):
    g = golden(name)
    p, c, v, x = g["ptrow"], g["indcol"], g["coef"], g["x"]
    for fn in shim.SPM2V_VARIANTS:  # mpk/SpM2V.cpp:79-332, mpk/SpMVmulti0.cpp:44-104
        y, z = shim.spm2v_csr(fn, p, c, v, x)
        assert_bit_equal(y, g["m2_y_opt"], fn + " y vs reference SpM2V_CSR_OPT")
        assert_bit_equal(z, g["m2_z_opt"], fn + " z vs reference SpM2V_CSR_OPT")
        assert O.rel_error(g["m2_z_scalar"], z) <= 1e-15
        assert O.rel_error(g["pow_fused2"][1], z) <= 1e-15  # SpM2V0 (x87 build)


def test_SpM2V_BCSR_symbols_vs_reference_m2b_goldens(golden):
    """SpM2V_BCSR{,_OPT,_FMA,_AVX2} (mpk/SpM2V.cpp:375-801) against the reference-made m2b_* vectors."""
    g = golden("edge_coo_n40")
    bp, bc, bv = g["bcsr_ptrow"], g["bcsr_indcol"], g["bcsr_coef"]
    # the CPU traversal leaves y = 0 in block rows that no block references as a column (SURVEY §8a-10
    # caveat); the GPU returns the true product there, so compare where the reference computed something
    seen = np.zeros(len(bp) - 1, bool)
    seen[bc] = True
    rows = np.repeat(seen, 4)
    assert rows.any()
    for fn in shim.SPM2VB_VARIANTS:
        y, z = shim.spm2v_bcsr(fn, bp, bc, bv, g["x"])
        for var in ("fma", "avx2"):  # one continuous fma chain per row, like SpMV_BCSR_FMA
            assert_bit_equal(y[rows], g["m2b_y_" + var][rows], f"{fn} y vs reference SpM2V_BCSR_{var.upper()}")
            assert_bit_equal(z, g["m2b_z_" + var], f"{fn} z vs reference SpM2V_BCSR_{var.upper()}")
        # SpM2V_BCSR_OPT sums each block's four products separately and adds the partial (mpk/SpM2V.cpp:502-507):
        # another association of the same sum (oracle: orc_spmv_bcsr4_blockacc, bit-pinned to it)
        assert O.rel_error(g["m2b_z_opt"], z) <= 1e-15 and O.rel_error(g["m2b_y_opt"][rows], y[rows]) <= 1e-15
        assert_bit_equal(y, O.spmv_bcsr4(bp, bc, bv, g["x"]), fn + " y on every row")


@pytest.mark.parametrize("name", CSR_CASES)
def test_SpM3V_SpM4V_symbols(golden, name):
    g = golden(name)
    p, c, v, x = g["ptrow"], g["indcol"], g["coef"], g["x"]
    assert_bit_equal(shim.powers("SpM3V", p, c, v, x), g["pow_fused3"], "SpM3V vs reference SpM3V (fma build)")
    Y4 = shim.powers("SpM4V", p, c, v, x)
    Y4a = shim.powers("SpM4V_AVX2", p, c, v, x)
    assert_bit_equal(Y4, Y4a, "SpM4V vs SpM4V_AVX2 (one GPU kernel)")
    assert_bit_equal(Y4[:3], g["pow_fused3"], "first three powers")
    for k in range(4):
        assert O.rel_error(g["pow_fused4"][k], Y4[k]) <= 1e-15   # reference SpM4V: x87 build
        assert O.rel_error(g["pow_avx2_4"][k], Y4[k]) <= 1e-15   # reference SpM4V_AVX2: lane-interleaved row sums


@pytest.mark.parametrize("name", ["blas1_n1003", "blas1_n2000"])
def test_orthogonalize_symbols(golden, name):
    """dot + AXPY between the SpMVs.  The GPU's dot is a fixed tree, not the CPU's left-to-right sum, so beta
    carries a rounding-level difference (bound below); GIVEN beta the update is the reference's fma, bit for bit."""
    g = golden(name)
    b, x1, alpha, n = g["b"], g["x1"], float(g["alpha"]), int(g["n"])
    scale = np.abs(b * x1).sum()
    x3 = shim.orthogonalize3(b, x1, alpha)            # mpk/SpMVmulti.cpp:146-151
    yi = shim.orthogonalize_inplace(b, x1, alpha)     # mpk/2SpMV.cpp:3-11
    assert_bit_equal(x3, yi, "the two forms share one GPU kernel")
    tol = alpha * 1e-13 * scale * np.abs(b).max() + 2e-16 * np.abs(x1).max()
    assert np.abs(x3 - g["x3_ortho3"]).max() <= tol
    assert np.abs(yi - g["y_ortho_inplace"]).max() <= tol
    # structure: x3 == fma(-(alpha*beta), b, x1) with the beta the GPU itself reports
    from navierstokes_amd import mpk
    out = np.empty(n)
    beta = mpk.orthogonalize(n, b, x1, out, alpha)
    assert abs(beta - O.dot_gccvec(b, x1)) <= 1e-13 * scale
    assert_bit_equal(out, O.ortho_update(alpha * beta, b, x1), "update given beta")
    assert_bit_equal(out, x3)


@pytest.mark.parametrize("name", ["blas1_n1003", "blas1_n2000"])
def test_orthonormalize_against_basis_symbol(golden, name):
    """mpk/2SpMV.cpp:13-28 — sequential projections, each on the y updated so far."""
    g = golden(name)
    basis, x1 = g["basis"], g["x1"]
    y = shim.orthonormalize_against_basis(basis, x1)
    ref = g["y_mgs"]
    # each of the m steps adds a rounding-level error in its dot, amplified by at most |v|^2 per later step
    assert O.rel_error(ref, y) <= 1e-12
    # and with the GPU's own coefficients the recurrence is the reference's fma, bit for bit
    from navierstokes_amd import mpk
    yy = x1.copy()
    dots = mpk.orthonormalize_against_basis(basis, yy)
    assert_bit_equal(yy, y)
    z = x1.copy()
    for j in range(len(basis)):
        z = O.ortho_update(dots[j], basis[j], z)
    assert_bit_equal(z, y, "recurrence with the GPU's dots")
    _, odots = O.mgs(basis, x1)
    assert np.abs(dots - odots).max() <= 1e-11 * max(1.0, np.abs(odots).max())


def test_inplace_coefficient_edit_is_seen(golden):
    """The stale-copy hazard: a caller rewrites coefficients IN PLACE (same arrays, same pattern) between two
    products, as the Newton loop does to its Jacobian (src/solve_newton.c:1245-1265).  The second product must
    use the new values — including an edit of ONE entry that no sampled fingerprint would hit."""
    from navierstokes_amd import synth
    n = 20000
    p, c, v = synth.rows("s15", n, w=300)
    x = synth.x_sin(0, n)
    nnz = len(c)
    for idx in ([nnz // 2 + 1], [7], [nnz - 1], [3, 1001, 77777, nnz - 5]):
        vals = [v[i] * -3.25 + 0.125 for i in idx]
        y0, y1 = shim.spmv_csr_inplace_edit(p, c, v, x, idx, vals)
        assert_bit_equal(y0, O.spmv(p, c, v, x), "before the edit")
        v2 = v.copy()
        v2[idx] = vals
        assert_bit_equal(y1, O.spmv(p, c, v2, x), f"after editing entries {idx} in place")
        assert not np.array_equal(y0, y1)
    # an FE matrix (AUTO may run the blocked copy: its values must be refreshed too) and the BCSR handle
    p, c, v = synth.fe_matrix(6)
    n = len(p) - 1
    x = synth.x_sin(0, n)
    idx = [len(c) // 3]
    y0, y1 = shim.spmv_csr_inplace_edit(p, c, v, x, idx, [2.5])
    v2 = v.copy()
    v2[idx] = 2.5
    assert_bit_equal(y0, O.spmv(p, c, v, x))
    assert_bit_equal(y1, O.spmv(p, c, v2, x), "FE matrix, one coefficient edited in place")
    bp, bc, bv = synth.csr_to_bcsr4(p, c, v)
    idx = [len(bv) // 2 + 3]
    y0, y1 = shim.spmv_bcsr_inplace_edit(bp, bc, bv, x, idx, [-1.75])
    bv2 = bv.copy()
    bv2[idx] = -1.75
    assert_bit_equal(y0, O.spmv_bcsr4(bp, bc, bv, x))
    assert_bit_equal(y1, O.spmv_bcsr4(bp, bc, bv2, x), "bcsr4x4_matrix, one coefficient edited in place")


def test_update_values_through_the_c_abi():
    """mi_csr_update_values on a handle big enough to be autotuned and (FE) blocked: same kernel choice, new bits."""
    from navierstokes_amd import mpk, synth
    for make in (lambda: synth.rows("s15", 300000), lambda: synth.fe_matrix(24)):
        p, c, v = make()
        n = len(p) - 1
        x = synth.x_sin(0, n)
        A = mpk.csrmatrix(n, p, c, v)
        y = np.empty(n)
        mpk.SpMV_CSR(y, x, A)
        assert_bit_equal(y, O.spmv(p, c, v, x))
        k0 = A.kernel_name()
        v2 = v * np.cos(np.arange(len(v)))
        A.update_values(v2)
        assert A.kernel_name() == k0
        mpk.SpMV_CSR(y, x, A)
        assert_bit_equal(y, O.spmv(p, c, v2, x), "after mi_csr_update_values (" + k0 + ")")


# ---- the reference's matrix-powers DRIVERS, unmodified, routed to the GPU (oracle/driver_main.cpp) ----------------

REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def _write_mtx(path, n, p, c, v):
    """PETSc-style MatrixMarket as the reference's readers expect it (mpk/SpM2V.cpp:815-852)."""
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, len(c)))
        for i in range(n):
            for k in range(p[i], p[i + 1]):
                f.write("%d %d %.9g\n" % (i + 1, c[k] + 1, v[k]))


def _run(exe, mtx, **env):
    import subprocess
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([os.path.join(REFDIR, exe), mtx], capture_output=True, text=True, timeout=300, env=e)


def _bound_to_shim(stderr, wanted):
    """LD_DEBUG=bindings lines 'binding file <driver.so> to <lib>: normal symbol `X'': which library each wanted symbol bound to."""
    out = {}
    for ln in stderr.splitlines():
        if "binding file" in ln and "libdrv_" in ln.split(" to ")[0] and "symbol `" in ln:
            sym = ln.split("symbol `")[1].split("'")[0]
            for w in wanted:
                if w in sym:
                    out[w] = os.path.basename(ln.split(" to ")[1].split(" ")[0].rstrip(":"))
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "spm2v_mi355")), reason="oracle/_ref/spm2v_mi355 not built")
def test_reference_SpM2V_driver_routed_to_the_gpu(tmp_path):
    """mpk/SpM2V.cpp's own main (:804-987), compiled where it lies and bound to the shim: its seven rel-err columns."""
    from navierstokes_amd import synth
    for kind, n, w in (("sfe", 268, 40), ("sfe", 20000, 300)):
        p, c, v = synth.rows(kind, n, w=w)
        mtx = str(tmp_path / f"{kind}{n}.mtx")
        _write_mtx(mtx, n, p, c, v)
        r = _run("spm2v_mi355", mtx, LD_DEBUG="bindings")
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("SpM2V")]
        assert len(lines) == 8, r.stdout
        errs = [float(ln.split("rel err =")[1]) for ln in lines[1:]]
        assert len(errs) == 7 and max(errs) <= 1e-15, lines
        b = _bound_to_shim(r.stderr, ["9SpM2V_CSR", "13SpM2V_CSR_OPT", "14SpM2V_CSR_AVX2", "10SpM2V_BCSR", "15SpM2V_BCSR_AVX2",
                                      "16Generate1stlayer", "22Generate1stlayer_BCSR4", "7COO2CSR", "14generate_BCSR4", "9rel_error"])
        assert b and all(lib == "libmpk_mi355.so" for lib in b.values()), b
        assert len(b) == 10, b


@pytest.mark.skipif(not (os.path.exists(os.path.join(REFDIR, "multi0_mi355")) and os.path.exists(os.path.join(REFDIR, "multi0_ref"))),
                    reason="oracle/_ref/multi0_* not built")
def test_reference_SpMVmulti0_driver_routed_to_the_gpu(tmp_path):
    """mpk/SpMVmulti0.cpp's own main (:317-418): prints the SpMV chain x_k beside the fused y_k for k = 1..4.  The
    routed driver's columns must equal each other and the CPU reference driver's columns (printed with %g)."""
    from navierstokes_amd import synth
    n = 268
    p, c, v = synth.rows("sfe", n, w=40)
    mtx = str(tmp_path / "sfe268.mtx")
    _write_mtx(mtx, n, p, c, v)

    def table(stderr):
        rows = [ln for ln in stderr.splitlines() if ln.startswith(":: ")]
        assert len(rows) == n
        return np.array([[float(t) for t in ln.replace(":", " ").split()[1:]] for ln in rows])  # x1 y1 x2 y2 x3 y3 x4 y4

    g = _run("multi0_mi355", mtx, LD_DEBUG="bindings")
    assert g.returncode == 0, g.stderr[-2000:]
    b = _bound_to_shim(g.stderr, ["4SpMV", "5SpM2V", "5SpM3V", "5SpM4V", "16Generate1stlayer", "16Generate2ndlayer", "16Generate3rdlayer", "7COO2CSR"])
    assert len(b) == 8 and all(lib == "libmpk_mi355.so" for lib in b.values()), b
    T = table(g.stderr)
    for k in range(4):
        assert np.array_equal(T[:, 2 * k], T[:, 2 * k + 1]), f"chain vs fused, power {k + 1}"
    cpu = _run("multi0_ref", mtx)
    assert cpu.returncode == 0
    R = table(cpu.stderr)
    assert np.allclose(T, R, rtol=2e-6, atol=1e-300), np.abs(T - R).max()  # six printed digits


def test_shim_degenerate_shapes():
    """Empty matrix, empty rows, one row: the shim (full-content hash, device copy) must not trip over empty arrays."""
    assert len(shim.spmv_csr("SpMV_CSR", np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0), np.zeros(0))) == 0
    p = np.array([0, 0, 2, 2, 3], np.int32)   # rows 0 and 2 empty
    c = np.array([1, 3, 0], np.int32)
    v = np.array([2.0, -1.0, 0.5])
    x = np.array([1.0, 2.0, 3.0, 4.0])
    assert np.array_equal(shim.spmv_csr("SpMV_CSR_FMA", p, c, v, x), [0.0, 0.0, 0.0, 0.5])
    y, z = shim.spm2v_csr("SpM2V_CSR", p, c, v, x)
    assert np.array_equal(y, [0.0, 0.0, 0.0, 0.5]) and np.array_equal(z, O.spmv(p, c, v, y))
    assert np.array_equal(shim.spmv_csr("SpMV_CSR", np.array([0, 1], np.int32), np.array([0], np.int32), np.array([3.0]), np.array([2.0])), [6.0])
