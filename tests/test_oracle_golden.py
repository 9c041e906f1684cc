"""The CPU oracle (oracle/cpu_ref.c) against golden vectors produced by the
reference's own object code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from oracle import oracle as O

CSR_CASES = ["s15_n512", "svar_n400", "sfe_n268"]


@pytest.mark.parametrize("name", CSR_CASES)
def test_spmv_variants_bitwise(golden, name):
    g = golden(name)
    p, c, v, x = g["ptrow"], g["indcol"], g["coef"], g["x"]
    # SpMV_CSR_OPT / SpMV_CSR_FMA (mpk/SpMV.cpp:23-56) == sequential fma chain
    assert_bit_equal(O.spmv(p, c, v, x, "fma"), g["y_fma"], "fma vs SpMV_CSR_FMA")
    assert_bit_equal(O.spmv(p, c, v, x, "fma"), g["y_opt"], "fma vs SpMV_CSR_OPT")
    # SpMV_CSR (mpk/SpMV.cpp:5-20), x87 build
    assert_bit_equal(O.spmv(p, c, v, x, "x87"), g["y_scalar"], "x87 vs SpMV_CSR")
    # the fma oracle is within 1e-15 of the x87 one: the level that licenses it as THE oracle
    assert O.rel_error(g["y_scalar"], O.spmv(p, c, v, x, "fma")) <= 1e-15
    if "y_avx2" in g:
        assert O.rel_error(g["y_scalar"], g["y_avx2"]) <= 1e-15


@pytest.mark.parametrize("name", CSR_CASES)
def test_layer1_and_spm2v(golden, name):
    g = golden(name)
    p, c, v, x = g["ptrow"], g["indcol"], g["coef"], g["x"]
    assert np.array_equal(O.gen_layer1(p, c), g["end1"])  # Generate1stlayer, mpk/SpM2V.cpp:5-26
    y, z = O.spm2v_fused(p, c, v, x)
    assert_bit_equal(y, g["m2_y_opt"], "SpM2V_CSR_OPT y")
    assert_bit_equal(z, g["m2_z_opt"], "SpM2V_CSR_OPT z")
    yx, zx = O.spm2v_fused(p, c, v, x, "x87")
    assert_bit_equal(yx, g["m2_y_scalar"], "SpM2V_CSR y")
    assert_bit_equal(zx, g["m2_z_scalar"], "SpM2V_CSR z")
    assert O.rel_error(g["m2_z_scalar"], z) <= 1e-15
    # fused == two chained SpMVs when every row is referenced as a column (diagonal present)
    Y = O.spmk_chain(2, p, c, v, x)
    assert_bit_equal(Y[0], y)
    assert_bit_equal(Y[1], z)


@pytest.mark.parametrize("name", CSR_CASES)
def test_matrix_powers_k234(golden, name):
    g = golden(name)
    p, c, v, x = g["ptrow"], g["indcol"], g["coef"], g["x"]
    # SpM2V0 and SpM4V are x87 builds, SpM3V is an fma build (mpk/SpMVmulti0.cpp:42-61,132-155,189-221)
    assert_bit_equal(O.spmkv_fused(2, p, c, v, x, "x87"), g["pow_fused2"], "SpM2V0")
    assert_bit_equal(O.spmkv_fused(3, p, c, v, x, "fma"), g["pow_fused3"], "SpM3V")
    assert_bit_equal(O.spmkv_fused(4, p, c, v, x, "x87"), g["pow_fused4"], "SpM4V")
    # chain of 4 x87 SpMVs (mpk/SpMVmulti0.cpp:369-373) == fused x87 traversal
    assert_bit_equal(g["pow_chain4"], g["pow_fused4"], "reference chain vs reference fused")
    # our fma chain (what the GPU path is compared with) is within 1e-15 of every power
    Y = O.spmk_chain(4, p, c, v, x)
    for k in range(4):
        assert O.rel_error(g["pow_chain4"][k], Y[k]) <= 1e-15
    assert_bit_equal(O.spmkv_fused(4, p, c, v, x, "fma"), Y, "fma fused == fma chain")


@pytest.mark.parametrize("name", CSR_CASES)
def test_norm_and_rel_error(golden, name):
    g = golden(name)
    assert abs(O.norm2(g["y_scalar"]) - float(g["norm2_y"])) <= 4e-16 * float(g["norm2_y"])
    r = O.rel_error(g["y_scalar"], g["pert"])
    assert abs(r - float(g["rel_err_pert"])) <= 1e-12 * float(g["rel_err_pert"])


@pytest.mark.parametrize("name", ["edge_coo_n37", "edge_coo_n40"])
def test_coo_builders(golden, name):
    g = golden(name)
    nrow = int(g["nrow"])
    p, c, v = O.coo2csr(nrow, g["irow"], g["jcol"], g["val"])  # COO2CSR: sorted, first duplicate wins
    assert np.array_equal(p, g["csr_ptrow"]) and np.array_equal(c, g["csr_indcol"])
    assert_bit_equal(v, g["csr_coef"], "COO2CSR coef")
    assert_bit_equal(O.spmv(p, c, v, g["x"], "fma"), g["y_fma"])
    assert_bit_equal(O.spmv(p, c, v, g["x"], "x87"), g["y_scalar"])
    bp, bc, bv = O.coo2bcsr4(nrow, g["irow"], g["jcol"], g["val"])  # generate_BCSR4: appearance order, last wins
    assert np.array_equal(bp, g["bcsr_ptrow"]) and np.array_equal(bc, g["bcsr_indcol"])
    assert_bit_equal(bv, g["bcsr_coef"], "generate_BCSR4 coef")
    xpad = np.concatenate([g["x"], np.zeros(4)])
    yb = O.spmv_bcsr4(bp, bc, bv, xpad)
    assert_bit_equal(yb, g["yb_fma"], "SpMV_BCSR_FMA")
    assert_bit_equal(yb, g["yb_opt"], "SpMV_BCSR_OPT")
    assert O.rel_error(g["yb_scalar"], yb) <= 1e-15
    assert O.rel_error(g["yb_scalar"], g["yb_avx2"]) <= 1e-15


def test_empty_rows_and_empty_matrix():
    # rows 1 and 3 empty; a 0-nnz matrix; y must be fully overwritten with zeros (mpk/SpMV.cpp:13)
    p = np.array([0, 2, 2, 3, 3], np.int32)
    c = np.array([0, 3, 2], np.int32)
    v = np.array([2.0, -1.0, 0.5])
    x = np.array([1.0, 2.0, 3.0, 4.0])
    assert np.array_equal(O.spmv(p, c, v, x), [-2.0, 0.0, 1.5, 0.0])
    p0 = np.zeros(5, np.int32)
    assert np.array_equal(O.spmv(p0, np.zeros(0, np.int32), np.zeros(0), x), np.zeros(4))


def test_unreferenced_row_caveat():
    # a row never referenced as a column keeps y[j] = 0 in the fused kernel
    p = np.array([0, 1, 2, 3], np.int32)
    c = np.array([0, 0, 0], np.int32)  # only column 0 is ever referenced
    v = np.array([1.0, 2.0, 3.0])
    x = np.array([1.0, 1.0, 1.0])
    y, z = O.spm2v_fused(p, c, v, x)
    assert np.array_equal(y, [1.0, 0.0, 0.0])  # rows 1, 2 never computed
    assert np.array_equal(z, [1.0, 2.0, 3.0])  # z still right
    assert np.array_equal(O.spmk_chain(2, p, c, v, x)[1], z)


def test_mtx_reader_float_rounding(tmp_path):
    # mpk/SpM2V.cpp:844-851 reads "%d %d %f" into float: coefficients are rounded to binary32
    txt = "%%MatrixMarket matrix coordinate real general\n% a comment\n3 3 4\n1 1 0.1\n2 3 -1.23456789012345\n3 1 1e-3\n2 2 7\n"
    f = tmp_path / "m.mtx"
    f.write_text(txt)
    nrow, ir, jc, va = O.read_mtx(str(f))
    assert nrow == 3 and list(ir) == [0, 1, 2, 1] and list(jc) == [0, 2, 0, 1]
    exp = np.array([np.float32("0.1"), np.float32("-1.23456789012345"), np.float32("1e-3"), np.float32(7)], np.float64)
    assert np.array_equal(va, exp)
    assert va[0] != 0.1  # i.e. NOT the double nearest to 0.1


def test_orthogonalize_and_axpy():
    rng = np.random.default_rng(1)
    b, x1 = rng.standard_normal(1000), rng.standard_normal(1000)
    beta, x3 = O.orthogonalize(b, x1, 1e-8)
    assert abs(beta - float(np.dot(b, x1))) <= 1e-12 * np.abs(b * x1).sum()
    assert np.allclose(x3, x1 - 1e-8 * beta * b, rtol=0, atol=1e-18)
    assert np.allclose(O.axpy(0.5, b, x1), x1 + 0.5 * b, rtol=1e-15)


@pytest.mark.parametrize("name", ["blas1_n1003", "blas1_n2000"])
def test_blas1_pinned_to_reference_object_code(golden, name):
    """orthogonalize (both forms) and orthonormalize_against_basis, bit for bit."""
    g = golden(name)
    b, x1, alpha = g["b"], g["x1"], float(g["alpha"])
    beta, x3 = O.orthogonalize(b, x1, alpha)              # mpk/SpMVmulti.cpp:146-151
    assert_bit_equal(x3, g["x3_ortho3"], "orthogonalize(nrow, b, x1, x3, alpha)")
    assert beta == O.dot_gccvec(b, x1)
    beta2, y = O.orthogonalize_inplace(b, x1, alpha)      # mpk/2SpMV.cpp:3-11
    assert_bit_equal(y, g["y_ortho_inplace"], "orthogonalize(nrow, x, y, alpha) in place")
    assert beta2 == O.dot(b, x1)
    ym, dots = O.mgs(g["basis"], x1)                      # mpk/2SpMV.cpp:13-28
    assert_bit_equal(ym, g["y_mgs"], "orthonormalize_against_basis")
    assert len(dots) == int(g["m"])
    # the two dot orders agree to rounding: what licenses a tolerance (not bits) for the GPU's tree reduction
    assert abs(beta - beta2) <= 1e-13 * np.abs(b * x1).sum()


@pytest.mark.parametrize("name", CSR_CASES)
def test_nested_layer_tables_and_spm4v_avx2(golden, name):
    g = golden(name)
    p, c, v, x = g["ptrow"], g["indcol"], g["coef"], g["x"]
    lay = O.gen_layers(p, c)  # Generate2ndlayer / Generate3rdlayer, mpk/SpMVmulti0.cpp:106-130, :157-187
    assert np.array_equal(lay["e1"], g["end1"])
    for key in ("len2", "e2", "len3", "e3"):
        assert np.array_equal(lay[key], g["lay_" + key]), key
    # SpM4V_AVX2 (mpk/SpMVmulti-1.cpp:434-493): bitwise with its lane-interleaved row sums, 1e-15 from the fma chain
    assert_bit_equal(O.spmkv_fused(4, p, c, v, x, "avx2row"), g["pow_avx2_4"], "SpM4V_AVX2")
    Y = O.spmk_chain(4, p, c, v, x)
    for k in range(4):
        assert O.rel_error(g["pow_avx2_4"][k], Y[k]) <= 1e-15


@pytest.mark.parametrize("name", ["edge_coo_n37", "edge_coo_n40"])
def test_first_touch_tables_of_coo_built_matrices(golden, name):
    g = golden(name)
    assert np.array_equal(O.gen_layer1(g["csr_ptrow"], g["csr_indcol"]), g["end1"])
    if "bcsr_end1" in g:
        assert np.array_equal(O.gen_layer1_bcsr4(g["bcsr_ptrow"], g["bcsr_indcol"]), g["bcsr_end1"])


def test_fused_bcsr_variants_pin_two_associations(golden):
    """SpM2V_BCSR_FMA / _AVX2 = one continuous fma chain per row (= two chained SpMV_BCSR_FMA); SpM2V_BCSR_OPT =
    per-block partial sums (mpk/SpM2V.cpp:502-507).  Both restated and pinned bitwise (rows the CPU traversal computed)."""
    g = golden("edge_coo_n40")
    bp, bc, bv, x = g["bcsr_ptrow"], g["bcsr_indcol"], g["bcsr_coef"], g["x"]
    seen = np.zeros(len(bp) - 1, bool)
    seen[bc] = True
    rows = np.repeat(seen, 4)
    y = O.spmv_bcsr4(bp, bc, bv, x)
    z = O.spmv_bcsr4(bp, bc, bv, np.where(rows, y, 0.0))
    for var in ("fma", "avx2"):
        assert_bit_equal(y[rows], g["m2b_y_" + var][rows])
        assert_bit_equal(z, g["m2b_z_" + var])
    yb = O.spmv_bcsr4_blockacc(bp, bc, bv, x)
    zb = O.spmv_bcsr4_blockacc(bp, bc, bv, np.where(rows, yb, 0.0))
    assert_bit_equal(yb[rows], g["m2b_y_opt"][rows], "SpM2V_BCSR_OPT y")
    assert_bit_equal(zb, g["m2b_z_opt"], "SpM2V_BCSR_OPT z")
    assert O.rel_error(zb, z) <= 1e-15
