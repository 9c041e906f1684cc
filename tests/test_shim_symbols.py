"""The C++ symbols that libmpk_mi355.so exports under the reference's names — the functions a reference
driver actually binds — against the goldens produced by the reference's object code.  This file: the
host-side integer builders (no GPU needed), BIT-EXACT.  The compute symbols are in test_shim_gpu.py."""
import os

import numpy as np
import pytest

import shim
from conftest import assert_bit_equal

pytestmark = pytest.mark.skipif(not os.path.exists(shim.SHIM), reason="libmpk_mi355.so not built (run __graft_entry__.build())")

CSR_CASES = ["s15_n512", "svar_n400", "sfe_n268"]
COO_CASES = ["edge_coo_n37", "edge_coo_n40"]


@pytest.mark.parametrize("name", COO_CASES)
def test_COO2CSR_symbol_vs_reference_golden(golden, name):
    """COO2CSR / generate_CSR (mpk/utils.cpp:5-43, :97-127): ascending columns, FIRST duplicate kept,
    a.nnz left at the COO count, empty rows, nrow % 4 != 0."""
    g = golden(name)
    nrow = int(g["nrow"])
    p, c, v, nnz_field = shim.coo2csr(nrow, g["irow"], g["jcol"], g["val"])
    assert np.array_equal(p, g["csr_ptrow"]) and np.array_equal(c, g["csr_indcol"])
    assert_bit_equal(v, g["csr_coef"], "COO2CSR coef (first duplicate wins)")
    assert nnz_field == len(g["irow"])  # mpk/utils.cpp:100: the COO count even when duplicates were dropped
    assert len(c) < len(g["irow"])      # the cases do hold duplicates


@pytest.mark.parametrize("name", COO_CASES)
def test_generate_BCSR4_symbol_vs_reference_golden(golden, name):
    """generate_BCSR4 (mpk/utils.cpp:45-95): blocks in order of first appearance, LAST duplicate wins,
    nrow / 4 truncating."""
    g = golden(name)
    bp, bc, bv = shim.coo2bcsr4(int(g["nrow"]), g["irow"], g["jcol"], g["val"])
    assert np.array_equal(bp, g["bcsr_ptrow"]) and np.array_equal(bc, g["bcsr_indcol"])
    assert_bit_equal(bv, g["bcsr_coef"], "generate_BCSR4 coef (last duplicate wins)")


@pytest.mark.parametrize("name", CSR_CASES)
def test_Generate_layers_symbols_vs_reference_golden(golden, name):
    """Generate1stlayer (mpk/SpM2V.cpp:5-26), Generate2ndlayer / Generate3rdlayer (mpk/SpMVmulti0.cpp:106-130, :157-187)."""
    g = golden(name)
    p, c = g["ptrow"], g["indcol"]
    assert np.array_equal(shim.gen_layer1(p, c), g["end1"])
    lay = shim.gen_layers(p, c)
    assert np.array_equal(lay["e1"], g["end1"])
    for key in ("len2", "e2", "len3", "e3"):
        assert np.array_equal(lay[key], g["lay_" + key]), key


@pytest.mark.parametrize("name", COO_CASES)
def test_Generate1stlayer_symbols_on_coo_built_matrices(golden, name):
    g = golden(name)
    assert np.array_equal(shim.gen_layer1(g["csr_ptrow"], g["csr_indcol"]), g["end1"])
    if "bcsr_end1" in g:  # Generate1stlayer_BCSR4, mpk/SpM2V.cpp:28-46
        assert np.array_equal(shim.gen_layer1_bcsr4(g["bcsr_ptrow"], g["bcsr_indcol"]), g["bcsr_end1"])


def test_builders_random_vs_oracle():
    """More shapes than the goldens hold, against the oracle (itself pinned to the same goldens)."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    for nrow in (1, 3, 4, 8, 33, 64, 130):
        m = nrow * 7
        ir = rng.integers(0, nrow, m).astype(np.int32)
        jc = rng.integers(0, nrow, m).astype(np.int32)
        va = rng.uniform(-1, 1, m)
        p, c, v, _ = shim.coo2csr(nrow, ir, jc, va)
        po, co, vo = O.coo2csr(nrow, ir, jc, va)
        assert np.array_equal(p, po) and np.array_equal(c, co) and np.array_equal(v, vo)
        a, b = shim.coo2bcsr4(nrow, ir, jc, va), O.coo2bcsr4(nrow, ir, jc, va)
        assert all(np.array_equal(s, t) for s, t in zip(a, b))
        la, lb = shim.gen_layers(p, c), O.gen_layers(p, c)
        assert all(np.array_equal(la[k], lb[k]) for k in la)


@pytest.mark.parametrize("name", CSR_CASES)
def test_python_mirror_Generate1stlayer(golden, name):
    """navierstokes_amd.mpk.Generate1stlayer fills the table as the reference does (it used to be a no-op)."""
    from navierstokes_amd import mpk
    g = golden(name)
    A = mpk.csrmatrix(int(g["n"]), g["ptrow"], g["indcol"], g["coef"])
    assert np.array_equal(mpk.Generate1stlayer(None, A), g["end1"])
