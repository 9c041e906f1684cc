"""The CPU oracle against the REAL reference object code (oracle/_ref), on
larger seeded matrices than the committed goldens.  Runs only where
oracle/_ref was built (a container with /root/reference); skipped elsewhere."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from navierstokes_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("kind,n,w", [("s15", 30000, 2000), ("svar", 20000, 500), ("sfe", 12000, 400)])
def test_spmv_all_variants(kind, n, w):
    p, c, v = synth.rows(kind, n, w=w)
    x = synth.x_sin(0, n)
    y = O.spmv(p, c, v, x, "fma")
    assert_bit_equal(y, O.ref_spmv(p, c, v, x, "opt"))
    assert_bit_equal(y, O.ref_spmv(p, c, v, x, "fma"))
    assert_bit_equal(O.spmv(p, c, v, x, "x87"), O.ref_spmv(p, c, v, x, "scalar"))
    assert O.rel_error(O.ref_spmv(p, c, v, x, "scalar"), y) <= 1e-15


@pytest.mark.parametrize("kind,n,w", [("s15", 5000, 300), ("svar", 4000, 100), ("sfe", 2000, 100)])
def test_powers(kind, n, w):
    p, c, v = synth.rows(kind, n, w=w)
    x = synth.x_ones(n)
    assert np.array_equal(O.gen_layer1(p, c), O.ref_gen_layer1(p, c))
    y, z = O.spm2v_fused(p, c, v, x)
    yr, zr = O.ref_spm2v(p, c, v, x, "opt")
    assert_bit_equal(y, yr)
    assert_bit_equal(z, zr)
    assert_bit_equal(O.spmkv_fused(3, p, c, v, x, "fma"), O.ref_powers(3, p, c, v, x))
    assert_bit_equal(O.spmkv_fused(4, p, c, v, x, "x87"), O.ref_powers(4, p, c, v, x))
    assert_bit_equal(O.spmkv_fused(2, p, c, v, x, "x87"), O.ref_powers(2, p, c, v, x))


def test_coo_rules_random():
    rng = np.random.default_rng(7)
    for nrow in (16, 33, 64):
        m = nrow * 9
        ir = rng.integers(0, nrow, m).astype(np.int32)
        jc = rng.integers(0, nrow, m).astype(np.int32)
        va = rng.uniform(-1, 1, m)
        a = O.coo2csr(nrow, ir, jc, va)
        b = O.ref_coo2csr(nrow, ir, jc, va)
        assert all(np.array_equal(s, t) for s, t in zip(a, b))
        a = O.coo2bcsr4(nrow, ir, jc, va)
        b = O.ref_coo2bcsr4(nrow, ir, jc, va)
        assert all(np.array_equal(s, t) for s, t in zip(a, b))


def test_blas1_random_lengths():
    """orthogonalize (3-vector and in-place) and orthonormalize_against_basis, live object code, every tail length."""
    rng = np.random.default_rng(11)
    for n in list(range(1, 13)) + [255, 1000, 4099]:
        b, x1 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        assert_bit_equal(O.orthogonalize(b, x1, 0.37)[1], O.ref_orthogonalize3(b, x1, 0.37), f"orthogonalize3 n={n}")
        assert_bit_equal(O.orthogonalize_inplace(b, x1, 0.37)[1], O.ref_orthogonalize_inplace(b, x1, 0.37), f"in place n={n}")
        B = rng.uniform(-1, 1, (5, n))
        assert_bit_equal(O.mgs(B, x1)[0], O.ref_mgs(B, x1), f"mgs n={n}")


@pytest.mark.parametrize("kind,n,w", [("s15", 3000, 200), ("svar", 2500, 80), ("sfe", 1200, 60)])
def test_layer_tables_and_avx2_powers(kind, n, w):
    p, c, v = synth.rows(kind, n, w=w)
    a, b = O.gen_layers(p, c), O.ref_gen_layers(p, c)
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    x = synth.x_sin(0, n)
    assert_bit_equal(O.spmkv_fused(4, p, c, v, x, "avx2row"), O.ref_spm4v_avx2(p, c, v, x), "SpM4V_AVX2")
