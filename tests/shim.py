"""ctypes loader for tests/shim_harness/libshim_harness.so: extern "C" doors into the C++ symbols that
libmpk_mi355.so exports under the reference's names (TEST INFRASTRUCTURE)."""
import ctypes
import os
import subprocess

import numpy as np

from conftest import ROOT

_c = ctypes
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
DIR = os.path.join(ROOT, "tests", "shim_harness")
PATH = os.path.join(DIR, "libshim_harness.so")
SHIM = os.path.join(ROOT, "navierstokes_amd", "csrc", "libmpk_mi355.so")
_L = None


def available():
    return os.path.exists(PATH) or os.path.exists(SHIM)


def lib():
    global _L
    if _L is None:
        if not os.path.exists(PATH):
            subprocess.check_call(["make", "-C", DIR, "all"])  # g++ only; needs the built shim
        L = ctypes.CDLL(PATH)
        i = _c.c_int
        L.shim_coo2csr.argtypes = [i, i, _i32, _i32, _f64, _i32, _i32, _f64, _c.POINTER(i)]
        L.shim_coo2bcsr4.argtypes = [i, i, _i32, _i32, _f64, _c.c_void_p, _c.c_void_p, _c.c_void_p]
        L.shim_gen_layer1.argtypes = [i, i, _i32, _i32, _i32]
        L.shim_gen_layer1_bcsr4.argtypes = [i, i, _i32, _i32, _i32]
        L.shim_layers.argtypes = [i, i, _i32, _i32] + [_c.c_void_p] * 5 + [_c.POINTER(_c.c_longlong)] * 2
        L.shim_spmv_csr.argtypes = [i, i, i, _i32, _i32, _f64, _f64, _f64]
        L.shim_spmv_bcsr.argtypes = [i, i, i, _i32, _i32, _f64, _f64, _f64]
        L.shim_spmv_csr_inplace_edit.argtypes = [i, i, _i32, _i32, _f64, _f64, i, _i32, _f64, _f64, _f64]
        L.shim_spmv_bcsr_inplace_edit.argtypes = [i, i, _i32, _i32, _f64, _f64, i, _i32, _f64, _f64, _f64]
        L.shim_time_spmv_csr.argtypes = [i, i, _i32, _i32, _f64, _f64, _f64, i, i]
        L.shim_time_spmv_csr.restype = _c.c_double
        L.shim_spm2v_csr.argtypes = [i, i, i, _i32, _i32, _f64, _f64, _f64, _f64]
        L.shim_spm2v_bcsr.argtypes = [i, i, i, _i32, _i32, _f64, _f64, _f64, _f64]
        L.shim_powers.argtypes = [i, i, i, _i32, _i32, _f64, _f64, _f64]
        L.shim_orthogonalize3.argtypes = [i, _f64, _f64, _f64, _c.c_double]
        L.shim_orthogonalize_inplace.argtypes = [i, _f64, _f64, _c.c_double]
        L.shim_orthonormalize_against_basis.argtypes = [i, i, _f64, _f64]
        L.shim_norm2.argtypes = [i, _f64]
        L.shim_norm2.restype = _c.c_double
        L.shim_rel_error.argtypes = [i, _f64, _f64]
        L.shim_rel_error.restype = _c.c_double
        _L = L
    return _L


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def coo2csr(nrow, irow, jcol, val):
    irow, jcol, val = _i(irow), _i(jcol), _f(val)
    nnz = len(irow)
    p, c, v = np.empty(nrow + 1, np.int32), np.empty(max(nnz, 1), np.int32), np.empty(max(nnz, 1))
    field = _c.c_int(-1)
    stored = lib().shim_coo2csr(nrow, nnz, irow, jcol, val, p, c, v, _c.byref(field))
    return p, c[:stored].copy(), v[:stored].copy(), field.value


def coo2bcsr4(nrow, irow, jcol, val):
    irow, jcol, val = _i(irow), _i(jcol), _f(val)
    nnz = len(irow)
    nb = lib().shim_coo2bcsr4(nrow, nnz, irow, jcol, val, None, None, None)
    p, c, v = np.empty(nrow // 4 + 1, np.int32), np.empty(max(nb, 1), np.int32), np.empty(max(nb, 1) * 16)
    lib().shim_coo2bcsr4(nrow, nnz, irow, jcol, val, p.ctypes.data, c.ctypes.data, v.ctypes.data)
    return p, c[:nb].copy(), v[: 16 * nb].copy()


def gen_layer1(ptrow, indcol):
    ptrow, indcol = _i(ptrow), _i(indcol)
    out = np.empty(len(indcol), np.int32)
    assert lib().shim_gen_layer1(len(ptrow) - 1, len(indcol), ptrow, indcol, out) == 0
    return out


def gen_layer1_bcsr4(ptrow, indcol):
    ptrow, indcol = _i(ptrow), _i(indcol)
    out = np.empty(len(indcol), np.int32)
    assert lib().shim_gen_layer1_bcsr4(len(ptrow) - 1, len(indcol), ptrow, indcol, out) == 0
    return out


def gen_layers(ptrow, indcol):
    from oracle import oracle as O
    ptrow, indcol = _i(ptrow), _i(indcol)
    n, nnz = len(ptrow) - 1, len(indcol)
    L = lib()
    return O._layers(lambda *a: L.shim_layers(n, nnz, ptrow, indcol, *a), n, ptrow, indcol, nnz)


CSR_VARIANTS = {"SpMV_CSR": 0, "SpMV_CSR_OPT": 1, "SpMV_CSR_FMA": 2, "SpMV_CSR_AVX2": 3, "SpMV": 4}
BCSR_VARIANTS = {"SpMV_BCSR": 0, "SpMV_BCSR_OPT": 1, "SpMV_BCSR_FMA": 2, "SpMV_BCSR_AVX2": 3}
SPM2V_VARIANTS = {"SpM2V_CSR": 0, "SpM2V_CSR_OPT": 1, "SpM2V_CSR_FMA": 2, "SpM2V_CSR_AVX2": 3, "SpM2V0": 4, "SpM2V": 5}
SPM2VB_VARIANTS = {"SpM2V_BCSR": 0, "SpM2V_BCSR_OPT": 1, "SpM2V_BCSR_FMA": 2, "SpM2V_BCSR_AVX2": 3}
POWERS = {"SpM3V": 3, "SpM4V": 4, "SpM4V_AVX2": 5}


def spmv_csr(name, p, c, v, x):
    p, c, v, x = _i(p), _i(c), _f(v), _f(x)
    y = np.full(len(p) - 1, np.nan)
    assert lib().shim_spmv_csr(CSR_VARIANTS[name], len(p) - 1, len(c), p, c, v, x, y) == 0
    return y


def spmv_bcsr(name, p, c, v, x):
    p, c, v, x = _i(p), _i(c), _f(v), _f(x)
    y = np.full(4 * (len(p) - 1), np.nan)
    assert lib().shim_spmv_bcsr(BCSR_VARIANTS[name], len(p) - 1, len(c), p, c, v, x, y) == 0
    return y


def spm2v_csr(name, p, c, v, x):
    p, c, v, x = _i(p), _i(c), _f(v), _f(x)
    n = len(p) - 1
    y, z = np.full(n, np.nan), np.full(n, np.nan)
    assert lib().shim_spm2v_csr(SPM2V_VARIANTS[name], n, len(c), p, c, v, x, y, z) == 0
    return y, z


def spm2v_bcsr(name, p, c, v, x):
    p, c, v, x = _i(p), _i(c), _f(v), _f(x)
    n = 4 * (len(p) - 1)
    y, z = np.full(n, np.nan), np.full(n, np.nan)
    assert lib().shim_spm2v_bcsr(SPM2VB_VARIANTS[name], len(p) - 1, len(c), p, c, v, x, y, z) == 0
    return y, z


def powers(name, p, c, v, x):
    p, c, v, x = _i(p), _i(c), _f(v), _f(x)
    n = len(p) - 1
    k = 3 if name == "SpM3V" else 4
    Y = np.full((k, n), np.nan)
    assert lib().shim_powers(POWERS[name], n, len(c), p, c, v, x, Y.reshape(-1)) == 0
    return Y


def spmv_csr_inplace_edit(p, c, v, x, idx, vals):
    p, c, v, x = _i(p), _i(c), _f(v), _f(x)
    n = len(p) - 1
    y0, y1 = np.full(n, np.nan), np.full(n, np.nan)
    assert lib().shim_spmv_csr_inplace_edit(n, len(c), p, c, v, x, len(idx), _i(idx), _f(vals), y0, y1) == 0
    return y0, y1


def spmv_bcsr_inplace_edit(p, c, v, x, idx, vals):
    p, c, v, x = _i(p), _i(c), _f(v), _f(x)
    n = 4 * (len(p) - 1)
    y0, y1 = np.full(n, np.nan), np.full(n, np.nan)
    assert lib().shim_spmv_bcsr_inplace_edit(len(p) - 1, len(c), p, c, v, x, len(idx), _i(idx), _f(vals), y0, y1) == 0
    return y0, y1


def orthogonalize3(b, x1, alpha):
    b, x1 = _f(b), _f(x1)
    out = np.full(len(b), np.nan)
    lib().shim_orthogonalize3(len(b), b, x1, out, alpha)
    return out


def orthogonalize_inplace(x, y, alpha):
    x, y = _f(x), _f(y).copy()
    lib().shim_orthogonalize_inplace(len(x), x, y, alpha)
    return y


def orthonormalize_against_basis(basis, y):
    basis = _f(basis)
    m, n = basis.shape
    y = _f(y).copy()
    lib().shim_orthonormalize_against_basis(n, m, basis.reshape(-1), y)
    return y


def time_spmv_csr(p, c, v, x, reps=3, trust=False):
    """(seconds per SpMV_CSR call through the shim on a live csrmatrix, y)"""
    p, c, v, x = _i(p), _i(c), _f(v), _f(x)
    y = np.full(len(p) - 1, np.nan)
    t = lib().shim_time_spmv_csr(len(p) - 1, len(c), p, c, v, x, y, reps, 1 if trust else 0)
    return t, y
