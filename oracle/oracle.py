"""oracle/oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes loaders for
  * ``libcpu_ref.so``  — our plain-C restatement (oracle/cpu_ref.c), always built;
  * ``_ref/libref_*.so`` — the REAL reference kernels compiled from
    /root/reference/mpk by oracle/Makefile (present only where they were built).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg
import this module.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_c = ctypes

_CPU = None
_REF = {}


def _as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def cpu():
    global _CPU
    if _CPU is None:
        path = os.path.join(_HERE, "libcpu_ref.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = ctypes.CDLL(path)
        L.orc_spmv_csr_fma.argtypes = [_c.c_int, _i32, _i32, _f64, _f64, _f64]
        L.orc_spmv_csr_muladd.argtypes = [_c.c_int, _i32, _i32, _f64, _f64, _f64]
        L.orc_spmv_csr_x87.argtypes = [_c.c_int, _i32, _i32, _f64, _f64, _f64]
        L.orc_spmk_chain.argtypes = [_c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64]
        L.orc_gen_layer1.argtypes = [_c.c_int, _i32, _i32, _i32]
        L.orc_spm2v_fused.argtypes = [_c.c_int, _i32, _i32, _f64, _i32, _f64, _f64, _f64]
        L.orc_spm2v_fused_x87.argtypes = [_c.c_int, _i32, _i32, _f64, _i32, _f64, _f64, _f64]
        L.orc_spmkv_fused.argtypes = [_c.c_int, _c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64]
        L.orc_spmv_bcsr4_fma.argtypes = [_c.c_int, _i32, _i32, _f64, _f64, _f64]
        L.orc_spmv_bcsr4_blockacc.argtypes = [_c.c_int, _i32, _i32, _f64, _f64, _f64]
        L.orc_norm2.argtypes = [_c.c_int, _f64]
        L.orc_norm2.restype = _c.c_double
        L.orc_rel_error.argtypes = [_c.c_int, _f64, _f64]
        L.orc_rel_error.restype = _c.c_double
        L.orc_dot.argtypes = [_c.c_int, _f64, _f64]
        L.orc_dot.restype = _c.c_double
        L.orc_orthogonalize.argtypes = [_c.c_int, _f64, _f64, _f64, _c.c_double]
        L.orc_orthogonalize.restype = _c.c_double
        L.orc_dot_gccvec.argtypes = [_c.c_int, _f64, _f64]
        L.orc_dot_gccvec.restype = _c.c_double
        L.orc_ortho_update.argtypes = [_c.c_int, _c.c_double, _f64, _f64, _f64]
        L.orc_orthogonalize_inplace.argtypes = [_c.c_int, _f64, _f64, _c.c_double]
        L.orc_orthogonalize_inplace.restype = _c.c_double
        L.orc_mgs.argtypes = [_c.c_int, _c.c_int, _f64, _f64, _c.c_void_p]
        L.orc_gen_layers.argtypes = [_c.c_int, _i32, _i32] + [_c.c_void_p] * 5 + [_c.POINTER(_c.c_longlong)] * 2
        L.orc_gen_layer1_bcsr4.argtypes = [_c.c_int, _i32, _i32, _i32]
        L.orc_axpy.argtypes = [_c.c_int, _c.c_double, _f64, _f64]
        L.orc_coo2csr.argtypes = [_c.c_int, _c.c_int, _i32, _i32, _f64, _i32, _i32, _f64]
        L.orc_coo2csr.restype = _c.c_int
        L.orc_coo2bcsr4.argtypes = [_c.c_int, _c.c_int, _i32, _i32, _f64, _c.c_void_p, _c.c_void_p, _c.c_void_p]
        L.orc_coo2bcsr4.restype = _c.c_int
        L.orc_read_mtx.argtypes = [_c.c_char_p, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int),
                                   _c.c_void_p, _c.c_void_p, _c.c_void_p]
        L.orc_read_mtx.restype = _c.c_int
        L.orc_time_spmv.argtypes = [_c.c_int, _i32, _i32, _f64, _f64, _f64, _c.c_int, _c.c_int]
        L.orc_time_spmv.restype = _c.c_double
        L.orc_spmv_csr_fma_omp.argtypes = [_c.c_int, _i32, _i32, _f64, _f64, _f64, _c.c_int]
        L.orc_time_spmv_omp.argtypes = [_c.c_int, _i32, _i32, _f64, _f64, _f64, _c.c_int, _c.c_int]
        L.orc_time_spmv_omp.restype = _c.c_double
        _CPU = L
    return _CPU


# ---------------------------------------------------------------- restatement

def spmv(ptrow, indcol, coef, x, kind="fma"):
    """y = A x.  kind: 'fma' (= SpMV_CSR_OPT/_FMA), 'muladd', 'x87' (= SpMV_CSR)."""
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    y = np.empty(n, np.float64)
    if kind == "fma":
        cpu().orc_spmv_csr_fma(n, ptrow, indcol, coef, x, y)
    elif kind == "muladd":
        cpu().orc_spmv_csr_muladd(n, ptrow, indcol, coef, x, y)
    elif kind == "x87":
        cpu().orc_spmv_csr_x87(n, ptrow, indcol, coef, x, y)
    else:
        raise ValueError(kind)
    return y


def spmk_chain(k, ptrow, indcol, coef, x):
    """[Ax, A^2x, ..., A^k x] as a (k, n) array by k chained fma SpMVs."""
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    Y = np.empty((k, n), np.float64)
    cpu().orc_spmk_chain(k, n, ptrow, indcol, coef, x, Y.reshape(-1))
    return Y


def gen_layer1(ptrow, indcol):
    ptrow, indcol = _as_i32(ptrow), _as_i32(indcol)
    n = len(ptrow) - 1
    end1 = np.empty(len(indcol), np.int32)
    cpu().orc_gen_layer1(n, ptrow, indcol, end1)
    return end1


def spm2v_fused(ptrow, indcol, coef, x, arith="fma"):
    """(y, z) = (Ax, A^2 x) by the reference's first-touch traversal.
    arith 'fma' = SpM2V_CSR_OPT, 'x87' = SpM2V_CSR object code (mpk/SpM2V.cpp:79-112)."""
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    end1 = gen_layer1(ptrow, indcol)
    y = np.empty(n, np.float64)
    z = np.empty(n, np.float64)
    fn = cpu().orc_spm2v_fused if arith == "fma" else cpu().orc_spm2v_fused_x87
    fn(n, ptrow, indcol, coef, end1, x, y, z)
    return y, z


ARITH = {"fma": 0, "x87": 1, "muladd": 2, "avx2row": 3}


def spmkv_fused(k, ptrow, indcol, coef, x, arith="fma"):
    """[Ax..A^k x] by the k-level first-touch traversal (SpM2V0/SpM3V/SpM4V)."""
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    Y = np.empty((k, n), np.float64)
    cpu().orc_spmkv_fused(ARITH[arith], k, n, ptrow, indcol, coef, x, Y.reshape(-1))
    return Y


def spmv_bcsr4(ptrow, indcol, coef, x):
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    nb = len(ptrow) - 1
    y = np.empty(4 * nb, np.float64)
    cpu().orc_spmv_bcsr4_fma(nb, ptrow, indcol, coef, x, y)
    return y


def spmv_bcsr4_blockacc(ptrow, indcol, coef, x):
    """y = A x with per-block partial sums (SpM2V_BCSR_OPT / spmm_avx2.c arithmetic, see cpu_ref.c)."""
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    nb = len(ptrow) - 1
    y = np.empty(4 * nb, np.float64)
    cpu().orc_spmv_bcsr4_blockacc(nb, ptrow, indcol, coef, x, y)
    return y


def norm2(x):
    x = _as_f64(x)
    return cpu().orc_norm2(len(x), x)


def rel_error(ref, test):
    ref, test = _as_f64(ref), _as_f64(test)
    assert ref.shape == test.shape
    return cpu().orc_rel_error(len(ref), ref, test)


def dot(a, b):
    a, b = _as_f64(a), _as_f64(b)
    return cpu().orc_dot(len(a), a, b)


def orthogonalize(b, x1, alpha=1e-8):
    """(beta, x3) with beta = b.x1, x3 = x1 - alpha*beta*b (mpk/SpMVmulti.cpp:146-151)."""
    b, x1 = _as_f64(b), _as_f64(x1)
    out = np.empty_like(x1)
    beta = cpu().orc_orthogonalize(len(b), b, x1, out, alpha)
    return beta, out


def dot_gccvec(a, b):
    """std::inner_product as the reference's object code evaluates it (rounded products, in-order sum)."""
    a, b = _as_f64(a), _as_f64(b)
    return cpu().orc_dot_gccvec(len(a), a, b)


def ortho_update(ab, b, x1):
    """fma(-ab, b, x1) elementwise: the update half of orthogonalize given alpha*beta."""
    b, x1 = _as_f64(b), _as_f64(x1)
    out = np.empty_like(x1)
    cpu().orc_ortho_update(len(b), float(ab), b, x1, out)
    return out


def orthogonalize_inplace(x, y, alpha=1e-8):
    """(beta, y') with y' = y - alpha*(x.y)*x — the in-place form of mpk/2SpMV.cpp:3-11."""
    x, y = _as_f64(x), _as_f64(y).copy()
    beta = cpu().orc_orthogonalize_inplace(len(x), x, y, alpha)
    return beta, y


def mgs(basis, y):
    """orthonormalize_against_basis (mpk/2SpMV.cpp:13-28): (y', dots) for basis of shape (m, n)."""
    basis = _as_f64(basis)
    m, n = basis.shape
    y = _as_f64(y).copy()
    dots = np.empty(m, np.float64)
    cpu().orc_mgs(n, m, basis.reshape(-1), y, dots.ctypes.data)
    return y, dots


def _layers(fn, n, ptrow, indcol, nnz):
    n2, n3 = _c.c_longlong(0), _c.c_longlong(0)
    fn(None, None, None, None, None, _c.byref(n2), _c.byref(n3))
    e1 = np.empty(nnz, np.int32)
    len2 = np.empty(nnz, np.int32)
    e2 = np.empty(max(n2.value, 1), np.int32)
    len3 = np.empty(max(n2.value, 1), np.int32)
    e3 = np.empty(max(n3.value, 1), np.int32)
    fn(e1.ctypes.data, len2.ctypes.data, e2.ctypes.data, len3.ctypes.data, e3.ctypes.data, _c.byref(n2), _c.byref(n3))
    return dict(e1=e1, len2=len2, e2=e2[: n2.value].copy(), len3=len3[: n2.value].copy(), e3=e3[: n3.value].copy())


def gen_layers(ptrow, indcol):
    """Flattened Generate1st/2nd/3rdlayer tables (see orc_gen_layers)."""
    ptrow, indcol = _as_i32(ptrow), _as_i32(indcol)
    n = len(ptrow) - 1
    L = cpu()
    return _layers(lambda *a: L.orc_gen_layers(n, ptrow, indcol, *a), n, ptrow, indcol, len(indcol))


def gen_layer1_bcsr4(ptrow, indcol):
    ptrow, indcol = _as_i32(ptrow), _as_i32(indcol)
    out = np.empty(len(indcol), np.int32)
    cpu().orc_gen_layer1_bcsr4(len(ptrow) - 1, ptrow, indcol, out)
    return out


def axpy(a, x, y):
    x = _as_f64(x)
    y = _as_f64(y).copy()
    cpu().orc_axpy(len(x), a, x, y)
    return y


def coo2csr(nrow, irow, jcol, val):
    irow, jcol, val = _as_i32(irow), _as_i32(jcol), _as_f64(val)
    nnz = len(irow)
    ptrow = np.empty(nrow + 1, np.int32)
    indcol = np.empty(max(nnz, 1), np.int32)
    coef = np.empty(max(nnz, 1), np.float64)
    stored = cpu().orc_coo2csr(nrow, nnz, irow, jcol, val, ptrow, indcol, coef)
    return ptrow, indcol[:stored].copy(), coef[:stored].copy()


def coo2bcsr4(nrow, irow, jcol, val):
    irow, jcol, val = _as_i32(irow), _as_i32(jcol), _as_f64(val)
    nnz = len(irow)
    nb = cpu().orc_coo2bcsr4(nrow, nnz, irow, jcol, val, None, None, None)
    ptrow = np.empty(nrow // 4 + 1, np.int32)
    indcol = np.empty(max(nb, 1), np.int32)
    coef = np.empty(max(nb, 1) * 16, np.float64)
    cpu().orc_coo2bcsr4(nrow, nnz, irow, jcol, val, ptrow.ctypes.data, indcol.ctypes.data, coef.ctypes.data)
    return ptrow, indcol[:nb].copy(), coef[: 16 * nb].copy()


def read_mtx(path):
    """(nrow, irow, jcol, val) with the reference reader's float32 rounding."""
    nrow, nnz = _c.c_int(0), _c.c_int(0)
    rc = cpu().orc_read_mtx(path.encode(), _c.byref(nrow), _c.byref(nnz), None, None, None)
    if rc != 0:
        raise OSError(f"orc_read_mtx({path}) -> {rc}")
    irow = np.empty(nnz.value, np.int32)
    jcol = np.empty(nnz.value, np.int32)
    val = np.empty(nnz.value, np.float64)
    rc = cpu().orc_read_mtx(path.encode(), _c.byref(nrow), _c.byref(nnz), irow.ctypes.data,
                            jcol.ctypes.data, val.ctypes.data)
    if rc != 0:
        raise OSError(f"orc_read_mtx({path}) -> {rc}")
    return nrow.value, irow, jcol, val


def time_spmv(ptrow, indcol, coef, x, reps=3, flush=True):
    """Best-of-reps seconds of the single-thread fma SpMV (cold cache if flush)."""
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    y = np.empty(n, np.float64)
    return cpu().orc_time_spmv(n, ptrow, indcol, coef, x, y, reps, 1 if flush else 0), y


def time_spmv_omp(ptrow, indcol, coef, x, nthreads, reps=5):
    """Best-of-reps seconds of the row-parallel fma SpMV on nthreads host threads (warm), and y."""
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    y = np.empty(n, np.float64)
    return cpu().orc_time_spmv_omp(n, ptrow, indcol, coef, x, y, reps, int(nthreads)), y


# ------------------------------------------------ the real reference (if built)

def have_ref():
    return all(os.path.exists(os.path.join(_HERE, "_ref", f"libref_{s}.so"))
               for s in ("spmv", "spm2v", "multi0", "2spmv", "multiold", "multi1"))


def ref(which):
    if which not in _REF:
        path = os.path.join(_HERE, "_ref", f"libref_{which}.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: `make -C oracle ref` needs /root/reference")
        L = ctypes.CDLL(path)
        if which == "spmv":
            L.ref_spmv_csr.argtypes = [_c.c_int, _c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64]
            L.ref_spmv_bcsr.argtypes = [_c.c_int, _c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64]
            L.ref_coo2csr.argtypes = [_c.c_int, _c.c_int, _i32, _i32, _f64, _i32, _i32, _f64]
            L.ref_coo2bcsr4.argtypes = [_c.c_int, _c.c_int, _i32, _i32, _f64, _c.c_void_p, _c.c_void_p, _c.c_void_p]
            L.ref_norm2.argtypes = [_c.c_int, _f64]
            L.ref_norm2.restype = _c.c_double
            L.ref_rel_error.argtypes = [_c.c_int, _f64, _f64]
            L.ref_rel_error.restype = _c.c_double
            L.ref_time_spmv_csr.argtypes = [_c.c_int, _c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64, _c.c_int, _c.c_int]
            L.ref_time_spmv_csr.restype = _c.c_double
        elif which == "spm2v":
            L.ref_gen_layer1.argtypes = [_c.c_int, _c.c_int, _i32, _i32, _i32]
            L.ref_spm2v_csr.argtypes = [_c.c_int, _c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64, _f64]
            L.ref_spm2v_bcsr.argtypes = [_c.c_int, _c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64, _f64]
            L.ref_gen_layer1_bcsr4.argtypes = [_c.c_int, _c.c_int, _i32, _i32, _i32]
        elif which == "multi0":
            L.ref_multi0_powers.argtypes = [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64]
            L.ref_multi0_layers.argtypes = [_c.c_int, _c.c_int, _i32, _i32] + [_c.c_void_p] * 5 + [_c.POINTER(_c.c_longlong)] * 2
        elif which == "2spmv":
            L.ref_orthogonalize_inplace.argtypes = [_c.c_int, _f64, _f64, _c.c_double]
            L.ref_orthonormalize_against_basis.argtypes = [_c.c_int, _c.c_int, _f64, _f64]
        elif which == "multiold":
            L.ref_orthogonalize3.argtypes = [_c.c_int, _f64, _f64, _f64, _c.c_double]
        elif which == "multi1":
            L.ref_multi1_spm4v_avx2.argtypes = [_c.c_int, _c.c_int, _i32, _i32, _f64, _f64, _f64]
        _REF[which] = L
    return _REF[which]


REF_VARIANTS = {"scalar": 0, "opt": 1, "fma": 2, "avx2": 3}


def ref_spmv(ptrow, indcol, coef, x, variant="scalar"):
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    y = np.full(n, np.nan)
    rc = ref("spmv").ref_spmv_csr(REF_VARIANTS[variant], n, len(indcol), ptrow, indcol, coef, x, y)
    assert rc == 0
    return y


def ref_spmv_bcsr(ptrow, indcol, coef, x, variant="fma"):
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    nb = len(ptrow) - 1
    y = np.full(4 * nb, np.nan)
    rc = ref("spmv").ref_spmv_bcsr(REF_VARIANTS[variant], nb, len(indcol), ptrow, indcol, coef, x, y)
    assert rc == 0
    return y


def ref_coo2csr(nrow, irow, jcol, val):
    irow, jcol, val = _as_i32(irow), _as_i32(jcol), _as_f64(val)
    nnz = len(irow)
    ptrow = np.empty(nrow + 1, np.int32)
    indcol = np.empty(max(nnz, 1), np.int32)
    coef = np.empty(max(nnz, 1), np.float64)
    stored = ref("spmv").ref_coo2csr(nrow, nnz, irow, jcol, val, ptrow, indcol, coef)
    return ptrow, indcol[:stored].copy(), coef[:stored].copy()


def ref_coo2bcsr4(nrow, irow, jcol, val):
    irow, jcol, val = _as_i32(irow), _as_i32(jcol), _as_f64(val)
    nnz = len(irow)
    nb = ref("spmv").ref_coo2bcsr4(nrow, nnz, irow, jcol, val, None, None, None)
    ptrow = np.empty(nrow // 4 + 1, np.int32)
    indcol = np.empty(max(nb, 1), np.int32)
    coef = np.empty(max(nb, 1) * 16, np.float64)
    ref("spmv").ref_coo2bcsr4(nrow, nnz, irow, jcol, val, ptrow.ctypes.data, indcol.ctypes.data, coef.ctypes.data)
    return ptrow, indcol[:nb].copy(), coef[: 16 * nb].copy()


def ref_gen_layer1(ptrow, indcol):
    ptrow, indcol = _as_i32(ptrow), _as_i32(indcol)
    end1 = np.empty(len(indcol), np.int32)
    ref("spm2v").ref_gen_layer1(len(ptrow) - 1, len(indcol), ptrow, indcol, end1)
    return end1


def ref_spm2v(ptrow, indcol, coef, x, variant="opt"):
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    y = np.full(n, np.nan)
    z = np.full(n, np.nan)
    rc = ref("spm2v").ref_spm2v_csr(REF_VARIANTS[variant], n, len(indcol), ptrow, indcol, coef, x, y, z)
    assert rc == 0
    return y, z


def ref_spm2v_bcsr(ptrow, indcol, coef, x, variant="fma"):
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    nb = len(ptrow) - 1
    y = np.full(4 * nb, np.nan)
    z = np.full(4 * nb, np.nan)
    rc = ref("spm2v").ref_spm2v_bcsr(REF_VARIANTS[variant], nb, len(indcol), ptrow, indcol, coef, x, y, z)
    assert rc == 0
    return y, z


def ref_powers(k, ptrow, indcol, coef, x, fused=True):
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    Y = np.full((k, n), np.nan)
    rc = ref("multi0").ref_multi0_powers(1 if fused else 0, k, n, len(indcol), ptrow, indcol, coef, x, Y.reshape(-1))
    assert rc == 0
    return Y


def ref_time_spmv(ptrow, indcol, coef, x, variant="scalar", reps=3, flush=True):
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    y = np.empty(n, np.float64)
    t = ref("spmv").ref_time_spmv_csr(REF_VARIANTS[variant], n, len(indcol), ptrow, indcol, coef, x, y, reps,
                                      1 if flush else 0)
    return t, y


def ref_gen_layers(ptrow, indcol):
    ptrow, indcol = _as_i32(ptrow), _as_i32(indcol)
    n, nnz = len(ptrow) - 1, len(indcol)
    L = ref("multi0")
    return _layers(lambda *a: L.ref_multi0_layers(n, nnz, ptrow, indcol, *a), n, ptrow, indcol, nnz)


def ref_gen_layer1_bcsr4(ptrow, indcol):
    ptrow, indcol = _as_i32(ptrow), _as_i32(indcol)
    out = np.empty(len(indcol), np.int32)
    rc = ref("spm2v").ref_gen_layer1_bcsr4(len(ptrow) - 1, len(indcol), ptrow, indcol, out)
    assert rc == 0
    return out


def ref_orthogonalize3(b, x1, alpha=1e-8):
    b, x1 = _as_f64(b), _as_f64(x1)
    out = np.full(len(b), np.nan)
    ref("multiold").ref_orthogonalize3(len(b), b, x1, out, alpha)
    return out


def ref_orthogonalize_inplace(x, y, alpha=1e-8):
    x, y = _as_f64(x), _as_f64(y).copy()
    ref("2spmv").ref_orthogonalize_inplace(len(x), x, y, alpha)
    return y


def ref_mgs(basis, y):
    basis = _as_f64(basis)
    m, n = basis.shape
    y = _as_f64(y).copy()
    ref("2spmv").ref_orthonormalize_against_basis(n, m, basis.reshape(-1), y)
    return y


def ref_spm4v_avx2(ptrow, indcol, coef, x):
    ptrow, indcol, coef, x = _as_i32(ptrow), _as_i32(indcol), _as_f64(coef), _as_f64(x)
    n = len(ptrow) - 1
    Y = np.full((4, n), np.nan)
    rc = ref("multi1").ref_multi1_spm4v_avx2(n, len(indcol), ptrow, indcol, coef, x, Y.reshape(-1))
    assert rc == 0
    return Y
