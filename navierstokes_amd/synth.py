"""Seeded synthetic CSR matrices (S15 / SVAR / SFE of SURVEY.md §8d).

Thin ctypes wrapper over ``csrc/synth_csr.c`` (plain C, built into
``csrc/libsynthcsr.so`` by ``__graft_entry__.build()``).  The generators are
counter-based per row, so ``rows(kind, n, rb, re)`` on any rank yields exactly
rows [rb, re) of the one global matrix.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

S15, SVAR, SFE = 0, 1, 2
KINDS = {"s15": S15, "svar": SVAR, "sfe": SFE}
DEFAULT_SEED = 0x5EED
DEFAULT_W = 2000

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "csrc", "libsynthcsr.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C navierstokes_amd/csrc`) first"
            )
        lib = ctypes.CDLL(path)
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        lib.synth_count.restype = ctypes.c_longlong
        lib.synth_count.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_longlong, ctypes.c_longlong]
        lib.synth_rows.restype = ctypes.c_int
        lib.synth_rows.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_longlong, ctypes.c_longlong, i32p, i32p, f64p]
        lib.synth_x_sin.restype = None
        lib.synth_x_sin.argtypes = [ctypes.c_longlong, ctypes.c_longlong, f64p]
        _LIB = lib
    return _LIB


def rows(kind, n, rb=0, re=None, seed=DEFAULT_SEED, w=DEFAULT_W):
    """Rows [rb, re) of the global n x n matrix.

    Returns (ptrow, indcol, coef): ptrow int32[(re-rb)+1] relative to the first
    generated nonzero, indcol int32 GLOBAL column ids (ascending per row),
    coef float64.
    """
    if isinstance(kind, str):
        kind = KINDS[kind.lower()]
    if re is None:
        re = n
    lib = _lib()
    nnz = lib.synth_count(kind, seed, n, w, rb, re)
    if nnz >= 2**31:
        raise ValueError("row range holds >= 2^31 nonzeros; int32 indices (mpk/SpMV.h:20-22) overflow")
    ptrow = np.empty(re - rb + 1, np.int32)
    indcol = np.empty(nnz, np.int32)
    coef = np.empty(nnz, np.float64)
    rc = lib.synth_rows(kind, seed, n, w, rb, re, ptrow, indcol, coef)
    if rc != 0:
        raise ValueError(f"synth_rows({kind=}, {n=}, {w=}, {rb=}, {re=}) rejected its arguments")
    return ptrow, indcol, coef


def x_sin(jb, je):
    """x_j = sin(0.001 j), j in [jb, je) — the reference's Krylov-basis seed vector (mpk/2SpMV.cpp:114)."""
    x = np.empty(je - jb, np.float64)
    _lib().synth_x_sin(jb, je, x)
    return x


def x_ones(n):
    """x = 1.0 — what the reference harnesses multiply by (mpk/SpM2V.cpp:879)."""
    return np.ones(n, np.float64)
