"""navierstokes_amd.mpk — host-side mirror of the reference's mpk/ interface.

Same names, argument order and meaning as the free functions of the
reference's ``mpk/SpMV.h:52-66`` (plus the per-file kernels ``SpM2V_CSR``
``mpk/SpM2V.cpp:80``, ``SpM4V`` ``mpk/SpMVmulti0.cpp:191`` and
``orthogonalize`` ``mpk/SpMVmulti.cpp:146``), so that tests read like the
reference's harnesses: ``SpMV_CSR(y, x, A)`` overwrites ``y`` with ``A x``.

Everything is computed by the HIP library ``csrc/libmi355spmv.so`` through its
C-ABI (``include/mi355_spmv.h``).  PyTorch appears only as plumbing: vectors
may be CUDA/HIP tensors (device-resident, launched on torch's current stream,
no synchronisation) or numpy arrays (copied in and out like the reference's
CPU functions).  There is NO CPU fallback: importing this module without the
built library, or calling it without a GPU, raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_SPMV_LIBRARY") or os.path.join(_HERE, "csrc", "libmi355spmv.so")  # env: A/B of two builds (tools/)

MI_OK = 0
KERNEL_AUTO, KERNEL_STREAM, KERNEL_RING, KERNEL_ROWPAR = 0, 1, 2, 3
KERNELS = {"auto": 0, "stream": 1, "ring": 2, "rowpar": 3, "bcsr4": 4, "tile": 5, "mring": 6, "sstream": 7}

_c = ctypes
_vp = ctypes.c_void_p
_LIB = None


class MiError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"libmi355spmv status {status}: {detail}")
        self.status = status


def lib():
    """The loaded C-ABI library (loads torch's HIP runtime first when torch is around)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing — the HIP extension was not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for this path."
        )
    try:  # one HIP runtime per process: let torch's copy win if torch is installed
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    L.mi_strerror.restype = _c.c_char_p
    L.mi_last_error.restype = _c.c_char_p
    L.mi_csr_kernel_name.restype = _c.c_char_p
    L.mi_csr_kernel_name.argtypes = [_vp]
    i, ll, d = _c.c_int, _c.c_longlong, _c.c_double
    P = _c.POINTER
    sigs = {
        "mi_version": [],
        "mi_device_count": [P(i)],
        "mi_set_device": [i],
        "mi_device_synchronize": [],
        "mi_flush_cache": [],
        "mi_flush_cache_async": [_vp],
        "mi_csr_create": [i, i, _vp, _vp, _vp, P(_vp)],
        "mi_csr_create_mapped": [i, i, _vp, _vp, _vp, _vp, P(_vp)],
        "mi_csr_destroy": [_vp],
        "mi_csr_update_values": [_vp, _vp],
        "mi_csr_update_values_dev": [_vp, _vp, _vp],
        "mi_bcsr4_update_values": [_vp, _vp],
        "mi_bcsr4_update_values_dev": [_vp, _vp, _vp],
        "mi_orthonormalize_against_basis": [i, i, _vp, _vp, _vp],
        "mi_orthonormalize_against_basis_dev": [i, i, _vp, _vp, _vp, _vp],
        "mi_part_status": [_vp],
        "mi_part_push_export": [_vp, _vp, _vp],
        "mi_part_push_connect": [_vp, _vp, _vp],
        "mi_part_spmv_push_dev": [_vp, _vp, _vp, _vp],
        "mi_part_push_info": [_vp, P(i), P(i), P(i)],
        "mi_part_sends_contiguous": [_vp, P(i)],
        "mi_part_combined_info": [_vp, P(i)],
        "mi_part_push_disable": [_vp],
        "mi_part_push_unfuse": [_vp],
        "mi_bcsr4_spmm": [_vp, i, _vp, ll, _vp, ll, i],
        "mi_bcsr4_spmm_info": [_vp, i, P(i), P(i), P(i), P(d)],
        "mi_bcsr4_spmm_dev": [_vp, i, _vp, ll, _vp, ll, i, _vp],
        "mi_spmm_dev": [_vp, i, _vp, ll, _vp, ll, _vp],
        "mi_krylov_basis_dev": [_vp, i, _vp, _vp, ll, i, _vp, _vp],
        "mi_csr_reorder_info": [_vp, P(i), P(i), P(d), P(d), P(d), P(d)],
        "mi_reorder_probe": [i, _vp, _vp, P(i), _vp, P(d), P(d)],
        "mi_csr_perm": [_vp, P(i), _vp],
        "mi_vec_to_internal_dev": [_vp, _vp, _vp, _vp],
        "mi_vec_from_internal_dev": [_vp, _vp, _vp, _vp],
        "mi_spmv_internal_dev": [_vp, _vp, _vp, _vp],
        "mi_spmk_internal_dev": [_vp, i, _vp, _vp, _vp],
        "mi_csr_dims": [_vp, P(i), P(i), P(ll)],
        "mi_csr_set_kernel": [_vp, i],
        "mi_csr_get_kernel": [_vp, P(i)],
        "mi_csr_ring_info": [_vp, P(i), P(i), P(i), P(d)],
        "mi_csr_ring_shape_info": [_vp, P(i), P(i), P(i), P(d), P(d)],
        "mi_csr_tune_info": [_vp, P(d), P(d)],
        "mi_csr_tune_detail": [_vp, P(d), P(_c.c_int), P(_c.c_int)],
        "mi_csr_set_nontemporal": [_vp, i, i],
        "mi_csr_block4_structure": [i, _vp, _vp, P(i), P(_c.c_longlong)],
        "mi_ring_plan_probe": [i, _vp, _vp, i, P(i), P(i), P(i), P(d), P(i)],
        "mi_ring_plan_lean": [i, _vp, _vp, i, P(i)],
        "mi_csr_tile_info": [_vp, P(i), P(i), P(d), P(d), P(i)],
        "mi_bcsr4_tile_info": [_vp, P(i), P(i), P(d), P(d)],
        "mi_bcsr4_sell_info": [_vp, P(i), P(i), P(ll), P(d), P(d)],
        "mi_csr_mring_info": [_vp, P(i), P(i), P(i), P(d), P(d), P(i)],
        "mi_csr_sstream_info": [_vp, P(i), P(i), P(ll), P(d), P(d), P(i)],
        "mi_sstream_plan_probe": [i, i, _vp, _vp, P(i), P(i), P(ll), P(d)],
        "mi_sstream_plan_probe_ex": [i, i, _vp, _vp, i, i, i, P(i), P(i), P(ll), P(d), P(i), P(i)],
        "mi_sstream_mw_plan_probe": [i, i, _vp, _vp, i, P(i), P(i), P(ll), P(d)],
        "mi_csr_placement_info": [_vp, P(i), P(i), P(d), i],
        "mi_vec_alloc_placed": [_vp, i, i, P(_vp), P(d), i, P(i)],
        "mi_vec_free_placed": [_vp],
        "mi_mring_plan_deal_probe": [i, _vp, _vp, P(i), _vp, i],
        "mi_mring_plan_probe": [i, _vp, _vp, P(i), P(i), P(i), P(d), P(ll)],
        "mi_tile_plan_probe": [i, _vp, _vp, i, P(i), P(ll), P(i), P(ll)],
        "mi_stream_read_probe": [ll, i, P(d)],
        "mi_spmv": [_vp, _vp, _vp],
        "mi_spmv_dev": [_vp, _vp, _vp, _vp],
        "mi_spmk": [_vp, i, _vp, _vp],
        "mi_spmk_dev": [_vp, i, _vp, _vp, _vp],
        "mi_csr_spmk_info": [_vp, i, P(i), P(i), P(d), P(d)],
        "mi_spmk_plan_probe": [i, _vp, _vp, P(i), P(i), P(i)],
        "mi_dot": [i, _vp, _vp, P(d)],
        "mi_dot_dev": [i, _vp, _vp, _vp, _vp],
        "mi_axpy": [i, d, _vp, _vp],
        "mi_axpy_dev": [i, d, _vp, _vp, _vp],
        "mi_orthogonalize": [i, _vp, _vp, _vp, d, P(d)],
        "mi_orthogonalize_dev": [i, _vp, _vp, _vp, d, _vp, _vp],
        "mi_spmv_dot_dev": [_vp, _vp, _vp, _vp, _vp, _vp],
        "mi_csr_dot_epilogue_info": [_vp, P(i)],
        "mi_spmv_orthogonalize_dev": [_vp, _vp, _vp, _vp, _vp, d, _vp, _vp],
        "mi_norm2": [i, _vp, P(d)],
        "mi_norm2_dev": [i, _vp, _vp, _vp],
        "mi_rel_error": [i, _vp, _vp, P(d)],
        "mi_rel_error_dev": [i, _vp, _vp, _vp, _vp],
        "mi_gather_dev": [i, _vp, _vp, _vp, _vp],
        "mi_bcsr4_create": [i, i, _vp, _vp, _vp, P(_vp)],
        "mi_bcsr4_destroy": [_vp],
        "mi_bcsr4_create_layout": [i, i, _vp, _vp, _vp, i, P(_vp)],
        "mi_bcsr4_update_values_layout": [_vp, _vp, i],
        "mi_bcsr4_update_values_layout_dev": [_vp, _vp, i, _vp],
        "mi_bcsr4_spmv": [_vp, _vp, _vp],
        "mi_bcsr4_spmv_dev": [_vp, _vp, _vp, _vp],
        "mi_bcsr4_spmk": [_vp, i, _vp, _vp],
        "mi_bcsr4_spmk_dev": [_vp, i, _vp, _vp, _vp],
        "mi_part_create": [i, i, _vp, _vp, _vp, _vp, P(_vp)],
        "mi_part_destroy": [_vp],
        "mi_part_sizes": [_vp, P(i), P(i), P(i), P(i)],
        "mi_part_recv_counts": [_vp, _vp],
        "mi_part_recv_ids": [_vp, i, _vp],
        "mi_part_set_send_ids": [_vp, i, i, _vp],
        "mi_part_send_counts": [_vp, _vp],
        "mi_part_local_csr": [_vp, i, P(i), P(_vp), P(_vp), P(_vp), P(_vp)],
        "mi_part_send_index": [_vp, P(i), P(_vp)],
        "mi_part_finalize": [_vp],
        "mi_part_set_kernel": [_vp, i],
        "mi_part_update_values": [_vp, _vp],
        "mi_part_pack_dev": [_vp, _vp, _vp, _vp],
        "mi_part_spmv_interior_dev": [_vp, _vp, _vp, _vp],
        "mi_part_spmv_boundary_dev": [_vp, _vp, _vp, _vp],
        "mi_comm_available": [],
        "mi_comm_unique_id": [_vp],
        "mi_part_comm_init": [_vp, _vp],
        "mi_part_comm_info": [_vp, P(i), P(i)],
        "mi_part_spmv_dev": [_vp, _vp, _vp, _vp],
        "mi_comm_selftest": [i, P(d)],
        "mi_part_send_union": [_vp, P(i), P(_vp)],
        "mi_part_allgather_setup": [_vp, _vp, _vp],
        "mi_part_set_allgather": [_vp, i],
        "mi_part_allgather_info": [_vp, P(i), P(i), P(i)],
        "mi_dist_create": [i, i, _vp, _vp, _vp, P(_vp)],
        "mi_dist_destroy": [_vp],
        "mi_dist_info": [_vp, P(i), P(i), P(i), P(i), P(ll), P(ll)],
        "mi_dist_rank_info": [_vp, i, P(i), P(ll), P(i), P(i), P(ll), P(_vp)],
        "mi_dist_update_values": [_vp, _vp],
        "mi_dist_spmv": [_vp, _vp, _vp],
        "mi_dist_spmk": [_vp, i, _vp, _vp],
        "mi_dist_dot": [_vp, _vp, _vp, P(d)],
        "mi_dist_orthogonalize": [_vp, _vp, _vp, _vp, d, P(d)],
        "mi_dist_vec_create": [_vp, P(_vp)],
        "mi_dist_vec_destroy": [_vp],
        "mi_dist_vec_set": [_vp, _vp],
        "mi_dist_vec_get": [_vp, _vp],
        "mi_dist_vec_ptr": [_vp, i, P(_vp)],
        "mi_dist_spmv_dev": [_vp, _vp, _vp],
        "mi_dist_spmk_dev": [_vp, i, _vp, _vp],
        "mi_dist_synchronize": [_vp],
        "mi_dist_dot_dev": [_vp, _vp, _vp, P(d)],
        "mi_dist_orthogonalize_dev": [_vp, _vp, _vp, _vp, d, P(d)],
    }
    for name, argt in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = argt
        fn.restype = _c.c_int
    for name in ("mi_dist_exchange_name", "mi_dist_exchange_note"):
        getattr(L, name).argtypes = [_vp]
        getattr(L, name).restype = _c.c_char_p
    L.mi_part_kernel_name.argtypes = [_vp, i]
    L.mi_part_kernel_name.restype = _c.c_char_p
    # diagnostics of include/mi355_devtools.h: present only in libmi355spmv_dev.so (`make devtools`; tools/ load it through
    # MI355_SPMV_LIBRARY), never in the product library
    for name, argt in {"mi_debug_xcc_map": [i, _vp], "mi_debug_touch_pages": [_vp, i, _vp, ll, _vp, ll],
                       "mi_part_push_debug_preset": [_vp, _c.c_uint], "mi_debug_part_push_trace": [_vp, _vp, _vp, i, _vp, P(i), _vp],
                       "mi_debug_part_ext_mode": [_vp, i], "mi_debug_part_ext_trace": [_vp, _vp, _vp, i, _vp, P(i), _vp]}.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.argtypes = argt
            fn.restype = _c.c_int
    _LIB = L
    return L


def check(status):
    if status != MI_OK:
        raise MiError(status, lib().mi_last_error().decode(errors="replace"))


# ----------------------------------------------------------------- plumbing

def _is_torch(t):
    return type(t).__module__.startswith("torch")


def _stream_ptr():
    import torch

    return _vp(torch.cuda.current_stream().cuda_stream)


def _dev_ptr(t, n=None, what="vector"):
    import torch

    if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise TypeError(f"{what}: expected a contiguous float64 CUDA tensor, got {t.dtype} on {t.device}")
    if n is not None and t.numel() < n:
        raise ValueError(f"{what}: needs {n} entries, has {t.numel()}")
    return _vp(t.data_ptr())


def _host_f64(a, n=None, what="vector", writable=False):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags.c_contiguous:
        if writable:
            raise TypeError(f"{what}: output must be a C-contiguous float64 numpy array")
        a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.size < n:
        raise ValueError(f"{what}: needs {n} entries, has {a.size}")
    return a


# ------------------------------------------------------------------ matrices

class csrmatrix:
    """struct csrmatrix of mpk/SpMV.h:18-24 — n, nnz, ptrow, indcol, coef — plus the
    device handle (lazily created; the matrix is immutable once used)."""

    def __init__(self, n, ptrow, indcol, coef, ncols=None, rowmap=None):
        self.n = int(n)
        self.ptrow = np.ascontiguousarray(ptrow, dtype=np.int32)
        self.indcol = np.ascontiguousarray(indcol, dtype=np.int32)
        self.coef = np.ascontiguousarray(coef, dtype=np.float64)
        if len(self.ptrow) != self.n + 1:
            raise ValueError("ptrow must have n+1 entries")
        self.nnz = int(self.ptrow[-1]) if self.n > 0 else 0
        self.ncols = self.n if ncols is None else int(ncols)
        self.rowmap = None if rowmap is None else np.ascontiguousarray(rowmap, dtype=np.int32)
        self._h = None
        self._kernel = KERNEL_AUTO

    @property
    def handle(self):
        if self._h is None:
            h = _vp()
            if self.rowmap is None:
                check(lib().mi_csr_create(self.n, self.ncols, self.ptrow.ctypes.data, self.indcol.ctypes.data,
                                          self.coef.ctypes.data, _c.byref(h)))
            else:
                check(lib().mi_csr_create_mapped(self.n, self.ncols, self.ptrow.ctypes.data, self.indcol.ctypes.data,
                                                 self.coef.ctypes.data, self.rowmap.ctypes.data, _c.byref(h)))
            self._h = h
            if self._kernel != KERNEL_AUTO:
                check(lib().mi_csr_set_kernel(h, self._kernel))
        return self._h

    def set_kernel(self, kernel):
        kid = KERNELS[kernel] if isinstance(kernel, str) else int(kernel)
        self._kernel = kid
        if self._h is not None:
            check(lib().mi_csr_set_kernel(self._h, kid))
        return self

    def kernel_name(self):
        return lib().mi_csr_kernel_name(self.handle).decode()

    def ring_info(self):
        """(config id, runs, runs not served by the ring, fraction of nonzeros the ring serves)."""
        cfg, runs, bad = _c.c_int(), _c.c_int(), _c.c_int()
        frac = _c.c_double()
        check(lib().mi_csr_ring_info(self.handle, _c.byref(cfg), _c.byref(runs), _c.byref(bad), _c.byref(frac)))
        return cfg.value, runs.value, bad.value, frac.value

    def ring_shape_info(self):
        """dict(blocks, lean, depth, us_aligned, us_unaligned) — mi_csr_ring_shape_info."""
        b, l, dd = _c.c_int(), _c.c_int(), _c.c_int()
        ua, uu = _c.c_double(), _c.c_double()
        check(lib().mi_csr_ring_shape_info(self.handle, _c.byref(b), _c.byref(l), _c.byref(dd), _c.byref(ua), _c.byref(uu)))
        return dict(blocks=b.value, lean=bool(l.value), depth=dd.value, us_aligned=ua.value, us_unaligned=uu.value)

    def tune_info(self):
        """(us per launch measured for ring, for stream) at create time; zeros if not measured."""
        a, b = _c.c_double(), _c.c_double()
        check(lib().mi_csr_tune_info(self.handle, _c.byref(a), _c.byref(b)))
        return a.value, b.value

    def tune_detail(self):
        """(us per launch measured at create time for ring / stream, each with temporal and non-temporal
        matrix loads; whether the kernel that AUTO resolves to uses non-temporal loads)."""
        us = (_c.c_double * 5)()
        rnt, snt = _c.c_int(), _c.c_int()
        check(lib().mi_csr_tune_detail(self.handle, us, _c.byref(rnt), _c.byref(snt)))
        nt = rnt.value if "ring" in self.kernel_name() else snt.value
        out = dict(ring=us[0], ring_nt=us[1], stream=us[2], stream_nt=us[3], bcsr4=us[4])
        m = self.mring_info()
        if m["built"]:
            out.update(mring=m["us"], mring_nt=m["us_nt"])
            if "mring" in self.kernel_name():
                nt = m["nt"]
        t = self.tile_info()
        if t["built"]:
            out.update(tile=t["us"], tile_nt=t["us_nt"])
            if "tile" in self.kernel_name():
                nt = t["nt"]
        return out, bool(nt)

    def placement_info(self):
        """dict(values=[us as first allocated, us after each draw ...], column_stream=[...]) — mi_csr_placement_info (create-time placement draws)."""
        nv, nt = _c.c_int(), _c.c_int()
        us = (_c.c_double * 32)()
        check(lib().mi_csr_placement_info(self.handle, _c.byref(nv), _c.byref(nt), us, 32))
        return dict(values=[round(us[k], 2) for k in range(nv.value)], column_stream=[round(us[k], 2) for k in range(nv.value, nt.value)])

    def alloc_vectors(self, nvec=2, draws=8):
        """nvec device vectors (torch float64 tensors of length n, zero-filled) placed for products with this matrix —
        mi_vec_alloc_placed: the library allocates `draws` candidate x / y pairs, times y = A x on each and keeps the fastest
        (where a vector lies in device memory moves a product by up to 12 % on some boxes, profiles/NOTES.md §4.12).
        Returns (tensors, us_per_candidate_pair).  The memory belongs to the returned tensors' `_placed` owner: it is released
        when the last of them is garbage-collected."""
        import torch
        ptrs = (_vp * nvec)()
        us = (_c.c_double * 64)()
        nus = _c.c_int()
        check(lib().mi_vec_alloc_placed(self.handle, nvec, draws, ptrs, us, 64, _c.byref(nus)))
        out = []
        for k in range(nvec):
            buf = _PlacedBuffer(int(ptrs[k]), max(self.n, self.ncols))
            t = torch.as_tensor(buf, device="cuda")[: self.n]
            t._placed = buf  # keeps the allocation alive as long as the tensor object lives
            out.append(t)
        return out, [round(us[k], 2) for k in range(nus.value)]

    def mring_info(self):
        """dict(built, runs, runs_not_served, nnz_fraction, us, us_nt, nt) — mi_csr_mring_info (multi-window ring plan)."""
        b, r, bad, nt = _c.c_int(), _c.c_int(), _c.c_int(), _c.c_int()
        f = _c.c_double()
        us = (_c.c_double * 2)()
        check(lib().mi_csr_mring_info(self.handle, _c.byref(b), _c.byref(r), _c.byref(bad), _c.byref(f), us, _c.byref(nt)))
        return dict(built=bool(b.value), runs=r.value, runs_not_served=bad.value, nnz_fraction=f.value, us=us[0], us_nt=us[1],
                    nt=bool(nt.value))

    def sstream_info(self):
        """dict(built, rounds, steps, padding, us_d8_nt, us_d8_temporal, us_d12_nt, us_d12_temporal, form) — mi_csr_sstream_info (sliced-stream kernel)."""
        b, r, fm = _c.c_int(), _c.c_int(), _c.c_int()
        st = _c.c_longlong()
        pad = _c.c_double()
        us = (_c.c_double * 4)()
        check(lib().mi_csr_sstream_info(self.handle, _c.byref(b), _c.byref(r), _c.byref(st), _c.byref(pad), us, _c.byref(fm)))
        return dict(built=bool(b.value), rounds=r.value, steps=st.value, padding=pad.value, us_d8_nt=us[0], us_d8_temporal=us[1], us_d12_nt=us[2],
                    us_d12_temporal=us[3], form=fm.value)

    def tile_info(self):
        """dict(built, nblk, unique_per_nnz, us, us_nt, nt) — mi_csr_tile_info (the tile kernel's plan on this handle)."""
        b, nb, nt = _c.c_int(), _c.c_int(), _c.c_int()
        u = _c.c_double()
        us = (_c.c_double * 2)()
        check(lib().mi_csr_tile_info(self.handle, _c.byref(b), _c.byref(nb), _c.byref(u), us, _c.byref(nt)))
        return dict(built=bool(b.value), nblk=nb.value, unique_per_nnz=u.value, us=us[0], us_nt=us[1], nt=bool(nt.value))

    def reorder_info(self):
        """dict(reordered, block, spread_before, spread_after, us_natural, us_reordered) — mi_csr_reorder_info."""
        r, b = _c.c_int(), _c.c_int()
        v = [_c.c_double() for _ in range(4)]
        check(lib().mi_csr_reorder_info(self.handle, _c.byref(r), _c.byref(b), *[_c.byref(t) for t in v]))
        return dict(reordered=bool(r.value), block=b.value, spread_before=v[0].value, spread_after=v[1].value,
                    us_natural=v[2].value, us_reordered=v[3].value)

    def spmk_info(self, k):
        """dict(eligible, one_launch, us_k_launches, us_one_launch): how this handle runs the k-step (mi_csr_spmk_info)."""
        e, o = _c.c_int(), _c.c_int()
        a, b = _c.c_double(), _c.c_double()
        check(lib().mi_csr_spmk_info(self.handle, int(k), _c.byref(e), _c.byref(o), _c.byref(a), _c.byref(b)))
        return dict(eligible=bool(e.value), one_launch=bool(o.value), us_k_launches=a.value, us_one_launch=b.value)

    def dot_in_epilogue(self):
        """True if SpMV_CSR_dot / SpMV_CSR_orthogonalize on this handle carry the dot inside the product's launch."""
        r = _c.c_int()
        check(lib().mi_csr_dot_epilogue_info(self.handle, _c.byref(r)))
        return bool(r.value)

    # -- the library's numbering (mi_csr_perm ...): permute once per solve, not once per product --------------------
    def perm(self):
        """(reordered, perm) with perm[old] = new — identity when the handle was not relabelled."""
        r = _c.c_int()
        p = np.empty(self.n, np.int32)
        check(lib().mi_csr_perm(self.handle, _c.byref(r), p.ctypes.data))
        return bool(r.value), p

    def to_internal(self, x, out=None):
        import torch
        out = torch.empty_like(x) if out is None else out
        check(lib().mi_vec_to_internal_dev(self.handle, _dev_ptr(x, self.n, "x"), _dev_ptr(out, self.n, "x_int"), _stream_ptr()))
        return out

    def from_internal(self, x_int, out=None):
        import torch
        out = torch.empty_like(x_int) if out is None else out
        check(lib().mi_vec_from_internal_dev(self.handle, _dev_ptr(x_int, self.n, "x_int"), _dev_ptr(out, self.n, "x"), _stream_ptr()))
        return out

    def update_values(self, coef):
        """New coefficients for the same pattern (mi_csr_update_values): numpy array (host) or CUDA tensor."""
        if _is_torch(coef):
            check(lib().mi_csr_update_values_dev(self.handle, _dev_ptr(coef, self.nnz, "coef"), _stream_ptr()))
        else:
            self.coef = _host_f64(coef, self.nnz, "coef")
            check(lib().mi_csr_update_values(self.handle, self.coef.ctypes.data))
        return self

    def drop_host_arrays(self):
        """Free the host copies of indcol/coef once the device handle exists (large benches)."""
        _ = self.handle
        self.indcol = self.indcol[:0]
        self.coef = self.coef[:0]

    def close(self):
        if self._h is not None:
            lib().mi_csr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class bcsr4x4_matrix:
    """struct bcsr4x4_matrix of mpk/SpMV.h:26-33 (row-major 4x4 blocks)."""

    def __init__(self, nrows, ptrow, indcol, coef, nbcols=None, layout="row"):
        """layout: "row" — blocks row-major as mpk/ stores them; "col" — column-major as PETSc's MATSEQBAIJ stores them
        (src/kernels/baij4_mad.c:73-76), transposed by the library on upload."""
        self.layout = {"row": 0, "col": 1}[layout]
        self.nrows = int(nrows)
        self.ptrow = np.ascontiguousarray(ptrow, dtype=np.int32)
        self.indcol = np.ascontiguousarray(indcol, dtype=np.int32)
        self.coef = np.ascontiguousarray(coef, dtype=np.float64)
        self.nblocks = len(self.indcol)
        self.nbcols = (int(self.indcol.max()) + 1 if self.nblocks else 0) if nbcols is None else int(nbcols)
        self.nbcols = max(self.nbcols, self.nrows)
        self._h = None

    @property
    def handle(self):
        if self._h is None:
            h = _vp()
            check(lib().mi_bcsr4_create_layout(self.nrows, self.nbcols, self.ptrow.ctypes.data, self.indcol.ctypes.data,
                                               self.coef.ctypes.data, self.layout, _c.byref(h)))
            self._h = h
        return self._h

    def update_values(self, coef):
        """New block values (same pattern, same layout as at construction): numpy array or CUDA tensor."""
        if _is_torch(coef):
            check(lib().mi_bcsr4_update_values_layout_dev(self.handle, _dev_ptr(coef, 16 * self.nblocks, "coef"), self.layout, _stream_ptr()))
        else:
            self.coef = _host_f64(coef, 16 * self.nblocks, "coef")
            check(lib().mi_bcsr4_update_values_layout(self.handle, self.coef.ctypes.data, self.layout))
        return self

    def close(self):
        if self._h is not None:
            lib().mi_bcsr4_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DistVector:
    """mi_dist_vec_t: a vector distributed like the rows of a DistMatrix (per rank [owned | halo] on the rank's device)."""

    def __init__(self, D, host=None):
        self.D = D
        h = _vp()
        check(lib().mi_dist_vec_create(D.handle, _c.byref(h)))
        self._h = h
        if host is not None:
            self.set(host)

    def set(self, host):
        host = _host_f64(host, self.D.n, "x")
        check(lib().mi_dist_vec_set(self._h, host.ctypes.data))
        return self

    def get(self, out=None):
        out = np.empty(self.D.n, np.float64) if out is None else _host_f64(out, self.D.n, "out", writable=True)
        check(lib().mi_dist_vec_get(self._h, out.ctypes.data))
        return out

    def close(self):
        if self._h is not None:  # (safe after its DistMatrix is closed: mi_dist_destroy released the device memory, this frees the host object)
            lib().mi_dist_vec_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DistMatrix:
    """mi_dist_t (include/mi355_spmv.h): one csrmatrix row-partitioned over `ndev` GPUs behind ONE handle of ONE process — the
    form of the multi-GPU path the reference's single-process harnesses reach (SpMV_CSR / SpM4V / orthogonalize of mpk/SpMV.h
    through the shim with MI355_NGPUS).  Host vectors: spmv / spmk / dot / orthogonalize mirror the reference's calls;
    device-resident vectors: vector(), spmv_dev / spmk_dev / synchronize."""

    def __init__(self, ndev, n, ptrow, indcol, coef):
        self.ndev, self.n = int(ndev), int(n)
        ptrow = np.ascontiguousarray(ptrow, dtype=np.int32)
        indcol = np.ascontiguousarray(indcol, dtype=np.int32)
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        self.nnz = int(ptrow[-1]) if self.n > 0 else 0
        h = _vp()
        self._h = None
        check(lib().mi_dist_create(self.ndev, self.n, ptrow.ctypes.data, indcol.ctypes.data, coef.ctypes.data, _c.byref(h)))
        self._h = h

    @property
    def handle(self):
        return self._h

    def info(self):
        L = lib()
        nr, dd, ex, fu, ht, hm = _c.c_int(), _c.c_int(), _c.c_int(), _c.c_int(), _c.c_longlong(), _c.c_longlong()
        check(L.mi_dist_info(self._h, _c.byref(nr), _c.byref(dd), _c.byref(ex), _c.byref(fu), _c.byref(ht), _c.byref(hm)))
        ranks = []
        for r in range(nr.value):
            dev, r0, nl, nh, nz, st = _c.c_int(), _c.c_longlong(), _c.c_int(), _c.c_int(), _c.c_longlong(), _vp()
            check(L.mi_dist_rank_info(self._h, r, _c.byref(dev), _c.byref(r0), _c.byref(nl), _c.byref(nh), _c.byref(nz), _c.byref(st)))
            ranks.append(dict(device=dev.value, row_start=r0.value, n_local=nl.value, n_halo=nh.value, nnz_local=nz.value))
        return dict(nranks=nr.value, distinct_devices=dd.value, exchange=L.mi_dist_exchange_name(self._h).decode(),
                    fused=bool(fu.value), halo_total=ht.value, halo_max=hm.value, note=L.mi_dist_exchange_note(self._h).decode(),
                    ranks=ranks)

    def update_values(self, coef):
        coef = _host_f64(coef, self.nnz, "coef")
        check(lib().mi_dist_update_values(self._h, coef.ctypes.data))

    # -- host vectors: the reference's calling convention
    def spmv(self, y, x):
        x, y = _host_f64(x, self.n, "x"), _host_f64(y, self.n, "y", writable=True)
        check(lib().mi_dist_spmv(self._h, x.ctypes.data, y.ctypes.data))
        return y

    def spmk(self, ys, x):
        x = _host_f64(x, self.n, "x")
        ys = [_host_f64(t, self.n, "y_out", writable=True) for t in ys]
        arr = (_vp * len(ys))(*[t.ctypes.data for t in ys])
        check(lib().mi_dist_spmk(self._h, len(ys), x.ctypes.data, arr))
        return ys

    def dot(self, x, y):
        out = _c.c_double()
        check(lib().mi_dist_dot(self._h, _host_f64(x, self.n).ctypes.data, _host_f64(y, self.n).ctypes.data, _c.byref(out)))
        return out.value

    def orthogonalize(self, b, x1, x3, alpha=1e-8):
        beta = _c.c_double()
        x3 = _host_f64(x3, self.n, "x3", writable=True)
        check(lib().mi_dist_orthogonalize(self._h, _host_f64(b, self.n).ctypes.data, _host_f64(x1, self.n).ctypes.data, x3.ctypes.data,
                                          float(alpha), _c.byref(beta)))
        return beta.value

    # -- device-resident vectors
    def vector(self, host=None):
        return DistVector(self, host)

    def spmv_dev(self, y, x):
        check(lib().mi_dist_spmv_dev(self._h, x._h, y._h))

    def spmk_dev(self, ys, x):
        arr = (_vp * len(ys))(*[t._h for t in ys])
        check(lib().mi_dist_spmk_dev(self._h, len(ys), x._h, arr))

    def dot_dev(self, a, b):
        out = _c.c_double()
        check(lib().mi_dist_dot_dev(self._h, a._h, b._h, _c.byref(out)))
        return out.value

    def orthogonalize_dev(self, b, x1, x3, alpha=1e-8, want_beta=True):
        """x3 = x1 - alpha (b . x1) b on distributed vectors.  want_beta=False: nothing is awaited (the ranks' partial dots meet on the
        devices; returns None); else the call waits for rank 0's copy of beta and returns it."""
        if not want_beta:
            check(lib().mi_dist_orthogonalize_dev(self._h, b._h, x1._h, x3._h, float(alpha), None))
            return None
        beta = _c.c_double()
        check(lib().mi_dist_orthogonalize_dev(self._h, b._h, x1._h, x3._h, float(alpha), _c.byref(beta)))
        return beta.value

    def synchronize(self):
        check(lib().mi_dist_synchronize(self._h))

    def close(self):
        if self._h is not None:
            lib().mi_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def COO2CSR(nrow, irow, jcol, val):
    """COO -> csrmatrix with the reference's rules (mpk/utils.cpp:5-43, :97-127):
    columns ascending per row, the FIRST of duplicated (i, j) entries wins.
    Host-side integer work (numpy): a stable sort on (row, col) keeps COO order
    among duplicates, so the first survivor of each run is the first COO entry."""
    irow = np.asarray(irow, dtype=np.int64)
    jcol = np.asarray(jcol, dtype=np.int64)
    val = np.asarray(val, dtype=np.float64)
    order = np.lexsort((np.arange(len(irow)), jcol, irow))
    r, c, v = irow[order], jcol[order], val[order]
    keep = np.ones(len(r), bool)
    keep[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
    r, c, v = r[keep], c[keep], v[keep]
    ptrow = np.zeros(nrow + 1, np.int64)
    np.add.at(ptrow, r + 1, 1)
    ptrow = np.cumsum(ptrow)
    return csrmatrix(nrow, ptrow.astype(np.int32), c.astype(np.int32), v)


class _PlacedBuffer:
    """A device vector allocated by mi_vec_alloc_placed, visible to torch through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.ptr = ptr
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2, "strides": None}

    def __del__(self):
        try:
            if self.ptr:
                lib().mi_vec_free_placed(self.ptr)
                self.ptr = 0
        except Exception:
            pass


# --------------------------------------------------------------------- SpMV

def SpMV_CSR(y, x, A):
    """y = A x — SpMV_CSR(double* y, double* x, csrmatrix& A), mpk/SpMV.cpp:6-20.
    y is fully overwritten.  Tensors on the GPU: asynchronous on torch's current
    stream; numpy arrays: copied in/out and synchronised."""
    if _is_torch(x):
        check(lib().mi_spmv_dev(A.handle, _dev_ptr(x, A.ncols, "x"), _dev_ptr(y, A.n if A.rowmap is None else None, "y"),
                                _stream_ptr()))
    else:
        xx = _host_f64(x, A.ncols, "x")
        yy = _host_f64(y, A.n, "y", writable=True)
        check(lib().mi_spmv(A.handle, xx.ctypes.data, yy.ctypes.data))
    return y


def SpMV_CSR_internal(y_int, x_int, A):
    """y_int = A x_int with both vectors in the library's numbering (csrmatrix.to_internal / from_internal): what a Krylov
    loop calls between its one permutation in and its one permutation out.  Device tensors only."""
    check(lib().mi_spmv_internal_dev(A.handle, _dev_ptr(x_int, A.n, "x_int"), _dev_ptr(y_int, A.n, "y_int"), _stream_ptr()))
    return y_int


# The reference's four CSR variants differ only in how the CPU is driven
# (scalar / compiler FMA / __builtin_fma / AVX2, mpk/SpMV.cpp:23-85); on the GPU
# they are one kernel, bit-equal to _OPT/_FMA.
SpMV_CSR_OPT = SpMV_CSR
SpMV_CSR_FMA = SpMV_CSR
SpMV_CSR_AVX2 = SpMV_CSR


def SpMV_BCSR(y, x, A):
    """y = A x for bcsr4x4_matrix — SpMV_BCSR*, mpk/SpMV.cpp:90-219."""
    if _is_torch(x):
        check(lib().mi_bcsr4_spmv_dev(A.handle, _dev_ptr(x, 4 * A.nbcols, "x"), _dev_ptr(y, 4 * A.nrows, "y"), _stream_ptr()))
    else:
        xx = _host_f64(x, None, "x")
        if xx.size < 4 * A.nbcols:  # the reference reads x by block column; pad like its callers' vectors
            xx = np.concatenate([xx, np.zeros(4 * A.nbcols - xx.size)])
        yy = _host_f64(y, 4 * A.nrows, "y", writable=True)
        check(lib().mi_bcsr4_spmv(A.handle, xx.ctypes.data, yy.ctypes.data))
    return y


SpMV_BCSR_OPT = SpMV_BCSR
SpMV_BCSR_FMA = SpMV_BCSR
SpMV_BCSR_AVX2 = SpMV_BCSR


# ------------------------------------------------------------ matrix powers

def SpMkV(ys, x, A):
    """ys[p] = A^(p+1) x for p = 0..k-1 (all intermediate powers, as SpM4V returns them)."""
    k = len(ys)
    if _is_torch(x):
        ptrs = (_vp * k)(*[_dev_ptr(t, A.n, f"y{p + 1}").value for p, t in enumerate(ys)])
        check(lib().mi_spmk_dev(A.handle, k, _dev_ptr(x, A.ncols, "x"), ptrs, _stream_ptr()))
    else:
        xx = _host_f64(x, A.ncols, "x")
        outs = [_host_f64(t, A.n, f"y{p + 1}", writable=True) for p, t in enumerate(ys)]
        ptrs = (_vp * k)(*[o.ctypes.data for o in outs])
        check(lib().mi_spmk(A.handle, k, xx.ctypes.data, ptrs))
    return ys


def Generate1stlayer(ptrowend1, A):
    """mpk/SpM2V.cpp:5-26: ptrowend1[ia] = ptrow[j + 1] the first time column j = indcol[ia] is met in
    CSR traversal order, else ptrow[j].  The table drives the CPU's serial traversal; the GPU path does
    not need it (each power is a full row-parallel sweep) but callers may inspect it, so it is filled
    as the reference fills it.  ptrowend1: int32 array of >= nnz entries (or None: a new one)."""
    nnz = A.nnz
    if ptrowend1 is None:
        ptrowend1 = np.zeros(nnz, np.int32)
    cols = A.indcol[:nnz].astype(np.int64)
    ptrowend1[:nnz] = A.ptrow[cols]
    _, first = np.unique(cols, return_index=True)
    ptrowend1[first] = A.ptrow[cols[first] + 1]
    return ptrowend1


def SpM2V_CSR(z, y, x, A, ptrowend1=None):
    """y = A x, z = A (A x) — SpM2V_CSR(z, y, x, A, ptrowend1), mpk/SpM2V.cpp:79-112."""
    SpMkV([y, z], x, A)
    return z, y


SpM2V_CSR_OPT = SpM2V_CSR
SpM2V_CSR_FMA = SpM2V_CSR
SpM2V_CSR_AVX2 = SpM2V_CSR


def SpM3V(w, z, y, x, A, ptrowend1=None, ptrowend2=None):
    """mpk/SpMVmulti0.cpp:132-155."""
    SpMkV([y, z, w], x, A)
    return w, z, y


def SpM4V(v, w, z, y, x, A, ptrowend1=None, ptrowend2=None, ptrowend3=None):
    """y=Ax, z=A^2x, w=A^3x, v=A^4x — SpM4V(v, w, z, y, x, A, ...), mpk/SpMVmulti0.cpp:189-221."""
    SpMkV([y, z, w, v], x, A)
    return v, w, z, y


def SpM2V_BCSR(z, y, x, A, ptrowend1=None):
    """y = A x, z = A (A x) on a bcsr4x4_matrix — SpM2V_BCSR{,_OPT,_FMA,_AVX2}(z, y, x, A, ptrowend1),
    mpk/SpM2V.cpp:375-801 (square matrices)."""
    n = 4 * A.nrows
    if _is_torch(x):
        ptrs = (_vp * 2)(_dev_ptr(y, n, "y").value, _dev_ptr(z, n, "z").value)
        check(lib().mi_bcsr4_spmk_dev(A.handle, 2, _dev_ptr(x, n, "x"), ptrs, _stream_ptr()))
    else:
        yy, zz = _host_f64(y, n, "y", writable=True), _host_f64(z, n, "z", writable=True)
        ptrs = (_vp * 2)(yy.ctypes.data, zz.ctypes.data)
        check(lib().mi_bcsr4_spmk(A.handle, 2, _host_f64(x, n, "x").ctypes.data, ptrs))
    return z, y


SpM2V_BCSR_OPT = SpM2V_BCSR
SpM2V_BCSR_FMA = SpM2V_BCSR
SpM2V_BCSR_AVX2 = SpM2V_BCSR


# ------------------------------------------------- multi-vector products, Krylov basis

ARITH = {"chain": 0, "blockacc": 1}


def MatMatMult_SeqBAIJ_4(A, X, Y, arith="chain"):
    """Y[:, j] = A X[:, j] for the s columns of X with the matrix read once — MatMatMult_SeqBAIJ_4_AVX2(A, X, Y, s_step),
    src/kernels/spmm_avx2.c:7-109.  A: bcsr4x4_matrix (any arith) or csrmatrix (chain).  X, Y: column-major (n, s): CUDA
    tensors of shape (s, n) (row j = column j, as MatDense stores it) or numpy arrays of shape (s, n)."""
    s = int(X.shape[0])
    if isinstance(A, csrmatrix):
        assert arith == "chain"
        check(lib().mi_spmm_dev(A.handle, s, _dev_ptr(X), int(X.stride(0)), _dev_ptr(Y), int(Y.stride(0)), _stream_ptr()))
        return Y
    if _is_torch(X):
        check(lib().mi_bcsr4_spmm_dev(A.handle, s, _dev_ptr(X), int(X.stride(0)), _dev_ptr(Y), int(Y.stride(0)), ARITH[arith], _stream_ptr()))
    else:
        XX = np.ascontiguousarray(X, dtype=np.float64)
        assert isinstance(Y, np.ndarray) and Y.flags.c_contiguous and Y.dtype == np.float64
        check(lib().mi_bcsr4_spmm(A.handle, s, XX.ctypes.data, XX.shape[1], Y.ctypes.data, Y.shape[1], ARITH[arith]))
    return Y


def BuildKrylovBasis(A, v0, s, orth=False):
    """orth=False: V[0] = v0, V[k+1] = A V[k], k < s — BuildKrylovBasis_AVX2, src/kernels/spmm_avx2.c:112-168.
    orth=True: the orthonormal (Arnoldi) basis of the same space (mi_krylov_basis_dev explains).  Returns (V, H, nrm0):
    V a CUDA tensor of shape (s+1, n) (row k = basis vector k); H (s, s+2) with H[k, :k+1] the dots of step k and
    H[k, k+1] the norm, nrm0 = ||v0|| (both None when orth is False)."""
    import torch
    n = A.n
    V = torch.empty((s + 1, n), dtype=torch.float64, device="cuda")
    coef = torch.zeros(s * (s + 2) + 1, dtype=torch.float64, device="cuda") if orth else None
    check(lib().mi_krylov_basis_dev(A.handle, s, _dev_ptr(v0, n), _dev_ptr(V), n, 1 if orth else 0,
                                    _dev_ptr(coef) if orth else None, _stream_ptr()))
    if not orth:
        return V, None, None
    return V, coef[: s * (s + 2)].reshape(s, s + 2), coef[s * (s + 2)]


# ------------------------------------------------------------------- BLAS-1

def _scalar_dev():
    import torch

    return torch.empty(1, dtype=torch.float64, device="cuda")


def dot(x, y):
    """sum x_i y_i (std::inner_product at mpk/SpMVmulti.cpp:147).  Device tensors: returns a 1-element device tensor."""
    n = int(x.numel() if _is_torch(x) else np.size(x))
    if _is_torch(x):
        out = _scalar_dev()
        check(lib().mi_dot_dev(n, _dev_ptr(x), _dev_ptr(y, n), _dev_ptr(out), _stream_ptr()))
        return out
    out = _c.c_double(0)
    check(lib().mi_dot(n, _host_f64(x).ctypes.data, _host_f64(y, n).ctypes.data, _c.byref(out)))
    return out.value


def axpy(a, x, y):
    """y += a x, in place."""
    n = int(x.numel() if _is_torch(x) else np.size(x))
    if _is_torch(x):
        check(lib().mi_axpy_dev(n, float(a), _dev_ptr(x), _dev_ptr(y, n), _stream_ptr()))
    else:
        yy = _host_f64(y, n, "y", writable=True)
        check(lib().mi_axpy(n, float(a), _host_f64(x).ctypes.data, yy.ctypes.data))
    return y


def orthogonalize(nrow, b, x1, x3, alpha=1e-8):
    """x3 = x1 - alpha*(b.x1)*b — orthogonalize(nrow, b, x1, x3, alpha), mpk/SpMVmulti.cpp:146-151.
    Returns beta = b.x1 (float for numpy inputs, 1-element device tensor otherwise)."""
    if _is_torch(b):
        beta = _scalar_dev()
        check(lib().mi_orthogonalize_dev(nrow, _dev_ptr(b, nrow), _dev_ptr(x1, nrow), _dev_ptr(x3, nrow), float(alpha),
                                         _dev_ptr(beta), _stream_ptr()))
        return beta
    out = _c.c_double(0)
    x3h = _host_f64(x3, nrow, "x3", writable=True)
    check(lib().mi_orthogonalize(nrow, _host_f64(b, nrow).ctypes.data, _host_f64(x1, nrow).ctypes.data,
                                 x3h.ctypes.data, float(alpha), _c.byref(out)))
    return out.value


def SpMV_CSR_dot(y, x, A, b):
    """y = A x and beta = b . y in one pass (mi_spmv_dot_dev): returns beta as a 1-element device tensor."""
    beta = _scalar_dev()
    check(lib().mi_spmv_dot_dev(A.handle, _dev_ptr(x, A.ncols, "x"), _dev_ptr(y, A.n, "y"), _dev_ptr(b, A.n, "b"), _dev_ptr(beta), _stream_ptr()))
    return beta


def SpMV_CSR_orthogonalize(x1, x, A, b, x3, alpha=1e-8):
    """SpMV_CSR(x1, x, A); orthogonalize(n, b, x1, x3, alpha) — mpk/SpMVmulti.cpp:563-565 — as two launches: the dot rides in
    the product's epilogue (mi_spmv_orthogonalize_dev).  Returns beta (1-element device tensor)."""
    beta = _scalar_dev()
    check(lib().mi_spmv_orthogonalize_dev(A.handle, _dev_ptr(x, A.ncols, "x"), _dev_ptr(x1, A.n, "x1"), _dev_ptr(b, A.n, "b"),
                                          _dev_ptr(x3, A.n, "x3"), float(alpha), _dev_ptr(beta), _stream_ptr()))
    return beta


def orthonormalize_against_basis(basis, y):
    """orthonormalize_against_basis(nrow, basis, y), mpk/2SpMV.cpp:13-28: for each basis vector in turn
    y -= (y . v) v on the y updated so far (nothing is normalised, like the reference).  y is updated in
    place; returns the m coefficients (numpy array, or a device tensor for device inputs).
    basis: sequence of m vectors (numpy rows or CUDA tensors)."""
    m = len(basis)
    if _is_torch(y):
        import torch
        n = int(y.numel())
        dots = torch.empty(max(m, 1), dtype=torch.float64, device=y.device)
        ptrs = (_vp * max(m, 1))(*[_dev_ptr(b, n, f"basis[{j}]").value for j, b in enumerate(basis)])
        check(lib().mi_orthonormalize_against_basis_dev(n, m, ptrs, _dev_ptr(y), _dev_ptr(dots), _stream_ptr()))
        return dots[:m]
    yy = _host_f64(y, None, "y", writable=True)
    n = yy.size
    rows = [_host_f64(b, n, f"basis[{j}]") for j, b in enumerate(basis)]
    ptrs = (_vp * max(m, 1))(*[r.ctypes.data for r in rows])
    dots = np.zeros(m)
    check(lib().mi_orthonormalize_against_basis(n, m, ptrs, yy.ctypes.data, dots.ctypes.data if m else None))
    return dots


def norm2(x):
    """sqrt(sum x^2) — norm2, mpk/utils.cpp:131-136."""
    n = int(x.numel() if _is_torch(x) else np.size(x))
    if _is_torch(x):
        out = _scalar_dev()
        check(lib().mi_norm2_dev(n, _dev_ptr(x), _dev_ptr(out), _stream_ptr()))
        return out
    out = _c.c_double(0)
    check(lib().mi_norm2(n, _host_f64(x).ctypes.data, _c.byref(out)))
    return out.value


def rel_error(ref, test):
    """||ref - test||_2 / ||ref||_2 — rel_error, mpk/utils.cpp:138-143 (the parity metric)."""
    n = int(ref.numel() if _is_torch(ref) else np.size(ref))
    if _is_torch(ref):
        out = _scalar_dev()
        check(lib().mi_rel_error_dev(n, _dev_ptr(ref), _dev_ptr(test, n), _dev_ptr(out), _stream_ptr()))
        return out
    out = _c.c_double(0)
    check(lib().mi_rel_error(n, _host_f64(ref).ctypes.data, _host_f64(test, n).ctypes.data, _c.byref(out)))
    return out.value


def stream_read_us(nbytes, launches=20):
    """Microseconds per launch of a plain read sweep over nbytes of device memory (mi_stream_read_probe): this box's HBM rate."""
    us = _c.c_double()
    check(lib().mi_stream_read_probe(int(nbytes), int(launches), _c.byref(us)))
    return us.value


def flush_cache(sync=True):
    """mpk/utils.cpp:146-154 evicts the CPU caches before a timed call.  The GPU analogue evicts the L2s and the 256 MiB
    Infinity Cache (512 MiB device fill + 512 MiB read sweep).  sync=False: enqueued on the current stream without the
    closing synchronise, so that the next launch starts on cold caches but not on an idle GPU (mi_flush_cache_async)."""
    if sync:
        check(lib().mi_flush_cache())
    else:
        check(lib().mi_flush_cache_async(_stream_ptr()))
