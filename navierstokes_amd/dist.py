"""navierstokes_amd.dist — one matrix row-partitioned over the GPUs of a node.

New design (the reference is single-process).  One process per
GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the GPU box,
``gloo`` in the CPU tests).  Rank r owns global rows
``[row_starts[r], row_starts[r+1])`` and the same slice of x and y.  Per SpMV:

    1. pack      owned x entries other ranks need -> sendbuf   (HIP gather kernel)
    2. exchange  packed halos, point-to-point                   (RCCL, its own stream)
    3. interior  rows without ghost columns                     (overlaps 2.)
    4. boundary  rows with ghost columns, after the halos land

Three drivers of the same step: ``mi_part_spmv_dev`` (C++, grouped ncclSend/ncclRecv
on the partition's own stream, no per-step Python) when librccl resolves on every
rank; ``mi_part_spmv_push_dev`` (C++, peer-push windows over HIP IPC, no RCCL and no
second stream; ``exchange="push"`` / ``MI355_DIST_EXCHANGE=push``); else steps 1-4 from
here with one ``all_to_all_single``.  Whichever is chosen is first checked bit for
bit against the ``torch.distributed`` exchange, collectively.

The planner (column relabelling, interior/boundary split, send lists) is the
C++ code behind ``mi_part_*`` in the C-ABI; this module only moves the ids and
halo values between ranks.  Local compute goes through the C-ABI's device
kernels unless the caller injects ``compute=`` (the CPU tests inject the test
oracle there; product code never does).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import mpk

_c = ctypes
_vp = ctypes.c_void_p


def balanced_row_starts(n, nranks, nnz_per_row=None, align=1):
    """Contiguous row ranges with (nearly) equal nonzero counts.
    nnz_per_row: None (equal row counts) or an int64 array of row lengths.
    align: cut only at multiples of `align` rows (4 for FE matrices: a node's four dofs stay on one rank, so every rank's
    rows keep their 4x4 node-block structure)."""
    if nnz_per_row is None:
        cuts = np.array([(n * r) // nranks for r in range(1, nranks)], dtype=np.int64)
    else:
        cum = np.concatenate([[0], np.cumsum(np.asarray(nnz_per_row, dtype=np.int64))])
        targets = cum[-1] * np.arange(1, nranks) / nranks
        cuts = np.searchsorted(cum, targets, side="left").astype(np.int64)
    if align > 1:
        cuts = np.minimum((cuts + align // 2) // align * align, n // align * align)
    return np.concatenate([[0], cuts, [n]]).astype(np.int64)


class DistSetupError(RuntimeError):
    """Raised on EVERY rank when a set-up phase of DistCSR failed on ANY rank (the message names the first failing rank's
    error): a caller may catch it and go on collectively — nobody is left waiting in a collective for a rank that died."""


class DistCSR:
    """This rank's share of a row-partitioned csrmatrix."""
    ALLGATHER_HALO = 16384  # ghosts of the widest rank from which the native (RCCL) step all-gathers boundary slices

    def _agree(self, err, phase):
        """Collective: every rank learns whether `phase` failed anywhere.  err: None or this rank's exception."""
        if self.nranks == 1:
            if err is not None:
                raise err
            return
        mine = None if err is None else f"rank {self.rank}: {type(err).__name__}: {str(err)[:300]}"
        every = [None] * self.nranks
        dist.all_gather_object(every, mine, group=self.group)
        bad = [e for e in every if e is not None]
        if bad:
            raise DistSetupError(f"DistCSR set-up failed in phase '{phase}' on {len(bad)} of {self.nranks} ranks; first: {bad[0]}")

    def __init__(self, row_starts, ptrow, indcol_global, coef, group=None, device=None, compute=None, kernel=None, exchange=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nranks = dist.get_world_size(group) if dist.is_initialized() else 1
        self._h = None
        # Set-up is a sequence of LOCAL phases (library calls that can fail on one rank alone: an allocation, a plan) and
        # COLLECTIVE ones; each local phase ends with _agree(), so that a failure anywhere raises DistSetupError everywhere
        # instead of leaving the other ranks in the next collective until the process group times out.
        err = None
        try:
            self._init_local(row_starts, ptrow, indcol_global, coef, device, compute)
        except Exception as e:  # noqa: BLE001
            err = e
        self._agree(err, "plan")
        self._init_exchange_ids()
        err = None
        try:
            self._init_finalize(compute, kernel, exchange)
        except Exception as e:  # noqa: BLE001
            err = e
        self._agree(err, "finalize")
        self._init_choose_exchange(compute)

    def _init_local(self, row_starts, ptrow, indcol_global, coef, device, compute):
        self.row_starts = np.ascontiguousarray(row_starts, dtype=np.int64)
        assert len(self.row_starts) == self.nranks + 1
        ptrow = np.ascontiguousarray(ptrow, dtype=np.int32)
        indcol_global = np.ascontiguousarray(indcol_global, dtype=np.int32)
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        L = mpk.lib()
        h = _vp()
        mpk.check(L.mi_part_create(self.nranks, self.rank, self.row_starts.ctypes.data, ptrow.ctypes.data,
                                   indcol_global.ctypes.data, coef.ctypes.data, _c.byref(h)))
        self._h = h
        nl, nh, ni, nb = _c.c_int(), _c.c_int(), _c.c_int(), _c.c_int()
        mpk.check(L.mi_part_sizes(h, _c.byref(nl), _c.byref(nh), _c.byref(ni), _c.byref(nb)))
        self.n_local, self.n_halo, self.n_interior, self.n_boundary = nl.value, nh.value, ni.value, nb.value
        self.nnz_local = int(ptrow[-1])
        # `compute=` is a TEST seam (the CPU tests inject the oracle as local compute to exercise planner + exchange without a
        # GPU).  It must never carry a product run: where a GPU is present it is refused unless the test harness says so.
        if compute is not None and torch.cuda.is_available():
            import os
            if os.environ.get("MI355_TEST_COMPUTE_HOOK") != "1":
                raise RuntimeError("DistCSR(compute=...) is a test-only hook and is refused on a machine with a GPU")
        self.compute = compute
        self.device = torch.device(device) if device is not None else torch.device("cuda" if compute is None else "cpu")

        # who needs what: recv_counts[p] ids I need from p; tell every owner
        rc = np.zeros(self.nranks, np.int32)
        mpk.check(L.mi_part_recv_counts(h, rc.ctypes.data))
        self.recv_counts = [int(v) for v in rc]
        recv_ids = self._recv_ids = np.empty(self.n_halo, np.int64)
        off = 0
        for p in range(self.nranks):
            if rc[p]:
                mpk.check(L.mi_part_recv_ids(h, p, recv_ids[off:].ctypes.data))
                off += int(rc[p])

    def _init_exchange_ids(self):
        """Collective: tell every owner which of its entries this rank needs."""
        L, h, group, recv_ids = mpk.lib(), self._h, self.group, self._recv_ids
        self._nccl = bool(dist.is_initialized() and dist.get_backend(group) == "nccl")
        comm_dev = self.device if self._nccl else torch.device("cpu")
        if self.nranks > 1:
            t_rc = torch.tensor(self.recv_counts, dtype=torch.int64, device=comm_dev)
            t_sc = torch.empty_like(t_rc)
            dist.all_to_all_single(t_sc, t_rc, group=group)
            self.send_counts = [int(v) for v in t_sc.cpu()]
            t_ids_in = torch.from_numpy(recv_ids).to(comm_dev)
            t_ids_out = torch.empty(sum(self.send_counts), dtype=torch.int64, device=comm_dev)
            dist.all_to_all_single(t_ids_out, t_ids_in, self.send_counts, self.recv_counts, group=group)
            send_ids = t_ids_out.cpu().numpy()
            self._send_ids = send_ids
        else:
            self.send_counts = [0]
            self._send_ids = None
        self.n_send = sum(self.send_counts)


    def _init_finalize(self, compute, kernel, exchange):
        """Local: send lists into the plan, pieces to the device."""
        L, h = mpk.lib(), self._h
        if self._send_ids is not None:
            off = 0
            for p in range(self.nranks):
                c = self.send_counts[p]
                ids = np.ascontiguousarray(self._send_ids[off:off + c])
                mpk.check(L.mi_part_set_send_ids(h, p, c, ids.ctypes.data if c else None))
                off += c
        self._send_ids = self._recv_ids = None
        if compute is None:
            mpk.check(L.mi_part_finalize(h))
            if kernel is not None:
                mpk.check(L.mi_part_set_kernel(h, mpk.KERNELS[kernel] if isinstance(kernel, str) else int(kernel)))
            self._send_idx = None
            import os
            # which exchange drives the step (collective choice; every candidate is self-checked against the
            # torch.distributed exchange before it is trusted): "native" = C++ step over RCCL send/recv (default where
            # librccl resolves), "push" = peer-push windows over HIP IPC, no RCCL (mi_part_spmv_push_dev), "torch"
            self.push_fused = False
            # "auto" (default): push, else native, else torch — each step down only after the collective self-check below
            self.exchange = exchange or os.environ.get("MI355_DIST_EXCHANGE", "auto")
            # "allgather" = native with the all-gather form of the RCCL exchange forced (mi_part_allgather_setup); "native" under
            # "auto" takes that form by itself when the widest rank's halo has >= ALLGATHER_HALO ghosts (FE slab partitions)
            assert self.exchange in ("auto", "native", "allgather", "push", "torch"), self.exchange
            self.push = self.native = False
        else:
            self.native = False
            self.push = False
            self.push_fused = False
            tot, ptr = _c.c_int(), _vp()
            mpk.check(L.mi_part_send_index(h, _c.byref(tot), _c.byref(ptr)))
            self._send_idx = (np.ctypeslib.as_array(_c.cast(ptr, _c.POINTER(_c.c_int)), shape=(tot.value,)).copy()
                              if tot.value else np.zeros(0, np.int32))

    def _init_choose_exchange(self, compute):
        """Collective: bring the requested exchange up; each candidate is self-checked against the torch.distributed exchange."""
        self.rccl_ranks = 0
        if compute is None:
            self.push = self.exchange in ("auto", "push") and self._try_push_exchange()
        self.sendbuf = torch.empty(max(self.n_send, 1), dtype=torch.float64, device=self.device)
        import os
        if self.push and not self._native_selfcheck():
            self.push = self.push_fused = False
            mpk.lib().mi_part_push_disable(self._h)
        self.allgather = False
        if compute is None and not self.push and self.exchange in ("auto", "native", "allgather"):
            self.native = self._try_native_exchange()
        if self.native and not self._native_selfcheck():
            self.native = False  # collective decision: every rank falls back to the torch.distributed exchange
        elif not self.native and compute is None and self.nranks > 1 and os.environ.get("MI355_DIST_FORCE_SELFCHECK") == "1":
            assert self._native_selfcheck()  # tests: run the check's own code on a backend without RCCL

    def _try_native_exchange(self):
        """Set up the C++/RCCL step (mi_part_spmv_dev).  Every decision is collective: either all
        ranks switch to the native path or none does.  MI355_DIST_NATIVE=0 keeps torch.distributed."""
        import os
        if self.nranks == 1 or os.environ.get("MI355_DIST_NATIVE", "1") == "0":
            return False
        if not self._nccl:
            return False
        L = mpk.lib()
        flag = torch.tensor([1 if L.mi_comm_available() == 0 else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag) == 0:
            return False
        idt = torch.zeros(128, dtype=torch.uint8, device=self.device)
        ok0 = torch.zeros(1, dtype=torch.int32, device=self.device)
        if self.rank == 0:
            buf = _c.create_string_buffer(128)
            if L.mi_comm_unique_id(buf) == 0:
                idt.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
                ok0.fill_(1)
        dist.broadcast(ok0, src=0, group=self.group)
        if int(ok0) == 0:
            return False
        dist.broadcast(idt, src=0, group=self.group)
        rc = self._native_init(bytes(idt.cpu().numpy().tobytes()))  # collective inside RCCL
        if rc == 0:
            cnt = _c.c_int()
            if L.mi_part_comm_info(self._h, _c.byref(cnt), None) == 0:
                self.rccl_ranks = cnt.value  # ncclCommCount of the communicator just made
        flag.fill_(1 if rc == 0 and self.rccl_ranks in (0, self.nranks) else 0)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag) != 1:
            return False
        # the all-gather form for wide halos: every rank's union of send lists (as global ids) goes to every rank
        hmax = torch.tensor([self.n_halo], dtype=torch.int64, device=self.device)
        dist.all_reduce(hmax, op=dist.ReduceOp.MAX, group=self.group)
        import os
        forced = os.environ.get("MI355_PART_EXCHANGE")
        want = (self.exchange == "allgather" or forced == "allgather" or
                (self.exchange == "auto" and forced != "sendrecv" and int(hmax) >= self.ALLGATHER_HALO))
        if want:
            cnt, ptr = _c.c_int(), _vp()
            rc = L.mi_part_send_union(self._h, _c.byref(cnt), _c.byref(ptr))
            mine = (np.ctypeslib.as_array(_c.cast(ptr, _c.POINTER(_c.c_int)), shape=(cnt.value,)).astype(np.int64) + int(self.row_starts[self.rank])
                    if rc == 0 and cnt.value else np.zeros(0, np.int64))
            every = [None] * self.nranks
            dist.all_gather_object(every, (rc, mine), group=self.group)
            ok = all(e[0] == 0 for e in every)
            if ok:
                counts = np.ascontiguousarray([len(e[1]) for e in every], dtype=np.int32)
                ids = np.ascontiguousarray(np.concatenate([e[1] for e in every] + [np.zeros(0, np.int64)]), dtype=np.int64)
                ok = L.mi_part_allgather_setup(self._h, counts.ctypes.data, ids.ctypes.data) == 0 and L.mi_part_set_allgather(self._h, 1) == 0
            flag.fill_(1 if ok else 0)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if int(flag) != 1:  # the form is collective: nobody uses it unless everybody can
                L.mi_part_set_allgather(self._h, 0)
                if self.exchange == "allgather":
                    return False
            else:
                self.allgather = True
        return True

    def _try_push_exchange(self):
        """Set up the peer-push exchange (include/mi355_spmv.h: mi_part_push_*): every rank exports its receive window,
        the IPC handles and layouts are all-gathered, every rank maps its neighbours' windows.  Collective."""
        if self.nranks == 1:
            return False
        L = mpk.lib()
        handle = _c.create_string_buffer(64)
        layout = np.zeros(2 * self.nranks + 1, np.int64)
        rc = L.mi_part_push_export(self._h, handle, layout.ctypes.data)
        mine = (rc, bytes(handle.raw), layout.tolist())
        everyone = [None] * self.nranks
        dist.all_gather_object(everyone, mine, group=self.group)
        ok = all(e[0] == 0 for e in everyone)
        if not ok:  # nothing was connected anywhere: the windows some ranks did allocate go away
            L.mi_part_push_disable(self._h)
            return False
        if ok:
            handles = b"".join(e[1] for e in everyone)
            layouts = np.ascontiguousarray([e[2] for e in everyone], dtype=np.int64)
            ok = L.mi_part_push_connect(self._h, _c.create_string_buffer(handles, len(handles)), layouts.ctypes.data) == 0
        flags = [None] * self.nranks
        dist.all_gather_object(flags, bool(ok), group=self.group)
        if not all(flags):
            L.mi_part_push_disable(self._h)
        if all(flags):
            # one form for everybody: the one-launch step only if EVERY rank can run it (it depends on each rank's measured
            # kernel).  Mixed forms are legal (tests/test_push_protocol.py); equal forms keep the ranks' step times balanced.
            fused = _c.c_int()
            rc = L.mi_part_push_info(self._h, None, _c.byref(fused), None)
            every = [None] * self.nranks
            dist.all_gather_object(every, (rc == 0, bool(fused.value)), group=self.group)
            want = all(f for _, f in every)
            rc2 = L.mi_part_push_unfuse(self._h) if (rc == 0 and fused.value and not want) else 0
            fine = [None] * self.nranks
            dist.all_gather_object(fine, rc == 0 and rc2 == 0, group=self.group)
            if not all(fine):
                L.mi_part_push_disable(self._h)
                return False
            self.push_fused = bool(fused.value) and want
        return all(flags)

    def update_values(self, coef):
        """New coefficients (this rank's rows, same order as at construction) for the same pattern."""
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        assert len(coef) == self.nnz_local
        mpk.check(mpk.lib().mi_part_update_values(self._h, coef.ctypes.data))

    def status(self):
        """Raises if a hand-off / halo wait of the native or push step ever gave up (call after synchronising)."""
        if self.compute is None:
            mpk.check(mpk.lib().mi_part_status(self._h))

    def _native_selfcheck(self):
        """One product through the native C++ step and one through the torch.distributed exchange on the same
        (seeded) vector: the two must agree bit for bit on every rank, or nobody uses the native step."""
        ok = 1
        try:
            g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
            x_ext = self.new_x_ext()
            x_ext[: self.n_local] = torch.rand(self.n_local, generator=g, dtype=torch.float64).to(self.device)
            y_native = self.new_y()
            self.spmv(x_ext, y_native)
            torch.cuda.synchronize()
            halo_native = x_ext[self.n_local:].clone()
            x_ext[self.n_local:] = float("nan")
            was_native, was_push, self.native, self.push = self.native, self.push, False, False
            y_torch = self.new_y()
            self.spmv(x_ext, y_torch)
            torch.cuda.synchronize()
            self.native, self.push = was_native, was_push
            # (the one-launch push step reads ghosts straight from its window and leaves the halo part of x_ext alone)
            halo_ok = self.push_fused or torch.equal(halo_native, x_ext[self.n_local:])
            if not (torch.equal(y_native, y_torch) and halo_ok):
                ok = 0
        except Exception:  # noqa: BLE001 - any failure means: do not use it
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device if self._nccl else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag) == 1

    def _native_init(self, id128):
        """ncclCommInitRank for this partition from the 128-byte unique id (all ranks, collectively)."""
        buf = _c.create_string_buffer(bytes(id128), 128)
        return mpk.lib().mi_part_comm_init(self._h, buf)

    # -- host views of the two local pieces (CPU checks) -------------------------------------
    def local_piece(self, which):
        """(ptrow, indcol_local, coef, rowmap) of the interior (0) / boundary (1) rows, as numpy copies."""
        L = mpk.lib()
        n, p, c, v, m = _c.c_int(), _vp(), _vp(), _vp(), _vp()
        mpk.check(L.mi_part_local_csr(self._h, which, _c.byref(n), _c.byref(p), _c.byref(c), _c.byref(v), _c.byref(m)))
        nr = n.value
        ptrow = np.ctypeslib.as_array(_c.cast(p, _c.POINTER(_c.c_int)), shape=(nr + 1,)).copy()
        nnz = int(ptrow[-1])
        if nnz:
            col = np.ctypeslib.as_array(_c.cast(c, _c.POINTER(_c.c_int)), shape=(nnz,)).copy()
            val = np.ctypeslib.as_array(_c.cast(v, _c.POINTER(_c.c_double)), shape=(nnz,)).copy()
        else:
            col, val = np.zeros(0, np.int32), np.zeros(0)
        rmap = np.ctypeslib.as_array(_c.cast(m, _c.POINTER(_c.c_int)), shape=(nr,)).copy() if nr else np.zeros(0, np.int32)
        return ptrow, col, val, rmap

    # -- vectors ---------------------------------------------------------------------------------
    def new_x_ext(self):
        """[x_local | halo] buffer; fill the first n_local entries with the owned slice of x."""
        return torch.zeros(self.n_local + self.n_halo, dtype=torch.float64, device=self.device)

    def new_y(self):
        return torch.empty(max(self.n_local, 1), dtype=torch.float64, device=self.device)[: self.n_local]

    # -- y_local = (A x)_local ---------------------------------------------------------------------
    def spmv(self, x_ext, y_local, stream_ptr=None):
        """y_local = (A x)_local.  x_ext = [x_local | halo]; the halo part is (re)filled here.
        stream_ptr: raw hipStream_t of the stream torch is currently on (looked up once if omitted)."""
        L = mpk.lib()
        work = None
        dev = self.compute is None
        if dev:
            sp = stream_ptr if stream_ptr is not None else mpk._stream_ptr()
            px, py = _vp(x_ext.data_ptr()), _vp(y_local.data_ptr())
            if self.native:  # the whole step inside the library: pack, RCCL exchange, interior, boundary
                mpk.check(L.mi_part_spmv_dev(self._h, px, py, sp))
                return y_local
            if self.push:    # the whole step inside the library, no RCCL: push, interior, wait + copy, boundary
                mpk.check(L.mi_part_spmv_push_dev(self._h, px, py, sp))
                return y_local
        if self.nranks > 1:
            if dev:
                mpk.check(L.mi_part_pack_dev(self._h, px, _vp(self.sendbuf.data_ptr()), sp))
            elif self.n_send:
                self.sendbuf[: self.n_send] = x_ext[torch.from_numpy(self._send_idx.astype(np.int64))]
            if dev and not self._nccl:
                # a backend without device collectives (gloo; development runs with several ranks on
                # one card): stage the packed halos through host memory.  Correct, not fast.
                hs = self.sendbuf[: self.n_send].cpu()
                hr = torch.empty(self.n_halo, dtype=torch.float64)
                dist.all_to_all_single(hr, hs, self.recv_counts, self.send_counts, group=self.group)
                x_ext[self.n_local:].copy_(hr)
            else:
                work = dist.all_to_all_single(x_ext[self.n_local:], self.sendbuf[: self.n_send], self.recv_counts,
                                              self.send_counts, group=self.group, async_op=True)
        if dev:
            mpk.check(L.mi_part_spmv_interior_dev(self._h, px, py, sp))
        else:
            self.compute(self, 0, x_ext, y_local)
        if work is not None:
            work.wait()  # NCCL: the current stream waits for the exchange; no host block
        if dev:
            mpk.check(L.mi_part_spmv_boundary_dev(self._h, px, py, sp))
        else:
            self.compute(self, 1, x_ext, y_local)
        return y_local

    def refresh_halo(self, x_ext):
        """Fill the halo part of x_ext through the torch.distributed exchange (checks and tools: the one-launch push
        step never writes it)."""
        dev = self.compute is None
        if self.nranks == 1:
            return x_ext
        assert dev
        L = mpk.lib()
        mpk.check(L.mi_part_pack_dev(self._h, _vp(x_ext.data_ptr()), _vp(self.sendbuf.data_ptr()), mpk._stream_ptr()))
        if not self._nccl:
            hs = self.sendbuf[: self.n_send].cpu()
            hr = torch.empty(self.n_halo, dtype=torch.float64)
            dist.all_to_all_single(hr, hs, self.recv_counts, self.send_counts, group=self.group)
            x_ext[self.n_local:].copy_(hr)
        else:
            dist.all_to_all_single(x_ext[self.n_local:], self.sendbuf[: self.n_send], self.recv_counts, self.send_counts, group=self.group)
        return x_ext

    # -- y_1..y_k = A x .. A^k x -----------------------------------------------------------------
    def new_power_buffers(self, k):
        """k buffers shaped like x_ext: the owned part of buffer i receives A^(i+1) x, its halo part the
        ghosts of that power (needed by the next product)."""
        return [self.new_x_ext() for _ in range(k)]

    def spmk(self, x_ext, bufs, stream_ptr=None):
        """The matrix-powers step across ranks: bufs[i][:n_local] = (A^(i+1) x)_local for i < len(bufs),
        one halo exchange per power (SURVEY §8e "k exchanges").  Same vectors as the reference's fused
        SpM2V_CSR / SpM3V / SpM4V return on one CPU (mpk/SpM2V.cpp:79-112, mpk/SpMVmulti0.cpp:132-221):
        every row of every power is the same sequential fma chain.  Returns the owned views."""
        src = x_ext
        for b in bufs:
            self.spmv(src, b[: self.n_local], stream_ptr)
            src = b
        return [b[: self.n_local] for b in bufs]

    def dot(self, a_local, b_local):
        """Global dot product: local fixed-tree reduction, then one all_reduce of a double."""
        if self.compute is None:
            part = mpk.dot(a_local, b_local)
        else:
            part = torch.dot(a_local, b_local).reshape(1)
        if self.nranks > 1:
            if part.is_cuda and not self._nccl:
                host = part.cpu()
                dist.all_reduce(host, group=self.group)
                part.copy_(host)
            else:
                dist.all_reduce(part, group=self.group)
        return part

    def orthogonalize(self, b_local, x1_local, x3_local, alpha=1e-8):
        """x3 = x1 - alpha (b . x1) b across ranks — orthogonalize(n, b, x1, x3, alpha), mpk/SpMVmulti.cpp:146-151, with the dot made
        global: every rank's fixed-tree partial, one all_reduce of a double, then the reference's fused update on the owned slice,
        x3_i = fma(-(alpha * beta), b_i, x1_i) — bit-equal to the single-GPU update GIVEN beta (an axpy with a = -(alpha * beta) is that
        very fma).  Returns beta as a 1-element tensor.  x3 may be x1 (in place, the form of mpk/2SpMV.cpp:3-11)."""
        beta = self.dot(b_local, x1_local)
        a = -(float(alpha) * float(beta))  # (one host read of the reduced scalar: the all_reduce has synchronised the ranks anyway)
        if x3_local.data_ptr() != x1_local.data_ptr():
            x3_local.copy_(x1_local)
        if self.compute is None:
            mpk.axpy(a, b_local, x3_local)
        else:  # CPU ranks (gloo tests): the same fma, element by element
            from fractions import Fraction  # exact product and sum, one rounding: int / int division rounds correctly
            bb, xx = b_local.numpy(), x3_local.numpy()
            fa = Fraction(a)
            for i in range(len(xx)):
                xx[i] = float(fa * Fraction(float(bb[i])) + Fraction(float(xx[i])))
        return beta

    def close(self):
        if self._h is not None:
            mpk.lib().mi_part_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
