"""Worker for test_gpu_parity.py::test_native_step_multirank_threads: N ranks as THREADS of this process on
cuda:0, the library's native C++ step (mi_part_comm_init + mi_part_spmv_dev) with tests/fake_rccl standing in
for librccl (MI355_RCCL_LIBRARY).  Everything but RCCL itself is the product path that runs on N GPUs: the
partition plan, send/recv offsets and counts, pack on the comm stream, interior rows beside the exchange,
boundary rows behind it, and the ordering between consecutive steps.  Checks, per rank, bitwise:
  * A x, A^2 x, A^3 x, A^4 x with a halo exchange per power (ping-pong [owned | halo] buffers),
  * 40 back-to-back repetitions of the same step into the same buffers (hazards between steps)."""
import ctypes
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from navierstokes_amd import dist as D  # noqa: E402
from navierstokes_amd import mpk, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)

vp = ctypes.c_void_p


def main():
    kind, n, w, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    push = os.environ.get("MI355_TEST_EXCHANGE") == "push"  # peer-push windows instead of the (stand-in) RCCL exchange
    assert push or os.environ.get("MI355_RCCL_LIBRARY", "").endswith("libfake_rccl.so")
    torch.cuda.set_device(0)
    L = mpk.lib()
    if not push:
        mpk.check(L.mi_comm_available())
    allgather = os.environ.get("MI355_TEST_ALLGATHER") == "1"  # the all-gather form of the RCCL step (wide halos)
    if kind == "fe":  # the FE matrix on an n^3-cell box: node-aligned cuts, a whole mesh plane per neighbour as halo
        Pg, Cg, Vg = synth.fe_matrix(n)
        n = len(Pg) - 1
        rs = D.balanced_row_starts(n, N, np.diff(Pg), align=4)
    else:
        Pg, Cg, Vg = synth.rows(kind, n, w=w)
        rs = D.balanced_row_starts(n, N, np.diff(Pg))
    # ---- setup on the main thread: plans, id exchange (all ranks live here), finalize
    parts, meta = [], []
    for r in range(N):
        lo, hi = int(rs[r]), int(rs[r + 1])
        if kind == "fe":
            p, c, v = (Pg[lo:hi + 1] - Pg[lo]).astype(np.int32), Cg[Pg[lo]:Pg[hi]].copy(), Vg[Pg[lo]:Pg[hi]].copy()
        else:
            p, c, v = synth.rows(kind, n, lo, hi, w=w)
        h = vp()
        mpk.check(L.mi_part_create(N, r, rs.ctypes.data, p.ctypes.data, c.ctypes.data, v.ctypes.data, ctypes.byref(h)))
        rc = np.zeros(N, np.int32)
        mpk.check(L.mi_part_recv_counts(h, rc.ctypes.data))
        parts.append(h)
        meta.append(dict(lo=lo, hi=hi, recv=rc))
    for r in range(N):          # rank r wants ids from q  ->  q learns what to send to r
        for q in range(N):
            cnt = int(meta[r]["recv"][q])
            ids = np.empty(max(cnt, 1), np.int64)
            if cnt:
                mpk.check(L.mi_part_recv_ids(parts[r], q, ids.ctypes.data))
            if q != r:
                mpk.check(L.mi_part_set_send_ids(parts[q], r, cnt, ids.ctypes.data))
    for r in range(N):
        mpk.check(L.mi_part_set_send_ids(parts[r], r, 0, np.empty(1, np.int64).ctypes.data))
        mpk.check(L.mi_part_finalize(parts[r]))
    idbuf = ctypes.create_string_buffer(128)
    if push:  # windows of ranks living in one process are connected through the library's registry, not hipIpcOpenMemHandle
        handles, layouts = b"", []
        for r in range(N):
            hb, lay = ctypes.create_string_buffer(64), np.zeros(2 * N + 1, np.int64)
            mpk.check(L.mi_part_push_export(parts[r], hb, lay.ctypes.data))
            handles += hb.raw
            layouts.append(lay)
        layouts = np.ascontiguousarray(layouts)
        for r in range(N):
            mpk.check(L.mi_part_push_connect(parts[r], ctypes.create_string_buffer(handles, len(handles)), layouts.ctypes.data))
        step_fn = L.mi_part_spmv_push_dev
    else:
        mpk.check(L.mi_comm_unique_id(idbuf))
        step_fn = L.mi_part_spmv_dev
    ag_counts = ag_ids = None
    if allgather:  # every rank's union of send lists as global ids (what a distributed host all-gathers by a side channel)
        cnts, lists = [], []
        for r in range(N):
            cnt, ptr = ctypes.c_int(), vp()
            mpk.check(L.mi_part_send_union(parts[r], ctypes.byref(cnt), ctypes.byref(ptr)))
            loc = (np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int)), shape=(cnt.value,)).astype(np.int64)
                   if cnt.value else np.zeros(0, np.int64))
            cnts.append(cnt.value)
            lists.append(loc + meta[r]["lo"])
        ag_counts = np.ascontiguousarray(cnts, dtype=np.int32)
        ag_ids = np.ascontiguousarray(np.concatenate(lists + [np.zeros(0, np.int64)]), dtype=np.int64)
    Y = O.spmk_chain(4, Pg, Cg, Vg, synth.x_sin(0, n))
    results = [None] * N

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            h, lo, hi = parts[r], meta[r]["lo"], meta[r]["hi"]
            nl, nh = ctypes.c_int(), ctypes.c_int()
            mpk.check(L.mi_part_sizes(h, ctypes.byref(nl), ctypes.byref(nh), None, None))
            nl, nh = nl.value, nh.value
            if not push:
                mpk.check(L.mi_part_comm_init(h, ctypes.create_string_buffer(idbuf.raw, 128)))  # collective over the threads
                if allgather:
                    mpk.check(L.mi_part_allgather_setup(h, ag_counts.ctypes.data, ag_ids.ctypes.data))
                    mpk.check(L.mi_part_set_allgather(h, 1))
                    use = ctypes.c_int()
                    mpk.check(L.mi_part_allgather_info(h, None, ctypes.byref(use), None))
                    assert use.value == 1
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                bufs = [torch.zeros(nl + nh, dtype=torch.float64, device="cuda") for _ in range(5)]
                bufs[0][:nl] = torch.from_numpy(synth.x_sin(lo, hi)).cuda()  # (x_sin of the global index: any rank can make its slice)
                sp = vp(st.cuda_stream)
                for k in range(4):  # powers: output of step k is the owned part of step k+1's input
                    mpk.check(step_fn(h, vp(bufs[k].data_ptr()), vp(bufs[k + 1].data_ptr()), sp))
                st.synchronize()
                ok = all(np.array_equal(bufs[k + 1][:nl].cpu().numpy().view(np.uint64), Y[k][lo:hi].view(np.uint64))
                         for k in range(4))
                y = torch.full((nl,), float("nan"), dtype=torch.float64, device="cuda")
                for _ in range(40):  # same buffers again and again, never synchronising in between
                    mpk.check(step_fn(h, vp(bufs[0].data_ptr()), vp(y.data_ptr()), sp))
                st.synchronize()
                mpk.check(L.mi_part_status(h))
                ok = ok and np.array_equal(y.cpu().numpy().view(np.uint64), Y[0][lo:hi].view(np.uint64))
            results[r] = ok
        except Exception as e:  # noqa: BLE001
            results[r] = repr(e)

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(N)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    alive = [t.is_alive() for t in ts]
    print(f"NATIVE_THREADS_RESULT ranks={N} results={results} alive={alive}")
    good = all(r is True for r in results) and not any(alive)
    if good:
        for h in parts:
            mpk.check(L.mi_part_destroy(h))
    sys.stdout.flush()
    os._exit(0 if good else 1)  # a stuck rank thread must not keep the process (and the GPU box) waiting


if __name__ == "__main__":
    main()
