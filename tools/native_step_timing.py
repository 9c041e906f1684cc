"""Relative cost of the native multi-rank step with flag-kernel vs event hand-offs: two ranks as threads on one GPU,
tests/fake_rccl as the exchange (host-blocking, so absolute numbers are not those of RCCL; the GPU is shared by both
ranks).  Each rank owns 625 k rows, the size of one rank's piece of C4 at 8 GPUs.
  MI355_RCCL_LIBRARY=tests/fake_rccl/libfake_rccl.so MI355_PART_HANDOFF=flags|events python tools/native_step_timing.py"""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navierstokes_amd import dist as D, mpk, synth
vp = ctypes.c_void_p
N, n = 2, 1_250_000
torch.cuda.set_device(0)
L = mpk.lib(); mpk.check(L.mi_comm_available())
rs = D.balanced_row_starts(n, N)
parts, meta = [], []
for r in range(N):
    lo, hi = int(rs[r]), int(rs[r + 1])
    p, c, v = synth.rows("s15", n, lo, hi)
    h = vp(); mpk.check(L.mi_part_create(N, r, rs.ctypes.data, p.ctypes.data, c.ctypes.data, v.ctypes.data, ctypes.byref(h)))
    rc = np.zeros(N, np.int32); mpk.check(L.mi_part_recv_counts(h, rc.ctypes.data))
    parts.append(h); meta.append((lo, hi, rc))
for r in range(N):
    for q in range(N):
        cnt = int(meta[r][2][q]); ids = np.empty(max(cnt, 1), np.int64)
        if cnt: mpk.check(L.mi_part_recv_ids(parts[r], q, ids.ctypes.data))
        if q != r: mpk.check(L.mi_part_set_send_ids(parts[q], r, cnt, ids.ctypes.data))
for r in range(N):
    mpk.check(L.mi_part_set_send_ids(parts[r], r, 0, np.empty(1, np.int64).ctypes.data)); mpk.check(L.mi_part_finalize(parts[r]))
idbuf = ctypes.create_string_buffer(128); mpk.check(L.mi_comm_unique_id(idbuf))
out = [None] * N
def rank_main(r):
    torch.cuda.set_device(0)
    h, (lo, hi, _) = parts[r], meta[r]
    nl, nh = ctypes.c_int(), ctypes.c_int(); mpk.check(L.mi_part_sizes(h, ctypes.byref(nl), ctypes.byref(nh), None, None))
    mpk.check(L.mi_part_comm_init(h, ctypes.create_string_buffer(idbuf.raw, 128)))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        x = torch.zeros(nl.value + nh.value, dtype=torch.float64, device="cuda"); x[: nl.value] = torch.from_numpy(synth.x_sin(lo, hi)).cuda()
        y = torch.empty(nl.value, dtype=torch.float64, device="cuda"); sp = vp(st.cuda_stream)
        for _ in range(30): mpk.check(L.mi_part_spmv_dev(h, vp(x.data_ptr()), vp(y.data_ptr()), sp))
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(st)
        for _ in range(300): mpk.check(L.mi_part_spmv_dev(h, vp(x.data_ptr()), vp(y.data_ptr()), sp))
        e1.record(st); t1 = time.perf_counter(); st.synchronize()
        out[r] = (e0.elapsed_time(e1) / 300 * 1e3, (t1 - t0) / 300 * 1e6)
ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(N)]
[t.start() for t in ts]; [t.join(timeout=200) for t in ts]
print(f"NATIVE_STEP handoff={os.environ.get('MI355_PART_HANDOFF', 'flags')}: per rank (device us/step, host us/step) = {out}")
os._exit(0)
