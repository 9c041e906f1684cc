"""Dev tool: GPU-side time of ONE rank's share of the C4 matrix at N ranks (no exchange):
pack + interior + boundary kernels, to see what the step costs once the host is out of the way."""
import sys, os, ctypes, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# diagnostics live in the devtools build of the library (make -C navierstokes_amd/csrc devtools; include/mi355_devtools.h)
os.environ.setdefault("MI355_SPMV_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "navierstokes_amd", "csrc", "libmi355spmv_dev.so"))
os.environ.setdefault("MI355_PUSH_EXT_SPLIT", "0")  # (the looped-back window lives on this device, which otherwise reads as "ranks share a card")
os.environ["MI355_PUSH_LOOPBACK"] = "1"  # this tool maps its OWN window as every peer's (flags preset, nothing waits)
from navierstokes_amd import mpk, synth, dist as D
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
what = sys.argv[3] if len(sys.argv) > 3 else "c4"   # "c4": S15 5 M rows; "fe": the 68^3-cell FE matrix, cuts at node boundaries
if what in ("fe", "mesh"):  # "mesh": the P1 pressure operator on a 170^3-cell Kuhn mesh, natural node order (5 M rows; SIM_MESH_CELLS)
    P_, C_, V_ = synth.fe_matrix(int(os.environ.get("SIM_FE_CELLS", "68"))) if what == "fe" else synth.pressure_matrix(int(os.environ.get("SIM_MESH_CELLS", "170")))
    n = len(P_) - 1
    rs = D.balanced_row_starts(n, N, np.diff(P_), align=4 if what == "fe" else 1)
    lo, hi = int(rs[rank]), int(rs[rank + 1])
    p, c, v = (P_[lo:hi + 1] - P_[lo]).astype(np.int32), C_[P_[lo]:P_[hi]].copy(), V_[P_[lo]:P_[hi]].copy()
    del P_, C_, V_
else:
    n = 5_000_000
    rs = D.balanced_row_starts(n, N)
    lo, hi = int(rs[rank]), int(rs[rank + 1])
    p, c, v = synth.rows("s15", n, lo, hi)
L = mpk.lib()
h = ctypes.c_void_p()
mpk.check(L.mi_part_create(N, rank, rs.ctypes.data, p.ctypes.data, c.ctypes.data, v.ctypes.data, ctypes.byref(h)))
nl, nh, ni, nb = (ctypes.c_int() for _ in range(4))
mpk.check(L.mi_part_sizes(h, *(ctypes.byref(t) for t in (nl, nh, ni, nb))))
rc = np.zeros(N, np.int32); mpk.check(L.mi_part_recv_counts(h, rc.ctypes.data))
# pretend the neighbours ask for what a symmetric band would: my first/last rc entries
for q in range(N):
    cnt = int(rc[q])
    if cnt and q != rank:
        ids = (np.arange(cnt) + (lo if q < rank else hi - cnt)).astype(np.int64)
        mpk.check(L.mi_part_set_send_ids(h, q, cnt, ids.ctypes.data))
mpk.check(L.mi_part_finalize(h))
halo_ids = np.concatenate([np.empty(0, np.int64)] + [np.empty(int(rc[q]), np.int64) for q in range(N)])
off = 0
for q in range(N):
    if rc[q]:
        mpk.check(L.mi_part_recv_ids(h, q, halo_ids[off:].ctypes.data)); off += int(rc[q])
xe = np.concatenate([synth.x_sin(lo, hi), np.sin(0.001 * halo_ids)])
x_ext = torch.from_numpy(xe).cuda()
y = torch.full((nl.value,), float("nan"), dtype=torch.float64, device="cuda")
send = torch.empty(int(rc.sum()) + 1, dtype=torch.float64, device="cuda")
sp = mpk._stream_ptr(); vp = ctypes.c_void_p
contig = ctypes.c_int()
mpk.check(L.mi_part_sends_contiguous(h, ctypes.byref(contig)))
def step():  # the RCCL-shaped route on ONE stream: [pack ->] interior -> boundary (in the real step the boundary rows run on the comm stream beside the interior rows)
    if not contig.value:  # a banded partition's send lists are slices of x, sent in place (mi_part_spmv_dev: enqueue_exchange(d_x_direct)): no pack launch
        mpk.check(L.mi_part_pack_dev(h, vp(x_ext.data_ptr()), vp(send.data_ptr()), sp))
    mpk.check(L.mi_part_spmv_interior_dev(h, vp(x_ext.data_ptr()), vp(y.data_ptr()), sp))
    mpk.check(L.mi_part_spmv_boundary_dev(h, vp(x_ext.data_ptr()), vp(y.data_ptr()), sp))
REPS = int(os.environ.get("SIM_RANK_REPS", "3000"))
for _ in range(300): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(REPS): step()
e1.record(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / REPS * 1e6
gpu_step_us = e0.elapsed_time(e1) / REPS * 1e3
def only(fn, name):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS): fn()
    b.record(); torch.cuda.synchronize()
    print(f"    {name}: {a.elapsed_time(b) / REPS * 1e3:.1f} us back-to-back")
only(lambda: mpk.check(L.mi_part_pack_dev(h, vp(x_ext.data_ptr()), vp(send.data_ptr()), sp)), "pack")
only(lambda: mpk.check(L.mi_part_spmv_interior_dev(h, vp(x_ext.data_ptr()), vp(y.data_ptr()), sp)), "interior")
only(lambda: mpk.check(L.mi_part_spmv_boundary_dev(h, vp(x_ext.data_ptr()), vp(y.data_ptr()), sp)), "boundary")
# ---- the peer-push step of the same rank, looped back: all "peers" are this rank's own window (same offsets and counts a
# symmetric band gives them), every flag preset, so the four launches run back to back exactly as they would with the
# neighbours' data already there — pushes are real stores of the real entries, into local HBM instead of over xGMI
hb, lay = ctypes.create_string_buffer(64), np.zeros(2 * N + 1, np.int64)
mpk.check(L.mi_part_push_export(h, hb, lay.ctypes.data))
lays = []
for q in range(N):
    lq = np.zeros(2 * N + 1, np.int64)
    if q != rank and rc[q]:           # q receives from me what I receive from it (symmetric band)
        lq[0] = lay[0]                # looped back: its window is mine
        lq[1 + rank] = 0 if rank < q else int(lay[0]) - int(rc[q])   # left neighbour's entries first, right neighbour's last
        lq[1 + N + rank] = int(rc[q])
    lays.append(lq)
lays = np.ascontiguousarray(lays)
handles = hb.raw * N
mpk.check(L.mi_part_push_connect(h, ctypes.create_string_buffer(handles, len(handles)), lays.ctypes.data))
mpk.check(L.mi_part_push_debug_preset(h, 0x3fffffff))
def pstep():
    mpk.check(L.mi_part_spmv_push_dev(h, vp(x_ext.data_ptr()), vp(y.data_ptr()), sp))
halo_keep = x_ext[nl.value:].clone()
for _ in range(300): pstep()
torch.cuda.synchronize()
t0 = time.perf_counter(); e0.record()
for _ in range(REPS): pstep()
t_enq = (time.perf_counter() - t0) / REPS * 1e6
e1.record(); torch.cuda.synchronize()
pwall = (time.perf_counter() - t0) / REPS * 1e6
mpk.check(L.mi_part_status(h))
fz = ctypes.c_int(); mpk.check(L.mi_part_push_info(h, None, ctypes.byref(fz), None))
print(f"    kernels: interior {L.mi_part_kernel_name(h, 0).decode()} | boundary {L.mi_part_kernel_name(h, 1).decode()} | one-launch step {L.mi_part_kernel_name(h, 2).decode() or '-'}")
print(f"    push step ({'ONE launch (fused)' if fz.value else 'push, interior, wait+copy, boundary'}; flags preset, pushes looped back): GPU {e0.elapsed_time(e1) / REPS * 1e3:.1f} us/step, "
      f"host {t_enq:.1f} us/step to enqueue ({pwall:.1f} with the final synchronise)")
if fz.value and "sstream" in L.mi_part_kernel_name(h, 2).decode() and hasattr(L, "mi_debug_part_push_trace"):
    # where one launch of the fused sliced-stream step goes: s_memrealtime per workgroup, ghost-reading / pushing workgroups apart
    G = 1024
    acc = {}
    for it in range(12):
        buf, wg, fl = np.zeros(4 * G, np.int64), ctypes.c_int(), np.zeros(G, np.int32)
        mpk.check(L.mi_debug_part_push_trace(h, vp(x_ext.data_ptr()), vp(y.data_ptr()), G, buf.ctypes.data, ctypes.byref(wg), fl.ctypes.data))
        if it < 2:
            continue
        t = buf[:4 * wg.value].reshape(-1, 4).astype(np.float64) * 0.01
        ran = t[:, 3] > 0
        t0 = t[ran, 0].min()
        for cls, sel in (("plain", (fl[:wg.value] & 3) == 0), ("ghost reader", (fl[:wg.value] & 1) == 1), ("pusher", (fl[:wg.value] & 2) == 2)):
            sel = sel & ran
            if sel.any():
                acc.setdefault(cls, []).append((t[sel, 0] - t0, t[sel, 1] - t[sel, 0], t[sel, 2] - t[sel, 1], t[sel, 3] - t0, (fl[:wg.value][sel] >> 2)))
        acc.setdefault("all", []).append(t[ran, 3].max() - t0)
    print(f"    one traced launch of the fused step (stamps compiled in; 10 launches): first start -> last end median {np.median(acc.pop('all')):.2f} us")
    for cls, rows in acc.items():
        st, fi, lo_, en, rd = (np.concatenate([r[i] for r in rows]) for i in range(5))
        print(f"      {cls:12s} ({len(rows[0][0]):3d} workgroups, {rd.min()}-{rd.max()} rounds): start +{np.median(st):.2f}, start -> loop {np.median(fi):.2f} (max {fi.max():.2f}), "
              f"loop {np.median(lo_):.2f}, end +{np.median(en):.2f} (max {en.max():.2f}) us")
if fz.value and "fused_ext" in L.mi_part_kernel_name(h, 2).decode() and hasattr(L, "mi_debug_part_ext_trace"):
    G = 65536
    rows = []
    for it in range(8):
        buf, wg, md = np.zeros(3 * G, np.int64), ctypes.c_int(), np.zeros(G, np.int32)
        mpk.check(L.mi_debug_part_ext_trace(h, vp(x_ext.data_ptr()), vp(y.data_ptr()), G, buf.ctypes.data, ctypes.byref(wg), md.ctypes.data))
        if it >= 2:
            t = buf[:3 * wg.value].reshape(-1, 3).astype(np.float64) * 0.01
            rows.append((t - t[:, 0].min(), md[:wg.value].copy()))
    print(f"    one traced launch of the staged step (6 launches; us from the first workgroup's start): last end median {np.median([r[0][:, 2].max() for r in rows]):.2f}")
    for cls, sel in (("push", lambda m: m == -2), ("wait + copy", lambda m: m == -1), ("plain units", lambda m: m == 0), ("waiting units", lambda m: (m >= 0) & ((m & 1) == 1))):
        st = np.concatenate([r[0][sel(r[1]), 0] for r in rows]); wt = np.concatenate([r[0][sel(r[1]), 1] for r in rows]); en = np.concatenate([r[0][sel(r[1]), 2] for r in rows])
        if len(st):
            print(f"      {cls:14s} ({int(sel(rows[0][1]).sum()):4d} workgroups): start median {np.median(st):.2f} (max {st.max():.2f}), wait over {np.median(wt):.2f} (max {wt.max():.2f}), "
                  f"end {np.median(en):.2f} (max {en.max():.2f}); wait over -> end median {np.median(en - wt):.2f} (max {(en - wt).max():.2f})")
if fz.value and "fused_ext" in L.mi_part_kernel_name(h, 2).decode() and hasattr(L, "mi_debug_part_ext_mode") and os.environ.get("SIM_RANK_EXT_PARTS", "1") == "1":
    # what each part of the staged one-launch step costs: the same launch with parts left out (results wrong; timing only)
    for mode, name in ((2, "no push"), (4, "no window copy"), (6, "no push, no copy"), (1, "nobody waits"), (7, "no push, no copy, nobody waits")):
        mpk.check(L.mi_debug_part_ext_mode(h, mode))
        for _ in range(300): pstep()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(REPS): pstep()
        e1.record(); torch.cuda.synchronize()
        print(f"      staged step, {name}: {e0.elapsed_time(e1) / REPS * 1e3:.1f} us")
x_ext[nl.value:] = halo_keep   # the looped-back window content is not this rank's true halo: restore it for the check below
y.fill_(float("nan")); step(); torch.cuda.synchronize()
cl = np.where((c >= lo) & (c < hi), c - lo, nl.value + np.searchsorted(halo_ids, c)).astype(np.int32)
ok = np.array_equal(O.spmv(p, cl, v, xe).view(np.uint64), y.cpu().numpy().view(np.uint64))
print(f"N={N} rank={rank} rows={nl.value} halo={nh.value} interior={ni.value} boundary={nb.value}: "
      f"{'interior+boundary (sends are slices of x: no pack)' if contig.value else 'pack+interior+boundary'} GPU {gpu_step_us:.1f} us/step, host wall {wall:.1f} us/step, bitwise={ok}")
