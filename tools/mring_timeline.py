"""Dev tool: timeline of one launch of the multi-window ring kernel (mi_debug_mring_trace, devtools library): when every
workgroup started and finished, on which XCD, with how many blocks — how the planner's dealing of the runs (mring_plan.hpp)
plays out on the hardware.  Natural and relabelled order of the same mesh operator, the latter handed over as the caller's
matrix (no twin).  Usage: MI355_SPMV_LIBRARY=navierstokes_amd/csrc/libmi355spmv_dev.so python tools/mring_timeline.py [cells]"""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MI355_SPMV_LIBRARY", os.path.join(ROOT, "navierstokes_amd", "csrc", "libmi355spmv_dev.so"))
os.environ["MI355_SPMV_AUTOTUNE"] = "0"; os.environ["MI355_REORDER"] = "0"; os.environ["MI355_SPMV_KERNEL"] = "mring"; os.environ["MI355_MRING_NT"] = "1"
from navierstokes_amd import mpk, synth
from test_ring_plan import relabelled
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 170
L = mpk.lib()
L.mi_debug_mring_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


def timeline(tag, p, c):
    n = len(p) - 1
    A = mpk.csrmatrix(n, p, c, np.ones(len(c)))
    x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): mpk.SpMV_CSR(y, x, A)
    e0.record()
    for _ in range(30): mpk.SpMV_CSR(y, x, A)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    out = np.zeros(4 * 4096, np.int64); wgs = ctypes.c_int()
    for _ in range(3):  # the last of three launches (warm)
        mpk.check(L.mi_debug_mring_trace(A.handle, x.data_ptr(), y.data_ptr(), 4096, out.ctypes.data, ctypes.byref(wgs)))
    t = out[: 4 * wgs.value].reshape(-1, 4).copy()
    hw = t[:, 2] >> 8
    t[:, 2] &= 15
    np.save(os.path.join(ROOT, "gpurun_out", "mring_trace_" + tag.replace(" ", "_").replace(",", "").replace("^", "") + ".npy"), np.concatenate([t, hw[:, None]], axis=1))
    live = t[:, 3] != 0
    t0 = t[live, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us
    print(f"{tag}: {us:.1f} us per product ({A.kernel_name()}); grid {wgs.value}, runs {int(live.sum())}, traced launch {en[live].max():.1f} us")
    bid = np.arange(wgs.value)
    for xcd in range(8):
        m = live & (t[:, 2] == xcd)
        if not m.any(): continue
        late = m & (st > 5.0)
        blocks = np.abs(t[m, 3])
        longr = m & (np.abs(t[:, 3]) >= 0.8 * blocks.max())
        print(f"  XCD {xcd}: {int(m.sum()):3d} runs ({int(longr.sum())} long), {int(blocks.sum()):5d} blocks, last end {en[m].max():6.1f} us, long runs end {en[longr].min():6.1f}..{en[longr].max():6.1f},"
              f" us/block of long runs {np.mean((en[longr] - st[longr]) / np.abs(t[longr, 3])):.3f}, late starters {int(late.sum())} (start {st[late].min() if late.any() else 0:.1f}..{st[late].max() if late.any() else 0:.1f})"
              f" blockIdx%8 of its workgroups: {sorted(set((bid[m] % 8).tolist()))}")
    # how many workgroups are running at time t
    grid_t = np.linspace(0, en[live].max(), 13)
    print("  running workgroups at t =", " ".join(f"{tt:.0f}us:{int(((st <= tt) & (en > tt) & live).sum())}" for tt in grid_t))


p, c, v = synth.pressure_matrix(cells)
timeline(f"natural {cells}^3", np.ascontiguousarray(p, np.int32), np.ascontiguousarray(c, np.int32))
if os.environ.get("MRING_TIMELINE_ONLY") == "natural":
    sys.exit(0)
ps, cs, _ = synth.permute_nodes(p, c, v, block=1)[:3]
p2, c2 = relabelled(np.ascontiguousarray(ps, np.int32), np.ascontiguousarray(cs, np.int32))
timeline(f"relabelled {cells}^3", p2, c2)
if os.environ.get("MI355_MRING_DEAL") is None:
    os.environ["MI355_MRING_DEAL"] = "0"
    timeline(f"relabelled {cells}^3, runs by weight only (round-2 rule)", p2, c2)
