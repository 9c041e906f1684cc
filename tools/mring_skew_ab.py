"""Dev tool: run lengths of the multi-window ring kernel, A/B on ONE placement (mi_debug_mring_replan rewrites the plan into the arrays the
handle already has — two handles of one matrix differ by more than the effect, profiles/NOTES.md §4.12).  The two workgroups of a CU do not share it
evenly (profiles/NOTES.md §4.10): with equal runs the older finishes at ~88 % of the launch; giving it the longer run lets both end together.
Usage: python tools/mring_skew_ab.py [cells] [skews ...]   (MRING_AB_ORDER=rcm: the relabelled order)"""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MI355_SPMV_LIBRARY", os.path.join(ROOT, "navierstokes_amd", "csrc", "libmi355spmv_dev.so"))
os.environ["MI355_SPMV_AUTOTUNE"] = "0"; os.environ["MI355_REORDER"] = "0"; os.environ["MI355_SPMV_KERNEL"] = "mring"; os.environ["MI355_MRING_NT"] = "1"
from navierstokes_amd import mpk, synth
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 170
skews = [int(a) for a in sys.argv[2:]] or [0, 6, 8, 10, 12, 14]
p, c, v = synth.pressure_matrix(cells)
which = os.environ.get("MRING_AB_ORDER", "natural")
if which != "natural":
    from test_ring_plan import relabelled
    ps, cs, _ = synth.permute_nodes(p, c, v, block=1)[:3]
    p, c = relabelled(np.ascontiguousarray(ps, np.int32), np.ascontiguousarray(cs, np.int32))
    v = np.ones(len(c))
n = len(p) - 1
L = mpk.lib()
L.mi_debug_mring_replan.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for handle in range(int(os.environ.get("MRING_AB_HANDLES", "2"))):
    A = mpk.csrmatrix(n, p, c, v)
    mpk.SpMV_CSR(y, x, A)
    ref = y.clone()
    res = {s: [] for s in skews}
    info = {}
    for rnd in range(3):
        for s in skews:
            tl, lr = ctypes.c_int(), ctypes.c_int()
            mpk.check(L.mi_debug_mring_replan(A.handle, s, ctypes.byref(tl), ctypes.byref(lr)))
            info[s] = (tl.value, lr.value)
            for _ in range(3): mpk.SpMV_CSR(y, x, A)
            e0.record()
            for _ in range(20): mpk.SpMV_CSR(y, x, A)
            e1.record(); torch.cuda.synchronize()
            assert torch.equal(y, ref), "a re-planned product differs"
            res[s].append(e0.elapsed_time(e1) * 1e3 / 20)
    print(f"{which} {cells}^3, handle {handle}: " + "  ".join(f"skew {s}: {min(res[s]):6.1f} us (grid {info[s][0]}, longest {info[s][1]})" for s in skews), flush=True)
