"""Dev tool: the kernels on a mesh operator ALREADY in relabelled (scramble + library RCM) order, handed over as the user's
matrix — no twin, no row map, no gather: isolates what the relabelled STRUCTURE costs each kernel."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from navierstokes_amd import mpk, synth
from test_ring_plan import relabelled
from test_mring_plan import probe as mprobe
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100
p, c, v = synth.pressure_matrix(cells)
ps, cs, _ = synth.permute_nodes(p, c, v, block=1)[:3]
p2, c2 = relabelled(np.ascontiguousarray(ps, np.int32), np.ascontiguousarray(cs, np.int32))
n = len(p2) - 1
print("mring plan (nblk, runs, bad, served, forced cuts):", mprobe(p2, c2), flush=True)
os.environ["MI355_SPMV_AUTOTUNE"] = "0"; os.environ["MI355_REORDER"] = "0"
A = mpk.csrmatrix(n, p2, c2, np.ones(len(c2)))
x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in ("stream", "tile", "mring"):
    A.set_kernel(k)
    for _ in range(5): mpk.SpMV_CSR(y, x, A)
    e0.record()
    for _ in range(30): mpk.SpMV_CSR(y, x, A)
    e1.record(); torch.cuda.synchronize()
    print(f"  relabelled mesh {cells}^3: {k} {e0.elapsed_time(e1) * 1e3 / 30:7.1f} us ({A.kernel_name()})", flush=True)
