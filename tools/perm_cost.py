"""Dev tool: where a relabelled (reordered) handle spends its time — run under rocprofv3 --kernel-trace --stats.
Single products go gather(x) -> twin kernel writing y through its row map; the power chain goes gather(x) once, then per
power twin kernel WITHOUT the row map + scatter kernel.  Comparing the two kernel rows isolates the cost of the mapped store."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
what = sys.argv[1] if len(sys.argv) > 1 else "fe"
perm = len(sys.argv) <= 2 or sys.argv[2] != "natural"
if what == "fe":
    p, c, v = synth.fe_matrix(68); block = 4
else:
    p, c, v = synth.pressure_matrix(int(what)); block = 1
n = len(p) - 1
if perm:
    p, c, v, _ = synth.permute_nodes(p, c, v, block=block)
A = mpk.csrmatrix(n, p, c, v)
print(what, "perm" if perm else "natural", A.kernel_name(), A.reorder_info(), flush=True)
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
for _ in range(30):
    mpk.SpMV_CSR(y, x, A)
torch.cuda.synchronize()
for _ in range(10):
    mpk.SpMkV(outs, x, A)
torch.cuda.synchronize()
