"""A/B of two builds of libmi355spmv.so inside one gpurun call (sequential processes, several rounds).
usage: python tools/ab_lib.py <n> <kind>   (run by tools/ab_lib.sh with MI355_LIB pointing at the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = int(sys.argv[1]); kind = sys.argv[2]
p, c, v = synth.rows(kind, n)
x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
A = mpk.csrmatrix(n, p, c, v)
for _ in range(30): mpk.SpMV_CSR(y, x, A)
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): mpk.SpMV_CSR(y, x, A)
    e1.record(); e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
print(f"ABLIB {os.environ.get('MI355_LIB_TAG', '?')} n={n} {kind}: {best:.2f} us  {A.kernel_name()}")
