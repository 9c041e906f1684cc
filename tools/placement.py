"""Launch time of the SAME matrix created several times in one process (different device placements)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = 5_000_000
p, c, v = synth.rows("s15", n)
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
keep = []
for rep in range(8):
    if rep in (3, 5): keep.append(torch.empty((37 + 61 * rep) * 1024 * 1024 // 8, dtype=torch.float64, device="cuda"))  # shift later placements
    A = mpk.csrmatrix(n, p, c, v)
    _ = A.handle
    for _ in range(20): mpk.SpMV_CSR(y, x, A)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): mpk.SpMV_CSR(y, x, A)
    e1.record(); e1.synchronize()
    print(f"create #{rep}: {e0.elapsed_time(e1) / 300 * 1e3:.1f} us/launch  {A.kernel_name()}  tune {A.tune_detail()[0]['ring_nt']:.1f}", flush=True)
    if rep % 2 == 0: keep.append(A)  # keep some handles alive so the next one lands elsewhere
