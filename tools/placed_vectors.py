"""What mi_vec_alloc_placed buys (profiles/NOTES.md §4.12): a C4 product on torch-allocated x / y against vectors the library placed by timing candidate pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navierstokes_amd import mpk, synth
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "s15"
if kind == "mesh":
    p, c, v = synth.pressure_matrix(n); n = len(p) - 1
else:
    p, c, v = synth.rows(kind, n)
xh = synth.x_sin(0, n)
A = mpk.csrmatrix(n, p, c, v); _ = A.handle
def timed(x, y):
    for _ in range(15): mpk.SpMV_CSR(y, x, A)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): mpk.SpMV_CSR(y, x, A)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    return best
x0 = torch.from_numpy(xh).cuda(); y0 = torch.empty(n, dtype=torch.float64, device="cuda")
t_plain = timed(x0, y0)
(xp, yp), us = A.alloc_vectors(2, draws=int(os.environ.get("DRAWS", "8")))
xp.copy_(x0)
t_placed = timed(xp, yp)
yo = O.spmv(p, c, v, xh)
ok = np.array_equal(yp.cpu().numpy().view(np.uint64), yo.view(np.uint64)) and np.array_equal(y0.cpu().numpy().view(np.uint64), yo.view(np.uint64))
print(f"PLACED {kind} n={n}: torch-allocated x / y {t_plain:.1f} us; placed {t_placed:.1f} us; candidates {us}; bitwise {ok}; {A.kernel_name()[:50]}", flush=True)
del xp, yp
