"""Which of the caller's vectors decides the placement level (profiles/NOTES.md §4.12): every x with every y on one C4 handle (blocks cut at the nonzero count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = 5_000_000
p, c, v = synth.rows("s15", n)
xh = torch.from_numpy(synth.x_sin(0, n))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
xs = [xh.cuda() for _ in range(K)]
ys = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(K)]
def timed(A, x, y):
    for _ in range(15): mpk.SpMV_CSR(y, x, A)
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): mpk.SpMV_CSR(y, x, A)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    return best
print("XY addresses x:", [hex(t.data_ptr()) for t in xs], "y:", [hex(t.data_ptr()) for t in ys], flush=True)
for h in range(2):
    A = mpk.csrmatrix(n, p, c, v); _ = A.handle
    for i, x in enumerate(xs):
        print(f"XY handle {h} x{i} with y0..y{K - 1}: " + " ".join(f"{timed(A, x, y):.1f}" for y in ys), flush=True)
