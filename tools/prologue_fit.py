"""Dev tool: ring kernel time against blocks per run (S15, 512 runs): T = a + b * blocks — what a launch costs outside its steady state."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
os.environ["MI355_SPMV_AUTOTUNE"] = "0"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pts = []
for nb in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    n = 512 * nb * 136
    p, c, v = synth.rows("s15", n)
    A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
    x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(10): mpk.SpMV_CSR(y, x, A)
    e0.record()
    for _ in range(100): mpk.SpMV_CSR(y, x, A)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / 100
    cfg, runs, bad, frac = A.ring_info()
    pts.append((nb, t))
    print(f"  {nb:3d} blocks per run ({n} rows, {A.kernel_name()[14:40]}): {t:7.2f} us  = {t / nb:6.2f} per block", flush=True)
nbs = np.array([q[0] for q in pts], float); ts = np.array([q[1] for q in pts])
b, a = np.polyfit(nbs, ts, 1)
print(f"fit: T = {a:.2f} us + {b:.3f} us per block")
# an empty kernel launch, back to back, for the launch-gap part
z = torch.zeros(64, device="cuda")
for _ in range(10): z.add_(1.0)
e0.record()
for _ in range(200): z.add_(1.0)
e1.record(); torch.cuda.synchronize()
print(f"tiny torch kernel back to back: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch")
