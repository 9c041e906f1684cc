"""Dev tool (round 4): ring vs sliced stream on S15 matrices of a rank's size at N = 8 / 4 / 2 (plain products, one GPU)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n in (625_000, 1_250_000, 2_500_000):
    p, c, v = synth.rows("s15", n)
    x = torch.from_numpy(synth.x_sin(0, n)).cuda()
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    out = dict(n=n)
    for k in ("ring", "sstream"):
        A = mpk.csrmatrix(n, p, c, v).set_kernel(k)
        out[k] = round(timed(lambda: mpk.SpMV_CSR(y, x, A)), 2)
        out[k + "_name"] = A.kernel_name()[:40]
        A.close()
    print(json.dumps(out), flush=True)
