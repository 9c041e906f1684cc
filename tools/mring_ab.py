"""Dev tool: ring vs multi-window ring vs stream on matrices both rings serve (a band narrower than one window) and on meshes."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
def timeit(A, n, reps=40):
    x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): mpk.SpMV_CSR(y, x, A)
    e0.record()
    for _ in range(reps): mpk.SpMV_CSR(y, x, A)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
os.environ["MI355_SPMV_AUTOTUNE"] = "0"
for name, gen in (("s15 1M w=300", lambda: synth.rows("s15", 1_000_000, w=300)), ("s15 5M w=300", lambda: synth.rows("s15", 5_000_000, w=300)),
                  ("mesh 100", lambda: synth.pressure_matrix(100)), ("mesh 170", lambda: synth.pressure_matrix(170))):
    p, c, v = gen()
    n = len(p) - 1
    A = mpk.csrmatrix(n, p, c, v)
    _ = A.handle
    out = []
    for k in ("ring", "mring", "stream", "tile"):
        A.set_kernel(k)
        out.append(f"{k} {timeit(A, n):7.1f} us ({A.kernel_name()[:44]})")
    print(name, " | ".join(out), flush=True)
    del A
