"""Dev tool: in-process A/B of the ring planner's block row alignment (MI355_RING_ROW_ALIGN is read at create)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
from oracle import oracle as O
os.environ["MI355_SPMV_AUTOTUNE"] = "0"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cases = [("s15", 5_000_000, "ring"), ("s15", 1_000_000, "ring"), ("mesh", 170, "mring"), ("mesh", 100, "mring")]
for kind, n, kern in cases:
    p, c, v = synth.rows("s15", n) if kind == "s15" else synth.pressure_matrix(n)
    n = len(p) - 1
    x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
    H = {}
    for al in ("1", "64"):
        os.environ["MI355_RING_ROW_ALIGN"] = al
        H[al] = mpk.csrmatrix(n, p, c, v).set_kernel(kern); _ = H[al].handle
    yo = O.spmv(p, c, v, synth.x_sin(0, n)) if n <= 1_000_000 else None
    for rnd in range(3):
        for al in ("1", "64"):
            A = H[al]
            for _ in range(5): mpk.SpMV_CSR(y, x, A)
            e0.record()
            for _ in range(50): mpk.SpMV_CSR(y, x, A)
            e1.record(); torch.cuda.synchronize()
            ok = "" if yo is None else f" bitwise={np.array_equal(y.cpu().numpy().view(np.uint64), yo.view(np.uint64))}"
            print(f"  {kind} n={n} {kern} round {rnd} row_align {al:>2s}: {e0.elapsed_time(e1) * 1e3 / 50:7.2f} us{ok}", flush=True)
