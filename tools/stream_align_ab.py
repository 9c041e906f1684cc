"""Dev tool: stream kernel with / without row blocks of whole waves (MI355_STREAM_ROW_ALIGN is read when the block table is built)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
os.environ["MI355_SPMV_AUTOTUNE"] = "0"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, gen in (("mesh 100", lambda: synth.pressure_matrix(100)), ("s15 1M", lambda: synth.rows("s15", 1_000_000)), ("mesh 170", lambda: synth.pressure_matrix(170))):
    p, c, v = gen(); n = len(p) - 1
    x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
    H = {}
    for al in ("1", "64"):
        os.environ["MI355_STREAM_ROW_ALIGN"] = al
        H[al] = mpk.csrmatrix(n, p, c, v).set_kernel("stream"); mpk.SpMV_CSR(y, x, H[al])
    for rnd in range(3):
        for al in ("1", "64"):
            A = H[al]
            for _ in range(5): mpk.SpMV_CSR(y, x, A)
            e0.record()
            for _ in range(40): mpk.SpMV_CSR(y, x, A)
            e1.record(); torch.cuda.synchronize()
            print(f"  {name} stream round {rnd} row_align {al:>2s}: {e0.elapsed_time(e1) * 1e3 / 40:7.2f} us", flush=True)
