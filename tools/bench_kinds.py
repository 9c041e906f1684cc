"""Dev tool: product-library SpMV rate on the three synthetic families (auto kernel choice + forced kernels)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
from oracle import oracle as O
for kind, n, w in [("s15", 5_000_000, 2000), ("svar", 5_000_000, 2000), ("sfe", 1_400_000, 2000), ("sfe", 1_400_000, 8000),
                   ("s15", 5_000_000, 4500), ("s15", 2_000_000, 200_000)]:
    p, c, v = synth.rows(kind, n, w=w)
    nnz = len(c)
    B = 12 * nnz + 4 * (n + 1) + 16 * n
    x = synth.x_sin(0, n)
    yr = O.spmv(p, c, v, x)
    dx = torch.from_numpy(x).cuda()
    line = f"{kind} n={n} w={w} nnz={nnz}:"
    for kern in ("auto", "ring", "stream"):
        A = mpk.csrmatrix(n, p, c, v).set_kernel(kern)
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        for _ in range(5): mpk.SpMV_CSR(y, dx, A)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): mpk.SpMV_CSR(y, dx, A)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        ok = np.array_equal(yr.view(np.uint64), y.cpu().numpy().view(np.uint64))
        cfg, runs, bad, frac = A.ring_info()
        line += f"  [{kern}: {us:.1f} us {B / us / 1e3:.0f} GB/s {2 * nnz / us / 1e3:.0f} GF {'exact' if ok else 'WRONG'} cfg{cfg} ring{frac:.2f}]"
        A.close()
    print(line, flush=True)
