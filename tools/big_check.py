"""One large matrix (default 20 M rows, 300 M nonzeros): more runs than resident workgroups, 32-bit offsets near
their upper range.  Full-vector bitwise check against the oracle and the launch time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navierstokes_amd import mpk, synth
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
t = time.time(); p, c, v = synth.rows("s15", n); print(f"generated {len(c)} nnz in {time.time() - t:.1f} s", flush=True)
t = time.time(); A = mpk.csrmatrix(n, p, c, v); _ = A.handle; print(f"mi_csr_create {time.time() - t:.2f} s  kernel {A.kernel_name()}  ring {A.ring_info()}  tune {A.tune_detail()}", flush=True)
xh = synth.x_sin(0, n); x = torch.from_numpy(xh).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(5): mpk.SpMV_CSR(y, x, A)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): mpk.SpMV_CSR(y, x, A)
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
B = 12.0 * len(c) + 4.0 * (n + 1) + 16.0 * n
t = time.time(); yr = O.spmv(p, c, v, xh); print(f"oracle {time.time() - t:.1f} s", flush=True)
print(f"BIG n={n}: {us:.1f} us  {2.0 * len(c) / us / 1e3:.0f} GFLOP/s  {B / us / 1e3:.0f} GB/s = {B / us / 1e3 / 80:.1f} % of 8 TB/s  bitwise {np.array_equal(yr.view(np.uint64), y.cpu().numpy().view(np.uint64))}")
