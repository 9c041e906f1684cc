"""The reference's Krylov-step pipeline on device-resident vectors (mpk/SpMVmulti.cpp:559-574,
mpk/2SpMV.cpp main): y = A x;  x3 = y - alpha (b.y) b  (orthogonalize = dot + AXPY);  z = A x3.
Time per pipeline pass, its split, and bitwise parity of z against the oracle's pipeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navierstokes_amd import mpk, synth
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
p, c, v = synth.rows("s15", n)
A = mpk.csrmatrix(n, p, c, v)
xh = synth.x_sin(0, n); bh = np.cos(0.002 * np.arange(n))
x, b = torch.from_numpy(xh).cuda(), torch.from_numpy(bh).cuda()
y, x3, z = (torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3))
def pipe():
    mpk.SpMV_CSR(y, x, A); mpk.orthogonalize(n, b, y, x3, 1e-8); mpk.SpMV_CSR(z, x3, A)
def timed(fn, reps=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t_pipe = timed(pipe)
t_spmv = timed(lambda: mpk.SpMV_CSR(y, x, A))
t_orth = timed(lambda: mpk.orthogonalize(n, b, y, x3, 1e-8))
pipe(); torch.cuda.synchronize()
yo = O.spmv(p, c, v, xh)
_, x3o = O.orthogonalize(bh, yo)          # the reference's left-to-right dot; the device's fixed tree differs in the last bits of beta
zo = O.spmv(p, c, v, x3o)
nnz = len(c)
print(f"PIPE n={n}: {t_pipe:.1f} us per pass; alone: SpMV {t_spmv:.1f} us, orthogonalize {t_orth:.1f} us "
      f"(so each SpMV costs {(t_pipe - t_orth) / 2:.1f} us inside the pipeline); {4.0 * nnz / t_pipe / 1e3:.0f} GFLOP/s on the SpMV flops; "
      f"y bitwise {np.array_equal(yo.view(np.uint64), y.cpu().numpy().view(np.uint64))}, "
      f"rel_error x3 {O.rel_error(x3o, x3.cpu().numpy()):.1e}, z {O.rel_error(zo, z.cpu().numpy()):.1e}")
