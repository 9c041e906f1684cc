"""A/B of the staging layout (MI355_RING_SKEW) inside one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "svar"
p, c, v = synth.rows(kind, n)
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
for rnd in range(2):
    for sk in ("0", "1"):
        os.environ["MI355_RING_SKEW"] = sk
        A = mpk.csrmatrix(n, p, c, v)
        A.set_kernel("ring")
        for _ in range(20): mpk.SpMV_CSR(y, x, A)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): mpk.SpMV_CSR(y, x, A)
        e1.record(); e1.synchronize()
        print(f"SKEWAB {kind} round {rnd} skew {sk}: {e0.elapsed_time(e1) / 300 * 1e3:.1f} us  {A.kernel_name()}", flush=True)
        del A
