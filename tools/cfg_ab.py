"""A/B of ring configurations inside ONE process (boxes and processes differ by +-10 %)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "s15"
cfgs = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "4,1,2").split(",")]
p, c, v = synth.rows(kind, n)
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
for rnd in range(2):
    for cfg in cfgs:
        os.environ["MI355_RING_CONFIG"] = str(cfg)
        A = mpk.csrmatrix(n, p, c, v)
        A.set_kernel("ring")
        for _ in range(20): mpk.SpMV_CSR(y, x, A)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): mpk.SpMV_CSR(y, x, A)
        e1.record(); e1.synchronize()
        print(f"CFGAB round {rnd} cfg {cfg}: {e0.elapsed_time(e1) / 300 * 1e3:.1f} us  {A.kernel_name()}", flush=True)
        del A
