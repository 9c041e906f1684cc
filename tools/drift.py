"""Does the launch time drift while the GPU stays busy?  (clock / power-state ramp on a fresh box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = 5_000_000
p, c, v = synth.rows("s15", n)
A = mpk.csrmatrix(n, p, c, v)
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
print("kernel", A.kernel_name(), A.tune_detail())
t_start = time.time()
while time.time() - t_start < float(sys.argv[1] if len(sys.argv) > 1 else 25):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): mpk.SpMV_CSR(y, x, A)
    e1.record(); e1.synchronize()
    print(f"t={time.time() - t_start:5.1f}s  {e0.elapsed_time(e1) * 2:.1f} us/launch", flush=True)
    if len(sys.argv) > 2: time.sleep(float(sys.argv[2]))
