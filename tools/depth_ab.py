"""Dev tool: in-process A/B of the ring kernel's prefetch depth (MI355_RING_DEPTH is read per launch)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
p, c, v = synth.rows("s15", n)
A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def pipelined(reps=50):
    for _ in range(5): mpk.SpMV_CSR(y, x, A)
    e0.record()
    for _ in range(reps): mpk.SpMV_CSR(y, x, A)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
def cold(reps=8):
    t = 0.0
    for _ in range(reps):
        mpk.flush_cache(sync=False)
        e0.record(); mpk.SpMV_CSR(y, x, A); e1.record(); torch.cuda.synchronize()
        t += e0.elapsed_time(e1) * 1e3 / reps
    return t
print(n, A.kernel_name(), flush=True)
for rnd in range(3):
    for d in (os.environ.get("MI355_DEPTHS", "2,3,4").split(",")):
        os.environ["MI355_RING_DEPTH"] = d
        print(f"  round {rnd} depth {d}: back-to-back {pipelined():7.2f} us   cold caches (busy GPU) {cold():7.2f} us", flush=True)
