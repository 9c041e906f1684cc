"""Dev tool: in-process A/B of the ring kernel's LEAN instantiation (MI355_RING_LEAN is read at create) at depths 2/3/4."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
p, c, v = synth.rows("s15", n)
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
os.environ["MI355_SPMV_AUTOTUNE"] = "0"
H = {}
for lean in ("0", "1"):
    os.environ["MI355_RING_LEAN"] = lean
    H[lean] = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
    _ = H[lean].handle
def pipelined(A, reps=50):
    for _ in range(5): mpk.SpMV_CSR(y, x, A)
    e0.record()
    for _ in range(reps): mpk.SpMV_CSR(y, x, A)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
def cold(A, reps=8):
    t = 0.0
    for _ in range(reps):
        mpk.flush_cache(sync=False)
        e0.record(); mpk.SpMV_CSR(y, x, A); e1.record(); torch.cuda.synchronize()
        t += e0.elapsed_time(e1) * 1e3 / reps
    return t
yo = O.spmv(p, c, v, synth.x_sin(0, n)) if n <= 1_000_000 else None
for rnd in range(2):
    for d in ("2", "4"):
        os.environ["MI355_RING_DEPTH"] = d
        for lean in ("0", "1"):
            A = H[lean]
            t = pipelined(A); tc = cold(A)
            ok = "" if yo is None else f" bitwise={np.array_equal(y.cpu().numpy().view(np.uint64), yo.view(np.uint64))}"
            print(f"  n={n} round {rnd} depth {d} lean {lean} ({A.kernel_name()[-32:]}): back-to-back {t:7.2f} us   cold caches {tc:7.2f} us{ok}", flush=True)
