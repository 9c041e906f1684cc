"""How many launches does a temporal / non-temporal ring kernel need to reach its steady time after the
other mode ran?  (Sizes the autotune in mi_csr_create.)  Usage: python tools/nt_warm.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
p, c, v = synth.rows("s15", n)
A = mpk.csrmatrix(n, p, c, v)
A.set_kernel("ring")
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
L = mpk.lib()
print("kernel", A.kernel_name(), "tune", A.tune_detail())
for rnd in range(3):
    for nt in (0, 1):
        mpk.check(L.mi_csr_set_nontemporal(A.handle, nt, -1))
        ts = []
        for i in range(24):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); mpk.SpMV_CSR(y, x, A); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"round {rnd} nt={nt}: " + " ".join(f"{t:.0f}" for t in ts))
# back-to-back (no sync between launches), 50 launches
for nt in (0, 1, 0, 1):
    mpk.check(L.mi_csr_set_nontemporal(A.handle, nt, -1))
    for _ in range(5): mpk.SpMV_CSR(y, x, A)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): mpk.SpMV_CSR(y, x, A)
    e1.record(); e1.synchronize()
    print(f"back-to-back nt={nt}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us")
