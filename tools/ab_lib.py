"""A/B of two builds of libmi355spmv.so inside one gpurun call (sequential processes, several rounds).
usage: python tools/ab_lib.py <n> <kind> [kernel]   (kind: s15 | svar | sfe | mesh (n = cells per edge); run by tools/ab_lib.sh with MI355_LIB pointing at the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = int(sys.argv[1]); kind = sys.argv[2]; force = sys.argv[3] if len(sys.argv) > 3 else None
if kind == "mesh":  # n = cells per edge of the Kuhn box (P1 pressure operator)
    p, c, v = synth.pressure_matrix(n); n = len(p) - 1
else:
    p, c, v = synth.rows(kind, n)
H = int(os.environ.get("MI355_AB_HANDLES", "1"))  # several handles of the one matrix and several x / y pairs: each array has its own
XY = int(os.environ.get("MI355_AB_XY", "1"))       # placement in device memory (profiles/NOTES.md §4.12) — min / median over them say more than one draw
xh = torch.from_numpy(synth.x_sin(0, n))
pairs = [(xh.cuda(), torch.empty(n, dtype=torch.float64, device="cuda")) for _ in range(XY)]
warm, cold = [], []
handles = []
for h in range(H):
    A = mpk.csrmatrix(n, p, c, v)
    if force:
        A.set_kernel(force)
    handles.append(A)
    for x, y in pairs:
        for _ in range(30): mpk.SpMV_CSR(y, x, A)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): mpk.SpMV_CSR(y, x, A)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        cl = []
        for _ in range(7):  # cold launches: caches evicted in front of each
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            mpk.flush_cache(sync=False)
            e0.record(); mpk.SpMV_CSR(y, x, A); e1.record(); e1.synchronize()
            cl.append(e0.elapsed_time(e1) * 1e3)
        cl.sort()
        warm.append(best); cold.append(cl[len(cl) // 2])
med = lambda v: sorted(v)[len(v) // 2]
print(f"ABLIB {os.environ.get('MI355_LIB_TAG', '?')} n={n} {kind}: warm min {min(warm):.2f} median {med(warm):.2f} us  cold(median of 7) min {min(cold):.2f} median {med(cold):.2f} us  "
      f"[{' '.join(f'{w:.1f}/{c_:.1f}' for w, c_ in zip(warm, cold))}]  {handles[0].kernel_name()}")
