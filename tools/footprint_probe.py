"""Does the spread of the warm rate (profiles/NOTES.md §4.12) come with the product's temporal footprint reaching the Infinity Cache's 256 MB?
S15 matrices of growing size, each timed on several x / y pairs: temporal footprint = 2 B/nnz column stream + 4 B/row pointers + 8 + 8 B/row.
usage: python tools/footprint_probe.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
sizes = [int(a) for a in sys.argv[1:]] or [3_600_000, 4_000_000, 4_400_000, 4_800_000, 5_000_000, 5_400_000, 6_000_000]
def timed(A, x, y):
    for _ in range(20): mpk.SpMV_CSR(y, x, A)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(150): mpk.SpMV_CSR(y, x, A)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 150 * 1e3)
    cl = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        mpk.flush_cache(sync=False)
        e0.record(); mpk.SpMV_CSR(y, x, A); e1.record(); e1.synchronize()
        cl.append(e0.elapsed_time(e1) * 1e3)
    return best, sorted(cl)[2]
for n in sizes:
    p, c, v = synth.rows("s15", n)
    nnz = int(p[-1])
    xh = torch.from_numpy(synth.x_sin(0, n))
    pairs = [(xh.cuda(), torch.empty(n, dtype=torch.float64, device="cuda")) for _ in range(3)]
    A = mpk.csrmatrix(n, p, c, v); _ = A.handle
    B = 12 * nnz + 4 * (n + 1) + 16 * n
    foot = (2 * nnz + 4 * n + 16 * n) / 1e6
    res = [timed(A, x, y) for x, y in pairs]
    print(f"FOOT n={n} temporal footprint {foot:.0f} MB  {A.kernel_name()[:44]}  warm/cold us: " + " ".join(f"{w:.1f}/{c_:.1f}" for w, c_ in res)
          + "  frac of 8 TB/s warm: " + " ".join(f"{B / w / 8e6:.3f}" for w, _ in res) + "  cold: " + " ".join(f"{B / c_ / 8e6:.3f}" for _, c_ in res), flush=True)
    del A, pairs
    torch.cuda.empty_cache()
