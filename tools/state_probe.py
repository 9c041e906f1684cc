"""What decides the discrete warm/cold levels of the ring kernel at C4 (profiles/NOTES.md §4.12)?  Handles of one matrix with the block shape forced
(MI355_RING_ROW_ALIGN=1 | 64 at create) and left to the create-time measurement, each timed on several x / y pairs.
usage: python tools/state_probe.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
p, c, v = synth.rows("s15", n)
xh = torch.from_numpy(synth.x_sin(0, n))
pairs = [(xh.cuda(), torch.empty(n, dtype=torch.float64, device="cuda")) for _ in range(3)]
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
keep = []
for forced in ("1", "64", None, "1", "64", None):
    if forced: os.environ["MI355_RING_ROW_ALIGN"] = forced
    else: os.environ.pop("MI355_RING_ROW_ALIGN", None)
    A = mpk.csrmatrix(n, p, c, v); _ = A.handle; keep.append(A)
    rs = A.ring_shape_info(); pl = A.placement_info()
    res = [timed(A, x, y) for x, y in pairs]
    print(f"STATE align={forced or 'auto':4s} blocks={rs['blocks']} create-time us aligned/unaligned {rs['us_aligned']:.1f}/{rs['us_unaligned']:.1f} draws {pl['values']} {pl['column_stream']}  "
          f"warm/cold per x-y pair: {' '.join(f'{w:.1f}/{c_:.1f}' for w, c_ in res)}", flush=True)
