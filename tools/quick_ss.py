"""Dev tool (round 5): one S15 matrix of N rows through the sliced-stream kernel from Python, timed like tools/sim_rank.py times a rank's
pieces — to tell what a partition piece costs beyond the same rows as a plain handle.   python tools/quick_ss.py [rows] [kernel]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 625_000
kernel = sys.argv[2] if len(sys.argv) > 2 else "sstream"
lo = int(os.environ.get("QUICK_ROW0", "0"))
if lo:
    ng = 5_000_000
    p, c, v = synth.rows("s15", ng, lo, lo + n)   # a rank's rows of the 5 M-row matrix, columns clipped to the rank (what the interior piece looks like)
    keep = (c >= lo) & (c < lo + n)
    rows = np.repeat(np.arange(n), np.diff(p))
    cnt = np.bincount(rows[keep], minlength=n)
    p = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32); c = (c[keep] - lo).astype(np.int32); v = v[keep]
else:
    p, c, v = synth.rows("s15", n)
off = int(os.environ.get("QUICK_OFFSET", "-1"))   # >= 0: a row-mapped handle, row r -> y[r + off] (what a rank's interior piece is)
A = mpk.csrmatrix(n, p, c, v, rowmap=(np.arange(n) + off).astype(np.int32)) if off >= 0 else mpk.csrmatrix(n, p, c, v)
_ = A.handle
print("auto:", A.kernel_name(), A.sstream_info())
A.set_kernel(kernel)
print("forced:", A.kernel_name())
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n + max(off, 0), dtype=torch.float64, device="cuda")
R = 3000
for rep in range(3):
    for _ in range(300): mpk.SpMV_CSR(y, x, A)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): mpk.SpMV_CSR(y, x, A)
    e1.record(); torch.cuda.synchronize()
    print(f"  {n} rows, {A.kernel_name()}: {e0.elapsed_time(e1) / R * 1e3:.2f} us per launch")
