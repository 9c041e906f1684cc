"""Placement levels against the VIRTUAL addresses of the caller's vectors: K x / y pairs on one C4 handle; prints address, offset inside 1 GiB and the time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = 5_000_000
p, c, v = synth.rows("s15", n)
xh = torch.from_numpy(synth.x_sin(0, n))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PRE = int(os.environ.get("XY_PRE", "0"))  # pairs allocated BEFORE the handle exists (fresh device memory)
pre = []
for k in range(PRE):
    x = xh.cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda"); pre.append((x, y))
A = mpk.csrmatrix(n, p, c, v); _ = A.handle
def timed(x, y):
    for _ in range(15): mpk.SpMV_CSR(y, x, A)
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): mpk.SpMV_CSR(y, x, A)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    return best
pairs = []
SLAB = int(os.environ.get("XY_SLAB_MB", "0"))  # > 0: every pair is carved out of ONE allocation of that many MB, 64 MB apart
if SLAB:
    slab = torch.empty(SLAB << 20, dtype=torch.uint8, device="cuda")
    print(f"XYADDR slab at {slab.data_ptr():#x}", flush=True)
    step = 64 << 20
    for k in range(K):
        x = slab[(2 * k) * step:(2 * k) * step + 8 * n].view(torch.float64); x.copy_(xh)
        y = slab[(2 * k + 1) * step:(2 * k + 1) * step + 8 * n].view(torch.float64)
        pairs.append((x, y))
else:
    for k in range(K):
        x = xh.cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda"); pairs.append((x, y))
def copy_us(x, y, cold):
    ts = []
    for _ in range(15):
        if cold: mpk.flush_cache(sync=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y.copy_(x); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for k, (x, y) in enumerate(pre + pairs):
    t = timed(x, y)
    print(f"XYCOPY pair {k}: y.copy_(x) warm {copy_us(x, y, False):.1f} us  cold {copy_us(x, y, True):.1f} us", flush=True)
    print(f"XYADDR {'pre ' if k < PRE else 'post'} pair {k}: x {x.data_ptr():#x} y {y.data_ptr():#x}  x mod 1GiB {x.data_ptr() % (1 << 30) >> 21:4d} x2MB  y mod 1GiB {y.data_ptr() % (1 << 30) >> 21:4d} x2MB  {t:.1f} us", flush=True)
# the same y with another x and vice versa, for the fastest and the slowest pair
