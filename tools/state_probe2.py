"""Which vector's placement moves the whole-wave ring plan at C4 (profiles/NOTES.md §4.12), and is it a matter of the virtual address (an offset
inside one allocation changes it) or of the physical pages (it does not)?   usage: python tools/state_probe2.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navierstokes_amd import mpk, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
p, c, v = synth.rows("s15", n)
xh = torch.from_numpy(synth.x_sin(0, n))
def timed(A, x, y):
    for _ in range(20): mpk.SpMV_CSR(y, x, A)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(150): mpk.SpMV_CSR(y, x, A)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 150 * 1e3)
    return best
os.environ["MI355_RING_ROW_ALIGN"] = "64"
A = mpk.csrmatrix(n, p, c, v); _ = A.handle
os.environ["MI355_RING_ROW_ALIGN"] = "1"
B = mpk.csrmatrix(n, p, c, v); _ = B.handle
xs = [xh.cuda() for _ in range(4)]
ys = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
print("STATE2 addresses x:", [hex(t.data_ptr()) for t in xs], " y:", [hex(t.data_ptr()) for t in ys], flush=True)
for name, H in (("whole-wave blocks", A), ("unaligned blocks", B)):
    for i, x in enumerate(xs):
        print(f"STATE2 {name}: x{i} with y0..y3: " + " ".join(f"{timed(H, x, y):.1f}" for y in ys), flush=True)
PAD = 1 << 16
xb = torch.empty(n + PAD, dtype=torch.float64, device="cuda"); yb = torch.empty(n + PAD, dtype=torch.float64, device="cuda")
offs = [0, 16, 32, 64, 128, 256, 512, 2048, 8192, 32768]
for name, H in (("whole-wave blocks", A), ("unaligned blocks", B)):
    r = []
    for o in offs:
        xo = xb[o:o + n]; xo.copy_(xs[0])
        r.append(timed(H, xo, yb[:n]))
    print(f"STATE2 {name}: x at +{[8 * o for o in offs]} B of one allocation, y fixed: " + " ".join(f"{t:.1f}" for t in r), flush=True)
    xo = xb[:n]; xo.copy_(xs[0])
    r = [timed(H, xo, yb[o:o + n]) for o in offs]
    print(f"STATE2 {name}: y at +{[8 * o for o in offs]} B of one allocation, x fixed: " + " ".join(f"{t:.1f}" for t in r), flush=True)
