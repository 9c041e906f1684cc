"""Stress of the one-launch powers step's hand-off (spmk_ring.hpp): many launches on one handle, a DIFFERENT x every launch,
outputs poisoned with NaN before every launch, every word of every power compared with k chained launches of the same handle
(bit-equal to the oracle by the test suite) — under uneven load (a second stream keeps a copy kernel running on and off).
A stale line (an L1 or L2 copy of y_p from before its publication) shows as a mismatch.  Prints mismatching launches."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
p, c, v = synth.rows("s15", n)
A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
g = torch.Generator(device="cuda").manual_seed(7)
xs = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(7)]
os.environ["MI355_SPMK_FUSED"] = "0"
refs = []
for x in xs:
    outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(k)]
    mpk.SpMkV(outs, x, A)
    refs.append(outs)
torch.cuda.synchronize()
os.environ["MI355_SPMK_FUSED"] = "1"
outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(k)]
side = torch.cuda.Stream()
junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
bad = 0
for r in range(reps):
    x = xs[(r * 3) % 7]
    for t in outs:
        t.fill_(float("nan"))
    if r % 3 == 0:  # uneven load: a memset on another stream competes for CUs and memory
        with torch.cuda.stream(side):
            junk.fill_(r & 255)
    mpk.SpMkV(outs, x, A)
    ok = all(torch.equal(outs[q].view(torch.int64), refs[(r * 3) % 7][q].view(torch.int64)) for q in range(k))
    if not ok:
        bad += 1
        nb = [int((outs[q].view(torch.int64) != refs[(r * 3) % 7][q].view(torch.int64)).sum()) for q in range(k)]
        print(f"launch {r}: mismatching words per power {nb}", flush=True)
torch.cuda.synchronize()
print(f"SPMK_STRESS n={n} k={k} launches={reps} mismatching_launches={bad} info={A.spmk_info(k)} noacq={os.environ.get('MI355_SPMK_NOACQ')}")
