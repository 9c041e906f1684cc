"""Dev tool (run under rocprofv3 --pmc ...): N handles of the same matrix, each launched `reps` times in a row; the per-dispatch
counters and durations in the profiler's CSV then show what differs between a fast and a slow placement (tools/placement_pmc_read.py)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355_SPMV_AUTOTUNE", "0")
from navierstokes_amd import mpk, synth
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p, c, v = synth.rows("s15", 5_000_000, w=2000)
n = len(p) - 1
x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
H = [mpk.csrmatrix(n, p, c, v) for _ in range(copies)]
for A in H: A.handle
torch.cuda.synchronize()
for A in H:
    for _ in range(reps): mpk.SpMV_CSR(y, x, A)
    torch.cuda.synchronize()
