"""Dev tool: the CPU baselines of bench.py side by side at C4 size — the oracle's fma chain and the reference's own object code (oracle/_ref), scalar / opt / fma, with and without the cache flush (profiles/NOTES.md: the cpu_baseline numbers)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from navierstokes_amd import synth
from oracle import oracle as O
n = 5_000_000
p, c, v = synth.rows("s15", n); x = synth.x_sin(0, n)
for rep in range(3):
    t = time.perf_counter(); y = O.spmv(p, c, v, x); dt = time.perf_counter() - t
    print(f"oracle fma chain warm call {rep}: {dt*1e3:.1f} ms = {2*len(c)/dt/1e9:.2f} GFLOP/s")
for var in ("scalar", "opt", "fma"):
    for flush in (True, False):
        t, _ = O.ref_time_spmv(p, c, v, x, var, reps=3, flush=flush)
        print(f"reference {var:6s} flush={flush}: {t*1e3:.1f} ms = {2*len(c)/t/1e9:.2f} GFLOP/s")
t, _ = O.time_spmv(p, c, v, x, reps=3, flush=True); print(f"oracle timed cold: {t*1e3:.1f} ms = {2*len(c)/t/1e9:.2f} GFLOP/s")
