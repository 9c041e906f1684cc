"""Dev tool: what one SpMV_CSR call costs through the mpk/SpMV.h shim (the reference's calling convention: host vectors, the
matrix checked against the caller's live arrays by a full-content hash) at C2 and C4 size, with and without the hash."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import shim
from navierstokes_amd import synth
from oracle import oracle as O
for n in (1_000_000, 5_000_000):
    p, c, v = synth.rows("s15", n)
    x = synth.x_sin(0, n)
    t_hash, y = shim.time_spmv_csr(p, c, v, x, reps=3, trust=False)
    t_trust, y2 = shim.time_spmv_csr(p, c, v, x, reps=3, trust=True)
    ok = np.array_equal(y.view(np.uint64), O.spmv(p, c, v, x).view(np.uint64)) and np.array_equal(y, y2)
    mb = (p.nbytes + c.nbytes + v.nbytes) / 1e6
    print(f"SHIM n={n}: SpMV_CSR through the shim {t_hash * 1e3:.2f} ms per call with the full-content hash of {mb:.0f} MB, "
          f"{t_trust * 1e3:.2f} ms with mi355_assume_unchanged(true); hash alone {1e3 * (t_hash - t_trust):.2f} ms = {mb / 1e3 / max(t_hash - t_trust, 1e-9):.1f} GB/s; bitwise {ok}")
