"""Per process: blockIdx -> XCD mapping of a 512-workgroup launch, next to the C4 launch time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
# diagnostics live in the devtools build of the library (make -C navierstokes_amd/csrc devtools; include/mi355_devtools.h)
os.environ.setdefault("MI355_SPMV_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "navierstokes_amd", "csrc", "libmi355spmv_dev.so"))
from navierstokes_amd import mpk, synth
L = mpk.lib()
m = np.zeros(512, np.int32)
mpk.check(L.mi_debug_xcc_map(512, m.ctypes.data))
match = float(np.mean(m == (np.arange(512) % 8)))
n = 5_000_000
p, c, v = synth.rows("s15", n)
A = mpk.csrmatrix(n, p, c, v)
x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(20): mpk.SpMV_CSR(y, x, A)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): mpk.SpMV_CSR(y, x, A)
e1.record(); e1.synchronize()
m2 = np.zeros(512, np.int32)
mpk.check(L.mi_debug_xcc_map(512, m2.ctypes.data))
print(f"XCCMAP first 24: {m[:24].tolist()}  share(b%8==xcc)={match:.2f}  again={float(np.mean(m2 == (np.arange(512) % 8))):.2f}  "
      f"launch {e0.elapsed_time(e1) / 300 * 1e3:.1f} us  {A.kernel_name()}")
