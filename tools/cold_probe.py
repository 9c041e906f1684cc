"""Dev tool: what makes a cold single launch slow.  For one workload: back-to-back launches, cold launches (mi_flush_cache
before each), and cold launches with the address translations brought back first (mi_debug_touch_pages: one 4-byte read per
`stride` bytes of every array the kernel streams).  If the third equals the first-plus-data-fetch, the cold penalty is the TLB's."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# diagnostics live in the devtools build of the library (make -C navierstokes_amd/csrc devtools; include/mi355_devtools.h)
os.environ.setdefault("MI355_SPMV_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "navierstokes_amd", "csrc", "libmi355spmv_dev.so"))
from navierstokes_amd import mpk, synth
what = sys.argv[1] if len(sys.argv) > 1 else "c4"
if what == "fe":
    p, c, v = synth.fe_matrix(68)
elif what == "mesh":
    p, c, v = synth.pressure_matrix(170)
else:
    p, c, v = synth.rows("s15", {"c4": 5_000_000, "c2": 1_000_000}[what])
n = len(p) - 1
A = mpk.csrmatrix(n, p, c, v)
if len(sys.argv) > 2:
    A.set_kernel(sys.argv[2])
L = mpk.lib()
x = torch.from_numpy(synth.x_sin(0, n)).cuda()
y = torch.empty(n, dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def one():
    e0.record(); mpk.SpMV_CSR(y, x, A); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3
for _ in range(5): one()
warm = np.mean([one() for _ in range(20)])
def cold(stride):
    ts = []
    for _ in range(10):
        mpk.flush_cache()
        if stride:
            mpk.check(L.mi_debug_touch_pages(A.handle, stride, ctypes.c_void_p(x.data_ptr()), 8 * n, ctypes.c_void_p(y.data_ptr()), 8 * n))
        ts.append(one())
    return np.mean(ts), np.min(ts)
e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e2.record()
for _ in range(20): mpk.SpMV_CSR(y, x, A)
e3.record(); torch.cuda.synchronize()
pipelined = e2.elapsed_time(e3) * 1e3 / 20
def cold_busy():
    ts = []
    for _ in range(10):
        mpk.flush_cache(sync=False)
        ts.append(one())
    return np.mean(ts), np.min(ts)
def warm_busy():  # an unrelated kernel right in front (keeps the GPU busy) that does not disturb the caches much: a small read sweep
    ts = []
    z = torch.empty(1 << 20, dtype=torch.float64, device="cuda")
    for _ in range(10):
        for _ in range(30): z.add_(1.0)
        ts.append(one())
    return np.mean(ts), np.min(ts)
print(f"{what} {A.kernel_name()} depth_env={os.environ.get('MI355_RING_DEPTH')}: back-to-back (single-launch event pairs) {warm:.1f} us; back-to-back without syncs {pipelined:.1f} us", flush=True)
m, lo = cold_busy(); print(f"    cold caches, GPU kept busy (flush enqueued without sync): mean {m:7.1f} us  min {lo:7.1f} us", flush=True)
m, lo = warm_busy(); print(f"    warm caches, GPU kept busy (30 small kernels in front):  mean {m:7.1f} us  min {lo:7.1f} us", flush=True)
for stride in (0, 4096):
    m, lo = cold(stride)
    print(f"    cold, pages touched every {stride:>8d} B: mean {m:7.1f} us  min {lo:7.1f} us", flush=True)
