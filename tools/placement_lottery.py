"""Dev tool: does the PLACEMENT of a handle's arrays in device memory change its speed?  Several handles of one and the same matrix
(same plan, same kernel, same x and y), timed interleaved inside one process.  Usage: python tools/placement_lottery.py [c4|c2|mesh|fe] [copies]
MI355_SPMV_AUTOTUNE=0 is set: every handle runs the planner's default kernel, nothing is measured at create."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355_SPMV_AUTOTUNE", "0")
from navierstokes_amd import mpk, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if wl == "c4": p, c, v = synth.rows("s15", 5_000_000, w=2000)
elif wl == "c2": p, c, v = synth.rows("s15", 1_000_000, w=2000)
elif wl == "mesh": p, c, v = synth.pressure_matrix(170)
elif wl == "fe": p, c, v = synth.fe_matrix(68)
n = len(p) - 1
soak_gb = float(os.environ.get("LOTTERY_SOAK_GB", "0"))   # device memory taken (and kept) before anything else is allocated
soak = torch.empty(int(soak_gb * 2**30), dtype=torch.uint8, device="cuda") if soak_gb > 0 else None
free0, total0 = torch.cuda.mem_get_info()
print(f"soak {soak_gb} GB; device memory free {free0 / 2**30:.1f} of {total0 / 2**30:.1f} GiB", flush=True)
x = torch.from_numpy(synth.x_sin(0, n)).cuda(); y = torch.empty(n, dtype=torch.float64, device="cuda")
H = []
for i in range(copies):
    A = mpk.csrmatrix(n, p, c, v)
    mpk.SpMV_CSR(y, x, A)
    H.append(A)
torch.cuda.synchronize()
res = [[] for _ in H]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(6):
    for i, A in enumerate(H):
        for _ in range(3): mpk.SpMV_CSR(y, x, A)
        e0.record()
        for _ in range(20): mpk.SpMV_CSR(y, x, A)
        e1.record(); torch.cuda.synchronize()
        res[i].append(e0.elapsed_time(e1) * 1e3 / 20)
for i, r in enumerate(res):
    r = np.array(r)
    print(f"{wl} handle {i}: min {r.min():7.2f} median {np.median(r):7.2f} max {r.max():7.2f} us  ({H[i].kernel_name()})  draws at create: {H[i].placement_info()}")
if os.environ.get("LOTTERY_COLD"):  # the same handles cold: caches evicted in front of every launch (bench.py's protocol)
    for i, A in enumerate(H):
        ts = []
        for _ in range(10):
            mpk.flush_cache(sync=False)
            e0.record(); mpk.SpMV_CSR(y, x, A); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"{wl} handle {i} COLD: min {min(ts):7.2f} median {np.median(ts):7.2f} us   (warm median {np.median(res[i]):7.2f})", flush=True)
    # and with four other x / y pairs, warm and cold
    for j in range(4):
        x2 = x.clone(); y2 = torch.empty_like(y)
        for _ in range(3): mpk.SpMV_CSR(y2, x2, H[0])
        e0.record()
        for _ in range(20): mpk.SpMV_CSR(y2, x2, H[0])
        e1.record(); torch.cuda.synchronize()
        warm = e0.elapsed_time(e1) * 1e3 / 20
        ts = []
        for _ in range(10):
            mpk.flush_cache(sync=False)
            e0.record(); mpk.SpMV_CSR(y2, x2, H[0]); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"{wl} handle 0, x / y pair {j}: warm {warm:7.2f} cold median {np.median(ts):7.2f} us", flush=True)
        keep = (x2, y2) if j % 2 == 0 else keep
    sys.exit(0)
if os.environ.get("LOTTERY_RECREATE"):  # destroy handle 0 and create it again: the same memory back?
    H[0].close() if hasattr(H[0], "close") else None
    H[0] = None
    for j in range(3):
        A = mpk.csrmatrix(n, p, c, v)
        for _ in range(3): mpk.SpMV_CSR(y, x, A)
        e0.record()
        for _ in range(20): mpk.SpMV_CSR(y, x, A)
        e1.record(); torch.cuda.synchronize()
        print(f"{wl} re-created handle ({j}): {e0.elapsed_time(e1) * 1e3 / 20:7.2f} us", flush=True)
        H.append(A)
    sys.exit(0)
if os.environ.get("LOTTERY_MOVE"):  # which array's placement is it?  (devtools library)
    import ctypes
    L = mpk.lib()
    L.mi_debug_move_array.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    hows = ["hipMalloc", "contiguous", "uncached", "fine-grained"]
    names = ["coefficients", "ring slots", "row pointers", "plan records"]
    def t(A):
        for _ in range(3): mpk.SpMV_CSR(y, x, A)
        e0.record()
        for _ in range(20): mpk.SpMV_CSR(y, x, A)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 20
    for hi in range(min(2, len(H))):
        A = H[hi]
        print(f"{wl} handle {hi} as created: {t(A):7.2f} us", flush=True)
        for which, how, reps in ((1, 0, 3), (2, 0, 3), (3, 0, 3), (0, 0, 6), (1, 0, 3), (0, 0, 3)):  # hipMalloc only: hipDeviceMallocContiguous faulted the GPU here (twice) and is refused by the library
            for rep in range(reps):
                o, nw = ctypes.c_ulonglong(), ctypes.c_ulonglong()
                rc = L.mi_debug_move_array(A.handle, which, how, ctypes.byref(o), ctypes.byref(nw))
                if rc != 0:
                    print(f"   moving {names[which]} ({hows[how]}) failed: {mpk.last_error() if hasattr(mpk, 'last_error') else rc}", flush=True)
                    break
                print(f"   moved {names[which]:13s} ({hows[how]:12s}) {o.value:#x} -> {nw.value:#x}: {t(A):7.2f} us", flush=True)
    sys.exit(0)
# the same handle with other x / y allocations
A = H[0]
for j in range(4):
    x2 = x.clone(); y2 = torch.empty_like(y)
    for _ in range(3): mpk.SpMV_CSR(y2, x2, A)
    e0.record()
    for _ in range(20): mpk.SpMV_CSR(y2, x2, A)
    e1.record(); torch.cuda.synchronize()
    print(f"{wl} handle 0 with another x / y pair ({j}): {e0.elapsed_time(e1) * 1e3 / 20:7.2f} us  x at {x2.data_ptr():#x} y at {y2.data_ptr():#x}")
    keep = (x2, y2) if j == 0 else keep
