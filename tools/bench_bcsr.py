"""Dev tool: BCSR 4x4 SpMV (SpMV_BCSR*, mpk/SpMV.cpp:90-219) rate on the SFE family vs the CSR path."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navierstokes_amd import mpk, synth
from oracle import oracle as O
for n, w in [(1_400_000, 2000), (1_400_000, 8000)]:
    p, c, v = synth.rows("sfe", n, w=w)
    nb = n // 4
    # SFE rows hold 14 groups of 4 consecutive columns, identical block columns for the 4 rows of a block row
    cc = c.reshape(n, 14, 4)
    vv = v.reshape(nb, 4, 14, 4)
    bcol = (cc[0::4, :, 0] // 4).astype(np.int32)                # [nb, 14]
    bval = np.ascontiguousarray(vv.transpose(0, 2, 1, 3))         # [nb, 14, 4(row), 4(col)] row-major blocks
    bptr = (np.arange(nb + 1) * 14).astype(np.int32)
    B = mpk.bcsr4x4_matrix(nb, bptr, bcol.reshape(-1), bval.reshape(-1), nbcols=nb)
    x = synth.x_sin(0, n)
    yr = O.spmv_bcsr4(bptr, bcol.reshape(-1), bval.reshape(-1), x)
    yc = O.spmv(p, c, v, x)
    dx = torch.from_numpy(x).cuda()
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(5): mpk.SpMV_BCSR(y, dx, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): mpk.SpMV_BCSR(y, dx, B)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nblk = nb * 14
    bytes_b = 132 * nblk + 4 * (nb + 1) + 16 * n
    ok = np.array_equal(yr.view(np.uint64), y.cpu().numpy().view(np.uint64))
    print(f"sfe n={n} w={w}: BCSR4 {us:.1f} us  {bytes_b / us / 1e3:.0f} GB/s (132 B/block model)  {2 * len(c) / us / 1e3:.0f} GFLOP/s  "
          f"bitwise vs SpMV_BCSR_FMA oracle={ok}  |bcsr-csr| rel={O.rel_error(yc, yr):.2e}", flush=True)
