"""Dev tool: the GPU fuzz of tests/test_gpu_parity.py::test_random_patterns_every_kernel over many more seeds and larger matrices
(large enough for mi_csr_create to build every plan and time every candidate), each kernel bitwise against the oracle."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from navierstokes_amd import mpk
from oracle import oracle as O
from test_planner_fuzz import random_pattern
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([20000, 60000, 150000, 400000]))
    p, c = random_pattern(rng, n)
    v = rng.uniform(-1, 1, len(c)); x = rng.uniform(-1, 1, n)
    yr = O.spmv(p, c, v, x)
    xd = torch.from_numpy(x).cuda()
    names = []
    for kernel in ("auto", "stream", "ring", "tile", "mring"):
        A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(y, xd, A)
        ok = np.array_equal(y.cpu().numpy().view(np.uint64), yr.view(np.uint64))
        names.append(A.kernel_name().split("<")[0][9:] + ("" if ok else "!!"))
        bad += not ok
        del A
    if os.environ.get("FUZZ_ROUND3", "1") != "0":  # round 3's paths on the same pattern: one-launch powers step, dot epilogue
        os.environ["MI355_SPMK_FUSED"] = "1"
        vs = v / max(1.0, float(np.diff(p).max()))
        A = mpk.csrmatrix(n, p, c, vs).set_kernel("ring")
        Y = O.spmk_chain(3, p, c, vs, x)
        for rep in range(2):
            outs = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3)]
            mpk.SpMkV(outs, xd, A)
            ok = all(np.array_equal(outs[q].cpu().numpy().view(np.uint64), Y[q].view(np.uint64)) for q in range(3))
            bad += not ok
        names.append(("spmk1" if A.spmk_info(3)["one_launch"] else "spmk3") + ("" if ok else "!!"))
        bv = rng.uniform(-1, 1, n)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        beta = float(mpk.SpMV_CSR_dot(y, xd, A, torch.from_numpy(bv).cuda()))
        ok = np.array_equal(y.cpu().numpy().view(np.uint64), Y[0].view(np.uint64)) and abs(beta - O.dot(bv, Y[0])) <= 1e-13 * float(np.abs(bv * Y[0]).sum()) + 1e-300
        names.append(("dotE" if A.dot_in_epilogue() else "dot2") + ("" if ok else "!!"))
        bad += not ok
        del A
    print(f"seed {seed} n {n} nnz {len(c)}: {' '.join(names)}", flush=True)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
