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
os.environ["MI355_SSTREAM"] = "1"  # round 4: the sliced copy is planned whatever the size (eligible patterns only: the others refuse the kernel)
bad = 0
for seed in range(first, last):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([20000, 60000, 150000, 400000]))
    p, c = random_pattern(rng, n)
    v = rng.uniform(-1, 1, len(c)); x = rng.uniform(-1, 1, n)
    yr = O.spmv(p, c, v, x)
    xd = torch.from_numpy(x).cuda()
    names = []
    for kernel in ("auto", "stream", "ring", "tile", "mring", "sstream"):
        A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        try:
            mpk.SpMV_CSR(y, xd, A)
        except mpk.MiError:
            if kernel != "sstream":
                raise
            names.append("sstream-n/a")  # rows too ragged or band too wide for the sliced copy: refused, loudly
            continue
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
    # round 4: a band with near-uniform rows (what the sliced-stream kernel is for), random half-bandwidth / row length / ragged tail; and a
    # random 4x4-blocked pattern through the sliced blocked kernel (every variant)
    hb, per = int(rng.integers(1, 3500)), int(rng.integers(1, 40))
    nb = int(rng.choice([5000, 70000, 300000]))
    i = np.arange(nb)
    lens = np.minimum(per - (rng.random(nb) < 0.05) * rng.integers(0, per, nb), 2 * hb + 1)
    lens = np.maximum(lens, 0)
    cols = [np.sort(rng.choice(np.arange(max(0, r - hb), min(nb, r + hb + 1)), size=min(l, min(nb, r + hb + 1) - max(0, r - hb)), replace=False)) for r, l in zip(i[:2000], lens[:2000])]
    # (drawing 300 k rows one by one is slow in numpy: the first 2000 rows are random, the rest repeat their offsets shifted)
    pb = [0]
    cb = []
    for r in range(nb):
        base = cols[r % 2000] - (r % 2000) + r
        base = base[(base >= 0) & (base < nb)]
        cb.append(base)
        pb.append(pb[-1] + len(base))
    pb = np.array(pb, np.int32)
    cb = np.concatenate(cb).astype(np.int32)
    vb = rng.uniform(-1, 1, len(cb))
    xb = rng.uniform(-1, 1, nb)
    A = mpk.csrmatrix(nb, pb, cb, vb).set_kernel("sstream")
    y = torch.full((nb,), float("nan"), dtype=torch.float64, device="cuda")
    try:
        mpk.SpMV_CSR(y, torch.from_numpy(xb).cuda(), A)
        yb_ref = O.spmv(pb, cb, vb, xb)
        ok = np.array_equal(y.cpu().numpy().view(np.uint64), yb_ref.view(np.uint64))
        names.append(f"band(hb={hb},per={per},n={nb}):{A.kernel_name().split('<')[0][5:]}" + ("" if ok else "!!"))
        bad += not ok
        # the dot epilogue of the sliced stream, and the same band through a scattered row map (two 8-byte stores per lane)
        bb = rng.uniform(-1, 1, nb)
        y.fill_(float("nan"))
        beta = float(mpk.SpMV_CSR_dot(y, torch.from_numpy(xb).cuda(), A, torch.from_numpy(bb).cuda()))
        ok = np.array_equal(y.cpu().numpy().view(np.uint64), yb_ref.view(np.uint64)) and abs(beta - O.dot(bb, yb_ref)) <= 1e-13 * float(np.abs(bb * yb_ref).sum()) + 1e-300
        names.append(("sdotE" if A.dot_in_epilogue() else "sdot2") + ("" if ok else "!!"))
        bad += not ok
        del A
        rowmap = rng.permutation(nb + 11)[:nb].astype(np.int32)
        A = mpk.csrmatrix(nb, pb, cb, vb, rowmap=rowmap).set_kernel("sstream")
        ym = torch.full((nb + 11,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(ym, torch.from_numpy(xb).cuda(), A)
        got = ym.cpu().numpy()
        rest = np.ones(nb + 11, bool); rest[rowmap] = False
        ok = np.array_equal(got[rowmap].view(np.uint64), yb_ref.view(np.uint64)) and np.isnan(got[rest]).all()
        names.append("smap" + ("" if ok else "!!"))
        bad += not ok
        # round 5: the same band behind a plain row offset (an odd one plans the rows one down: the sliced kernel keeps its 16-byte stores),
        # in a random variant of the kernel, then with new values (the LDS-staged refill that writes the CSR and the sliced values in one pass)
        del A
        off = int(rng.integers(0, 9))
        os.environ["MI355_SSTREAM_FORM"] = str(int(rng.integers(0, 4)))
        A = mpk.csrmatrix(nb, pb, cb, vb, rowmap=(np.arange(nb) + off).astype(np.int32)).set_kernel("sstream")
        yo_ = torch.full((nb + off + 2,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(yo_, torch.from_numpy(xb).cuda(), A)
        got = yo_.cpu().numpy()
        ok = np.array_equal(got[off:off + nb].view(np.uint64), yb_ref.view(np.uint64)) and np.isnan(got[:off]).all() and np.isnan(got[off + nb:]).all()
        vb2 = vb * np.cos(np.arange(len(vb)))
        A.update_values(torch.from_numpy(vb2).cuda())
        yo_.fill_(float("nan"))
        mpk.SpMV_CSR(yo_, torch.from_numpy(xb).cuda(), A)
        got = yo_.cpu().numpy()
        ok = ok and np.array_equal(got[off:off + nb].view(np.uint64), O.spmv(pb, cb, vb2, xb).view(np.uint64)) and np.isnan(got[:off]).all() and np.isnan(got[off + nb:]).all()
        names.append(f"soff{off}f{os.environ['MI355_SSTREAM_FORM']}:{A.kernel_name().split('<')[0][5:]}" + ("" if ok else "!!"))
        bad += not ok
        del os.environ["MI355_SSTREAM_FORM"]
    except mpk.MiError:
        names.append(f"band(hb={hb},per={per},n={nb}):n/a")
    del A
    # round 5: rows that name several column neighbourhoods (k bands far apart, ragged at the ends, a few columns dropped): the cut-ring form of
    # the sliced stream where its planner takes the pattern (spmv_sstream_mw), in a random variant, also after a value refresh and behind a row offset
    nm = int(rng.choice([30_000, 120_000, 400_000]))
    kb = int(rng.integers(2, 5))
    gap = int(rng.integers(2600, 30_000))
    wid = int(rng.integers(1, 6))
    offs = np.concatenate([np.arange(wid) * int(rng.integers(1, 40)) + (jb - kb // 2) * gap for jb in range(kb)])
    cols = np.arange(nm)[:, None] + offs[None, :]
    keep = (cols >= 0) & (cols < nm) & (rng.random(cols.shape) > 0.03)
    pm = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int32)
    cm = cols[keep].astype(np.int32)
    vm = rng.uniform(-1, 1, len(cm))
    xm = rng.uniform(-1, 1, nm)
    offm = int(rng.integers(0, 5))
    os.environ["MI355_SSTREAM_FORM"] = str(int(rng.integers(0, 4)))
    try:
        A = mpk.csrmatrix(nm, pm, cm, vm, rowmap=(np.arange(nm) + offm).astype(np.int32) if offm else None).set_kernel("sstream")
        ym = torch.full((nm + offm + 2,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(ym, torch.from_numpy(xm).cuda(), A)
        got = ym.cpu().numpy()
        ok = np.array_equal(got[offm:offm + nm].view(np.uint64), O.spmv(pm, cm, vm, xm).view(np.uint64)) and np.isnan(got[:offm]).all() and np.isnan(got[offm + nm:]).all()
        vm2 = vm * np.sin(1 + np.arange(len(vm)))
        A.update_values(torch.from_numpy(vm2).cuda())
        ym.fill_(float("nan"))
        mpk.SpMV_CSR(ym, torch.from_numpy(xm).cuda(), A)
        got = ym.cpu().numpy()
        ok = ok and np.array_equal(got[offm:offm + nm].view(np.uint64), O.spmv(pm, cm, vm2, xm).view(np.uint64))
        names.append(f"bands(k={kb},gap={gap},n={nm},off={offm}):{A.kernel_name().split('<')[0][5:]}" + ("" if ok else "!!"))
        bad += not ok
        del A
    except mpk.MiError:
        names.append(f"bands(k={kb},gap={gap},n={nm}):n/a")
    del os.environ["MI355_SSTREAM_FORM"]
    os.environ["MI355_BCSR_SELL"] = "1"
    nbr = int(rng.choice([300, 5000, 40000]))
    bl = rng.integers(0, 20, nbr)
    bp = np.concatenate([[0], np.cumsum(bl)]).astype(np.int32)
    bc = np.concatenate([np.sort(rng.choice(nbr, size=l, replace=False)) for l in bl] + [np.zeros(0, np.int64)]).astype(np.int32)
    bv = rng.uniform(-1, 1, 16 * len(bc))
    xx = rng.uniform(-1, 1, 4 * nbr)
    yb = O.spmv_bcsr4(bp, bc, bv, xx)
    for form in "0123":
        os.environ["MI355_BCSR_SELL_FORM"] = form
        B = mpk.bcsr4x4_matrix(nbr, bp, bc, bv, nbcols=nbr)
        yy = torch.full((4 * nbr,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_BCSR(yy, torch.from_numpy(xx).cuda(), B)
        ok = np.array_equal(yy.cpu().numpy().view(np.uint64), yb.view(np.uint64))
        names.append(f"sell{form}" + ("" if ok else "!!"))
        bad += not ok
        del B
    del os.environ["MI355_BCSR_SELL_FORM"], os.environ["MI355_BCSR_SELL"]
    print(f"seed {seed} n {n} nnz {len(c)}: {' '.join(names)}", flush=True)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
