"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU
oracle and the committed golden vectors.  fp64; the bar north_star sets is
rel_error <= 1e-10 — the tests below hold the path to the stronger property it
is built for: every row is the same sequential fma chain as the reference's
SpMV_CSR_OPT/_FMA, so results are compared BITWISE, and against the x87
SpMV_CSR goldens with rel_error <= 1e-15."""
import numpy as np
import pytest
import torch

from conftest import assert_bit_equal
from navierstokes_amd import mpk, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

KERNELS = ["stream", "ring", "rowpar", "tile", "mring"]
TOL = 1e-10  # north_star: "match the reference CPU SpMV output within 1e-10 relative (fp64)"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    mpk.lib()  # fails loudly if the HIP extension is missing
    yield


@pytest.mark.parametrize("name", ["s15_n512", "svar_n400", "sfe_n268"])
@pytest.mark.parametrize("kernel", KERNELS)
def test_golden_spmv(golden, name, kernel):
    g = golden(name)
    A = mpk.csrmatrix(int(g["n"]), g["ptrow"], g["indcol"], g["coef"]).set_kernel(kernel)
    # host-pointer entry (the reference's calling convention)
    y = np.full(A.n, np.nan)
    mpk.SpMV_CSR(y, g["x"], A)
    assert_bit_equal(y, g["y_fma"], f"{kernel} vs SpMV_CSR_FMA")
    assert_bit_equal(y, g["y_opt"], f"{kernel} vs SpMV_CSR_OPT")
    assert O.rel_error(g["y_scalar"], y) <= 1e-15  # vs the x87 SpMV_CSR
    # device-resident entry
    yd = torch.full((A.n,), float("nan"), dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR_AVX2(yd, dev(g["x"]), A)
    assert_bit_equal(yd.cpu().numpy(), g["y_fma"])


@pytest.mark.parametrize("name", ["s15_n512", "svar_n400", "sfe_n268"])
def test_golden_powers(golden, name):
    g = golden(name)
    n = int(g["n"])
    A = mpk.csrmatrix(n, g["ptrow"], g["indcol"], g["coef"])
    # SpM2V_CSR(z, y, x, A, ptrowend1) — mpk/SpM2V.cpp:79-112
    y, z = np.empty(n), np.empty(n)
    mpk.SpM2V_CSR(z, y, g["x"], A, None)
    assert_bit_equal(y, g["m2_y_opt"], "SpM2V y")
    assert_bit_equal(z, g["m2_z_opt"], "SpM2V z")
    assert O.rel_error(g["m2_z_scalar"], z) <= 1e-15
    # SpM4V(v, w, z, y, x, ...) — mpk/SpMVmulti0.cpp:189-221 (x87 goldens: 1e-15), SpM3V is an fma build: bitwise
    v, w = np.empty(n), np.empty(n)
    mpk.SpM4V(v, w, z, y, g["x"], A)
    for got, k in ((y, 0), (z, 1), (w, 2), (v, 3)):
        assert O.rel_error(g["pow_fused4"][k], got) <= 1e-15
    for got, k in ((y, 0), (z, 1), (w, 2)):
        assert_bit_equal(got, g["pow_fused3"][k], f"SpM3V power {k + 1}")


@pytest.mark.parametrize("name", ["edge_coo_n37", "edge_coo_n40"])
def test_golden_coo_edge_cases(golden, name):
    g = golden(name)
    nrow = int(g["nrow"])
    A = mpk.COO2CSR(nrow, g["irow"], g["jcol"], g["val"])  # duplicates, empty rows, missing diagonal
    assert np.array_equal(A.ptrow, g["csr_ptrow"]) and np.array_equal(A.indcol, g["csr_indcol"])
    assert_bit_equal(A.coef, g["csr_coef"])
    for kernel in KERNELS:
        y = np.full(nrow, np.nan)
        mpk.SpMV_CSR(y, g["x"], A.set_kernel(kernel))
        assert_bit_equal(y, g["y_fma"], kernel)
    B = mpk.bcsr4x4_matrix(len(g["bcsr_ptrow"]) - 1, g["bcsr_ptrow"], g["bcsr_indcol"], g["bcsr_coef"])
    yb = np.full(4 * B.nrows, np.nan)
    mpk.SpMV_BCSR(yb, g["x"], B)
    assert_bit_equal(yb, g["yb_fma"], "SpMV_BCSR_FMA")
    assert O.rel_error(g["yb_scalar"], yb) <= 1e-15


@pytest.mark.parametrize("form", ["0", "1", "2", "3"])
@pytest.mark.parametrize("name", ["s15_n512", "svar_n400", "sfe_n268", "edge_coo_n37", "edge_coo_n40"])
def test_golden_spmv_through_the_sliced_stream(golden, name, form, monkeypatch):
    """The reference-made goldens (SpMV_CSR_FMA / _OPT / the x87 SpMV_CSR object code, tests/golden/make_golden.py) DIRECTLY through the
    kernel that carries the headline since round 4 (spmv_sstream): a few hundred rows = one workgroup, one round (or
    none whole), mostly padding places (ragged rows: the padding limit is lifted for the test), n < 512, duplicates / empty rows / a
    missing diagonal (edge_coo_*).  Each of the four variants forced; host-pointer and device-resident entry."""
    monkeypatch.setenv("MI355_SSTREAM", "1")
    monkeypatch.setenv("MI355_SSTREAM_MAX_PADDING", "1e9")
    monkeypatch.setenv("MI355_SSTREAM_FORM", form)
    g = golden(name)
    if name.startswith("edge_coo"):
        A = mpk.COO2CSR(int(g["nrow"]), g["irow"], g["jcol"], g["val"])
        assert np.array_equal(A.ptrow, g["csr_ptrow"]) and np.array_equal(A.indcol, g["csr_indcol"])
    else:
        A = mpk.csrmatrix(int(g["n"]), g["ptrow"], g["indcol"], g["coef"])
    A.set_kernel("sstream")
    assert A.kernel_name().startswith("spmv_sstream<") and A.sstream_info()["form"] == int(form), (A.kernel_name(), A.sstream_info())
    y = np.full(A.n, np.nan)
    mpk.SpMV_CSR(y, g["x"], A)
    assert_bit_equal(y, g["y_fma"], f"sstream form {form} vs SpMV_CSR_FMA")
    if "y_opt" in g:
        assert_bit_equal(y, g["y_opt"], f"sstream form {form} vs SpMV_CSR_OPT")
    assert O.rel_error(g["y_scalar"], y) <= 1e-15  # vs the x87 SpMV_CSR
    yd = torch.full((A.n,), float("nan"), dtype=torch.float64, device="cuda")
    for _ in range(2):
        mpk.SpMV_CSR_AVX2(yd, dev(g["x"]), A)
    assert_bit_equal(yd.cpu().numpy(), g["y_fma"])
    if "pow_fused3" in g:  # the powers step over a sliced-stream handle (k launches of it): SpM3V is an fma build -> bitwise
        n = A.n
        y1, z, w, v = (np.empty(n) for _ in range(4))
        mpk.SpM4V(v, w, z, y1, g["x"], A)
        for got, k in ((y1, 0), (z, 1), (w, 2)):
            assert_bit_equal(got, g["pow_fused3"][k], f"SpM3V power {k + 1} over the sliced stream")
        assert O.rel_error(g["pow_fused4"][3], v) <= 1e-15


@pytest.mark.parametrize("form", ["0", "1", "2", "3"])
def test_golden_blocked_products_through_the_sliced_copy(golden, form, monkeypatch):
    """The reference-made BCSR goldens (SpMV_BCSR_FMA / _OPT / _AVX2 object code) DIRECTLY through spmv_bcsr4_sell, each variant forced
    (VERDICT r4 weak 1a): edge_coo_* (9 / 10 block rows: one slice, mostly padding; last-wins duplicates; blocks in appearance order) through
    the BCSR API, sfe_n268 (67 block rows of 14 blocks) as the blocked copy of a CSR handle through SpMV_CSR."""
    monkeypatch.setenv("MI355_BCSR_SELL", "1")
    monkeypatch.setenv("MI355_BCSR_SELL_FORM", form)
    import ctypes
    L = mpk.lib()
    for name in ("edge_coo_n37", "edge_coo_n40"):
        g = golden(name)
        B = mpk.bcsr4x4_matrix(len(g["bcsr_ptrow"]) - 1, g["bcsr_ptrow"], g["bcsr_indcol"], g["bcsr_coef"])
        b, f = ctypes.c_int(), ctypes.c_int()
        mpk.check(L.mi_bcsr4_sell_info(B.handle, ctypes.byref(b), ctypes.byref(f), None, None, None))
        assert (b.value, f.value) == (1, int(form))
        yb = np.full(4 * B.nrows, np.nan)
        mpk.SpMV_BCSR(yb, g["x"], B)
        assert_bit_equal(yb, g["yb_fma"], f"{name}: sliced form {form} vs SpMV_BCSR_FMA")
        assert_bit_equal(yb, g["yb_opt"], f"{name}: sliced form {form} vs SpMV_BCSR_OPT")
        assert O.rel_error(g["yb_scalar"], yb) <= 1e-15
        xd = torch.zeros(4 * B.nbcols, dtype=torch.float64, device="cuda")
        xd[:len(g["x"])] = dev(g["x"])
        yd = torch.full((4 * B.nrows,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_BCSR(yd, xd, B)
        assert_bit_equal(yd.cpu().numpy(), g["yb_fma"], f"{name}: device-resident entry")
    g = golden("sfe_n268")
    A = mpk.csrmatrix(int(g["n"]), g["ptrow"], g["indcol"], g["coef"]).set_kernel("bcsr4")
    y = np.full(A.n, np.nan)
    mpk.SpMV_CSR(y, g["x"], A)
    assert "sell" in A.kernel_name(), A.kernel_name()
    assert_bit_equal(y, g["y_fma"], f"sfe_n268 through the sliced blocked copy, form {form}")
    assert O.rel_error(g["y_avx2"], y) <= 1e-15  # SpMV_CSR_AVX2 sums four partial chains (mpk/SpMV.cpp:59-85): not this chain bit for bit
    assert O.rel_error(g["y_scalar"], y) <= 1e-15


@pytest.mark.parametrize("kind,n,w", [("s15", 200_000, 2000), ("svar", 150_000, 2000), ("sfe", 100_000, 2000),
                                       ("s15", 70_001, 40_000)])
@pytest.mark.parametrize("kernel", KERNELS)
def test_seeded_vs_oracle(kind, n, w, kernel):
    p, c, v = synth.rows(kind, n, w=w)
    A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
    for x in (synth.x_sin(0, n), synth.x_ones(n)):
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(y, dev(x), A)
        yr = O.spmv(p, c, v, x)
        assert_bit_equal(y.cpu().numpy(), yr, f"{kind} {kernel}")
        assert O.rel_error(O.spmv(p, c, v, x, "x87"), y.cpu().numpy()) <= 1e-15


@pytest.mark.parametrize("depth", ["2", "3", "4"])
def test_ring_prefetch_depths_give_the_same_bits(depth, monkeypatch):
    """mi_csr_create picks the ring kernel's blocks of prefetch by run length (4 for long runs); every depth is the same
    arithmetic.  Forced here on matrices whose runs are short, where the pipeline is mostly sentinel blocks, and on one with
    empty and long rows (PLAIN blocks behind the loop)."""
    monkeypatch.setenv("MI355_RING_DEPTH", depth)
    for kind, n, w in (("s15", 120_000, 2000), ("svar", 90_000, 1500), ("s15", 3_000, 300)):
        p, c, v = synth.rows(kind, n, w=w)
        A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
        assert f", {depth}, 160," in A.kernel_name()
        x = synth.x_sin(0, n)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(y, dev(x), A)
        assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v, x), f"{kind} depth {depth}")
    rng = np.random.default_rng(5)
    n = 40_000
    lens = rng.integers(0, 30, n)
    lens[1234] = 3000
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    c = np.concatenate([np.sort(np.clip(i + rng.choice(np.arange(-1500, 1500), l, replace=False), 0, n - 1)) if l < 3000
                        else np.sort(rng.choice(n, l, replace=False)) for i, l in enumerate(lens)]).astype(np.int32)
    v = rng.uniform(-1, 1, p[-1])
    x = rng.uniform(-1, 1, n)
    A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
    y = np.full(n, np.nan)
    mpk.SpMV_CSR(y, x, A)
    assert_bit_equal(y, O.spmv(p, c, v, x), f"ragged depth {depth}")


def test_empty_rows_long_rows_and_degenerate_shapes():
    rng = np.random.default_rng(3)
    n = 5000
    lens = rng.integers(0, 12, n)
    lens[7] = 0
    lens[100] = 4000      # longer than a row block (2048): the long-row path
    lens[101] = 2048      # exactly one block
    lens[102] = 2049
    lens[4999] = 3000     # long last row
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    c = np.concatenate([np.sort(rng.choice(n, l, replace=False)) for l in lens]).astype(np.int32)
    v = rng.uniform(-1, 1, p[-1])
    x = rng.uniform(-1, 1, n)
    yr = O.spmv(p, c, v, x)
    for kernel in KERNELS:
        A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
        y = np.full(n, np.nan)
        mpk.SpMV_CSR(y, x, A)
        assert_bit_equal(y, yr, kernel)
    # all rows empty; 1x1; zero rows
    A = mpk.csrmatrix(4, np.zeros(5, np.int32), np.zeros(0, np.int32), np.zeros(0))
    y = np.full(4, np.nan)
    mpk.SpMV_CSR(y, np.ones(4), A)
    assert np.array_equal(y, np.zeros(4))
    A = mpk.csrmatrix(1, [0, 1], [0], [3.0])
    y = np.zeros(1)
    mpk.SpMV_CSR(y, np.array([2.0]), A)
    assert y[0] == 6.0
    A = mpk.csrmatrix(0, [0], [], [])
    mpk.SpMV_CSR(np.zeros(0), np.zeros(0), A)
    # rectangular with a rowmap (the partitioned pieces): rows scatter into a longer y
    A = mpk.csrmatrix(2, [0, 2, 3], [0, 5, 2], [1.0, 2.0, 3.0], ncols=6, rowmap=[4, 1])
    yd = torch.zeros(5, dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(yd, dev(np.arange(6.0)), A)
    assert yd.cpu().tolist() == [0.0, 6.0, 0.0, 0.0, 10.0]


def test_unreferenced_rows_are_computed_unlike_cpu_traversal():
    # the CPU first-touch kernel leaves y[j]=0 for rows never referenced as a column;
    # the GPU path returns the true A x there (documented difference), and the same z.
    p = np.array([0, 1, 2, 3], np.int32)
    c = np.array([0, 0, 0], np.int32)
    v = np.array([1.0, 2.0, 3.0])
    A = mpk.csrmatrix(3, p, c, v)
    y, z = np.empty(3), np.empty(3)
    mpk.SpM2V_CSR(z, y, np.ones(3), A)
    assert y.tolist() == [1.0, 2.0, 3.0] and z.tolist() == [1.0, 2.0, 3.0]
    yo, zo = O.spm2v_fused(p, c, v, np.ones(3))
    assert yo.tolist() == [1.0, 0.0, 0.0] and np.array_equal(zo, z)


def test_powers_k4_device_large():
    n = 300_000
    p, c, v = synth.rows("s15", n)
    A = mpk.csrmatrix(n, p, c, v)
    x = synth.x_sin(0, n)
    ys = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
    mpk.SpM4V(ys[3], ys[2], ys[1], ys[0], dev(x), A)
    Y = O.spmk_chain(4, p, c, v, x)
    for k in range(4):
        assert_bit_equal(ys[k].cpu().numpy(), Y[k], f"A^{k + 1} x")
        assert O.rel_error(Y[k], ys[k].cpu().numpy()) <= TOL


def test_blas1():
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 511, 512, 513, 100_003, 1_048_576):
        a, b = rng.standard_normal(n), rng.standard_normal(n)
        scale = np.abs(a * b).sum()
        assert abs(mpk.dot(a, b) - O.dot(a, b)) <= 1e-13 * scale
        assert abs(float(mpk.dot(dev(a), dev(b)).cpu()) - O.dot(a, b)) <= 1e-13 * scale
        # deterministic: same bits on every call
        assert mpk.dot(a, b) == mpk.dot(a, b)
        assert abs(mpk.norm2(a) - O.norm2(a)) <= 1e-14 * O.norm2(a)
        t = a * (1 + 1e-9 * rng.standard_normal(n))
        r = O.rel_error(a, t)
        assert abs(mpk.rel_error(a, t) - r) <= 1e-9 * r + 1e-25
        # AXPY is elementwise fma: bitwise
        y = b.copy()
        mpk.axpy(0.37, a, y)
        assert_bit_equal(y, O.axpy(0.37, a, b))
        # orthogonalize(nrow, b, x1, x3, alpha) — mpk/SpMVmulti.cpp:146-151
        x3 = np.full(n, np.nan)
        beta = mpk.orthogonalize(n, a, b, x3, 1e-8)
        beta_o, x3_o = O.orthogonalize(a, b, 1e-8)
        assert abs(beta - beta_o) <= 1e-13 * scale
        assert O.rel_error(x3_o, x3) <= TOL
        # odd (8-byte-aligned only) device views take the unaligned path
        if n > 8:
            da, db = dev(a), dev(b)
            got = float(mpk.dot(da[1:], db[1:]).cpu())
            assert abs(got - O.dot(a[1:], b[1:])) <= 1e-13 * scale
    assert mpk.dot(np.zeros(0), np.zeros(0)) == 0.0


def test_spmv_orthogonalize_spmv_pipeline():
    """The SpMV -> dot+AXPY -> SpMV pattern of mpk/SpMVmulti.cpp:559-574, all device-resident."""
    n = 120_000
    p, c, v = synth.rows("s15", n)
    A = mpk.csrmatrix(n, p, c, v)
    x = synth.x_ones(n)
    dx = dev(x)
    b = torch.empty(n, dtype=torch.float64, device="cuda")
    x1, x3, x2 = torch.empty_like(b), torch.empty_like(b), torch.empty_like(b)
    mpk.SpMV_CSR(b, dx, A)        # b = A x   (warm-up line :561)
    mpk.SpMV_CSR(x1, b, A)        # x1 = A b
    mpk.orthogonalize(n, b, x1, x3)
    mpk.SpMV_CSR(x2, x3, A)
    bo = O.spmv(p, c, v, x)
    x1o = O.spmv(p, c, v, bo)
    _, x3o = O.orthogonalize(bo, x1o)
    x2o = O.spmv(p, c, v, x3o)
    assert_bit_equal(x1.cpu().numpy(), x1o)
    assert O.rel_error(x2o, x2.cpu().numpy()) <= TOL


def test_full_size_properties_c2():
    """BASELINE config C2 (1 M rows, 15 M nnz) at full size: bitwise vs the oracle (the CPU
    finishes it in well under a second) plus linearity, a size-independent property."""
    n = 1_000_000
    p, c, v = synth.rows("s15", n)
    A = mpk.csrmatrix(n, p, c, v)
    x1, x2 = synth.x_sin(0, n), synth.x_ones(n)
    d1, d2 = dev(x1), dev(x2)
    y1, y2, y12 = (torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3))
    mpk.SpMV_CSR(y1, d1, A)
    mpk.SpMV_CSR(y2, d2, A)
    mpk.SpMV_CSR(y12, d1 + 2.0 * d2, A)
    assert_bit_equal(y1.cpu().numpy(), O.spmv(p, c, v, x1), "C2 sin")
    assert_bit_equal(y2.cpu().numpy(), O.spmv(p, c, v, x2), "C2 ones")
    lin = (y1 + 2.0 * y2).cpu().numpy()
    assert O.rel_error(lin, y12.cpu().numpy()) <= 1e-14
    # row sums: A * ones == sum of each row's coefficients (different summation order: tolerance)
    rs = np.add.reduceat(v, p[:-1])
    assert O.rel_error(rs, y2.cpu().numpy()) <= 1e-14


def test_reference_driver_linked_against_shim(tmp_path):
    """BASELINE config 1 stand-in (the bundled mesh/matrices are missing blobs): the
    reference's own mpk/2SpMV.cpp main, compiled against ITS mpk/SpMV.h and linked against
    libmpk_mi355.so instead of mpk/SpMV.cpp + mpk/utils.cpp (oracle/Makefile target
    _ref/2spmv_mi355), run on a 268-row FE-like matrix written in PETSc's MatrixMarket layout."""
    import os
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "2spmv_mi355")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/2spmv_mi355 not built (needs /root/reference at build time)")
    n = 268
    p, c, v = synth.rows("sfe", n, w=40)
    mtx = tmp_path / "matrix1_aij.mtx"
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n} {len(c)}\n")
        rows = np.repeat(np.arange(n), np.diff(p))
        for i, j, a in zip(rows, c, v):
            f.write(f"{i + 1} {j + 1} {a:.17g}\n")
    r = subprocess.run([exe, str(mtx)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"Matrix loaded: {n} rows, {len(c)} nonzeros" in r.stdout
    errs = [float(m) for m in re.findall(r"rel err = ([0-9.eE+-]+)", r.stdout)]
    assert len(errs) == 7, r.stdout  # 3 more CSR variants + 4 BCSR variants vs the first
    assert max(errs) <= 1e-15, r.stdout
    # and the numbers are right, not merely self-consistent: redo the driver's computation through the
    # same shim-level entry points and compare with the oracle on the float32-rounded coefficients
    nrow, ir, jc, va = O.read_mtx(str(mtx))
    A = mpk.COO2CSR(nrow, ir, jc, va)
    y = np.empty(n)
    mpk.SpMV_CSR(y, np.ones(n), A)
    assert_bit_equal(y, O.spmv(A.ptrow, A.indcol, A.coef, np.ones(n)))


def test_rccl_plumbing_selftest():
    """The native halo exchange (dlopen'ed RCCL, grouped send/recv on a side stream, event hand-offs)
    exercised on ONE GPU: a communicator of size 1 sends to itself through the same code path."""
    import ctypes
    L = mpk.lib()
    if L.mi_comm_available() != 0:
        pytest.skip("librccl not resolvable here: " + L.mi_last_error().decode())
    err = ctypes.c_double(-1.0)
    mpk.check(L.mi_comm_selftest(100_000, ctypes.byref(err)))
    assert err.value == 0.0


def test_block_shape_of_large_and_small_matrices(monkeypatch):
    """Round 3: matrices of >= 20 M nonzeros take row blocks cut at the nonzero count (their rate does not depend on where the
    caller's vectors lie), smaller ones blocks of whole waves of rows; MI355_RING_SHAPE_COMPARE=1 times both shapes at create.
    Either way every row is the reference's fma chain (mpk/SpMV.cpp:23-56), bit for bit.  (The ring kernel's shapes: the sliced stream, which
    AUTO prefers on this matrix since round 4, is kept out of the race here.)"""
    monkeypatch.setenv("MI355_SSTREAM", "0")
    n = 1_400_000  # 21 M nonzeros
    p, c, v = synth.rows("s15", n)
    x = synth.x_sin(0, n)
    yo = O.spmv(p, c, v, x)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    A = mpk.csrmatrix(n, p, c, v)
    _ = A.handle
    rs = A.ring_shape_info()
    assert "spmv_csr_ring<" in A.kernel_name() and rs["us_aligned"] == 0.0 and rs["us_unaligned"] == 0.0, (A.kernel_name(), rs)
    assert rs["blocks"] == -(-n // 136), rs  # 136 rows of 15 nonzeros fill a 2048-nonzero block; 128-row blocks would be more
    mpk.SpMV_CSR(y, dev(x), A)
    assert_bit_equal(y.cpu().numpy(), yo, "large matrix, blocks cut at the nonzero count")
    A.close()
    monkeypatch.setenv("MI355_RING_SHAPE_COMPARE", "1")
    B = mpk.csrmatrix(n, p, c, v)
    _ = B.handle
    rs = B.ring_shape_info()
    assert rs["us_aligned"] > 0.0 and rs["us_unaligned"] > 0.0 and rs["blocks"] in (-(-n // 136), -(-n // 128)), rs
    y.zero_()
    mpk.SpMV_CSR(y, dev(x), B)
    assert_bit_equal(y.cpu().numpy(), yo, "large matrix after the create-time shape comparison")
    B.close()
    monkeypatch.delenv("MI355_RING_SHAPE_COMPARE")
    m = 200_000  # 3 M nonzeros: whole waves of rows
    p, c, v = synth.rows("s15", m)
    S = mpk.csrmatrix(m, p, c, v)
    S.set_kernel("ring")
    assert S.ring_shape_info()["blocks"] == -(-m // 128), S.ring_shape_info()
    ys = torch.empty(m, dtype=torch.float64, device="cuda")
    xs = synth.x_sin(0, m)
    mpk.SpMV_CSR(ys, dev(xs), S)
    assert_bit_equal(ys.cpu().numpy(), O.spmv(p, c, v, xs), "small matrix, whole-wave blocks")


def test_vectors_placed_by_the_library():
    """mi_vec_alloc_placed (round 3, profiles/NOTES.md §4.12): x and y allocated by the library after timing candidate pairs.  Placement is a matter
    of speed only — the product on placed vectors is the reference's fma chain bit for bit (mpk/SpMV.cpp:23-56) — candidates are timed
    only for matrices beyond the caches, and the vectors come zero-filled and are released with their tensors."""
    n = 1_400_000  # 21 M nonzeros: candidates are timed
    p, c, v = synth.rows("s15", n)
    x = synth.x_sin(0, n)
    A = mpk.csrmatrix(n, p, c, v)
    (xp, yp, zp), us = A.alloc_vectors(3, draws=4)
    assert len(us) == 4 and all(t > 0 for t in us), us
    for t in (xp, yp, zp):
        assert t.shape == (n,) and t.dtype == torch.float64 and t.is_cuda and float(t.abs().max()) == 0.0
        assert t.data_ptr() % 256 == 0
    assert len({xp.data_ptr(), yp.data_ptr(), zp.data_ptr()}) == 3
    xp.copy_(dev(x))
    mpk.SpMV_CSR(yp, xp, A)
    yo = O.spmv(p, c, v, x)
    assert_bit_equal(yp.cpu().numpy(), yo, "product on placed vectors")
    mpk.SpMV_CSR(zp, yp, A)  # and chained: a placed vector as the next product's input
    assert_bit_equal(zp.cpu().numpy(), O.spmv(p, c, v, yo), "second product on placed vectors")
    del xp, yp, zp
    A.close()
    m = 50_000  # small: plain allocations, nothing timed
    p, c, v = synth.rows("s15", m)
    S = mpk.csrmatrix(m, p, c, v)
    (a, b), us = S.alloc_vectors(2, draws=8)
    assert us == [] and float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0
    xs = synth.x_sin(0, m)
    a.copy_(dev(xs))
    mpk.SpMV_CSR(b, a, S)
    assert_bit_equal(b.cpu().numpy(), O.spmv(p, c, v, xs), "small matrix, placed (plainly allocated) vectors")


def test_full_size_c4_and_c3():
    """BASELINE configs at full size.  C4: 5 M rows / 75 M nnz single SpMV; C3: k = 4 matrix powers on the
    1 M-row matrix.  Both bitwise against the oracle's fma chain, plus the size-independent
    identities (A(ax1 + x2) = a A x1 + A x2 to rounding; y = A*1 equals the row sums)."""
    n = 5_000_000
    p, c, v = synth.rows("s15", n)
    A = mpk.csrmatrix(n, p, c, v)
    x1 = synth.x_sin(0, n)
    d1 = dev(x1)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(y, d1, A)
    yo = O.spmv(p, c, v, x1)
    assert_bit_equal(y.cpu().numpy(), yo, "C4")
    assert O.rel_error(O.spmv(p, c, v, x1, "x87"), y.cpu().numpy()) <= 1e-15  # vs the reference's SpMV_CSR arithmetic
    ones = torch.ones(n, dtype=torch.float64, device="cuda")
    y1 = torch.empty_like(y)
    mpk.SpMV_CSR(y1, ones, A)
    assert O.rel_error(np.add.reduceat(v, p[:-1]), y1.cpu().numpy()) <= 1e-14
    y12 = torch.empty_like(y)
    mpk.SpMV_CSR(y12, 0.5 * d1 + ones, A)
    assert O.rel_error((0.5 * y + y1).cpu().numpy(), y12.cpu().numpy()) <= 1e-14
    # the create-time placement draws (value array and column stream re-copied, fastest copy kept): recorded, and the copies that
    # are kept are the ones later value updates write into
    pi = A.placement_info()
    if "sstream" in A.kernel_name():  # (round 4: AUTO runs the sliced stream here; it reads its own copy, so nothing is drawn for the CSR arrays)
        assert pi["values"] == [] and A.sstream_info()["built"], (pi, A.sstream_info())
    else:
        assert len(pi["values"]) >= 2 and len(pi["column_stream"]) >= 2 and all(t > 0 for t in pi["values"] + pi["column_stream"]), pi
    v2 = v * np.cos(np.arange(len(v)))
    A.update_values(v2)
    mpk.SpMV_CSR(y, d1, A)
    assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v2, x1), "C4 after the placement draws and a value update")
    A.close()
    # the same with the ring kernel (the placement draws of round 3: value array and column stream re-copied, the kept copies are the ones
    # later value updates write into)
    A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
    import os
    os.environ["MI355_SSTREAM"] = "0"
    try:
        mpk.SpMV_CSR(y, d1, A)
    finally:
        del os.environ["MI355_SSTREAM"]
    pi = A.placement_info()
    assert len(pi["values"]) >= 2 and len(pi["column_stream"]) >= 2 and all(t > 0 for t in pi["values"] + pi["column_stream"]), pi
    assert_bit_equal(y.cpu().numpy(), yo, "C4, ring kernel")
    A.update_values(v2)
    mpk.SpMV_CSR(y, d1, A)
    assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v2, x1), "C4 (ring) after the placement draws and a value update")
    A.close()
    del p, c, v, v2
    n = 1_000_000
    p, c, v = synth.rows("s15", n)
    A = mpk.csrmatrix(n, p, c, v)
    x = synth.x_ones(n)  # the reference harness's x (mpk/SpM2V.cpp:879)
    ys = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
    mpk.SpM4V(ys[3], ys[2], ys[1], ys[0], dev(x), A)
    Y = O.spmk_chain(4, p, c, v, x)
    for k in range(4):
        assert_bit_equal(ys[k].cpu().numpy(), Y[k], f"C3 power {k + 1}")


def test_native_step_single_rank():
    """mi_part_comm_init + mi_part_spmv_dev (the C++ step used for N > 1) on a 1-rank partition: the ctypes
    argument path, RCCL communicator creation and the pack/interior/boundary sequence, bitwise vs the oracle."""
    import ctypes
    from navierstokes_amd import dist as D
    L = mpk.lib()
    if L.mi_comm_available() != 0:
        pytest.skip("librccl not resolvable here")
    n = 300_000
    p, c, v = synth.rows("s15", n)
    dc = D.DistCSR(np.array([0, n], np.int64), p, c, v)  # no process group: one rank
    assert dc.nranks == 1 and dc.n_halo == 0 and dc.n_boundary == 0 and not dc.native
    buf = ctypes.create_string_buffer(128)
    mpk.check(L.mi_comm_unique_id(buf))
    mpk.check(dc._native_init(buf.raw))
    x = synth.x_sin(0, n)
    x_ext = dc.new_x_ext()
    x_ext[:n] = torch.from_numpy(x).cuda()
    y = dc.new_y()
    mpk.check(L.mi_part_spmv_dev(dc._h, ctypes.c_void_p(x_ext.data_ptr()), ctypes.c_void_p(y.data_ptr()), mpk._stream_ptr()))
    torch.cuda.synchronize()
    assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v, x), "native step, 1 rank")


@pytest.mark.parametrize("world,kind,n,w,port,exchange", [(3, "svar", 120_000, 2000, 29611, "torch"), (2, "sfe", 60_000, 1500, 29612, "torch"),
                                                           (2, "s15", 200_000, 2000, 29613, "push"), (4, "svar", 160_000, 2000, 29614, "push"),
                                                           (3, "sfe", 60_000, 1500, 29615, "push"),
                                                           # upwind coupling: the last rank sends but receives nothing (no ghost reader in
                                                           # its one-launch step: the pushers gate themselves, push_exchange.hpp)
                                                           (3, "s15_up", 150_000, 2000, 29616, "push"),
                                                           # (round 5) several rounds per workgroup of the sliced stream's fused step: ghost readers with
                                                           # fewer rounds than their share, their whole column range taken in up front
                                                           (2, "s15", 1_400_000, 2000, 29617, "push"),
                                                           # (round 5) a blocked rank's one-launch step with the ghosts STAGED once per step
                                                           # (spmv_bcsr4_ext.hpp), the ONE-launch form forced
                                                           (3, "sfe_ext", 60_000, 1500, 29618, "push"),
                                                           # ... and as the library runs it when it finds a neighbour's window on its own device (ranks
                                                           # sharing a card, as here): two launches, only the exchange's workgroups wait in-kernel
                                                           (3, "sfe_ext2", 60_000, 1500, 29619, "push"),
                                                           # (round 5) a 3-D mesh operator over ranks: wide halos (a plane each side: the four-launch step),
                                                           # interior pieces served by the cut-ring sliced stream where forced ("sstream" in the worker's kernel loop)
                                                           (2, "mesh", 70, 0, 29620, "push"),
                                                           # ... and its staged one-launch step (spmv_csr_fused_ext): forced one launch, and as the library runs
                                                           # it for ranks sharing a card (two launches)
                                                           (2, "mesh_ext", 70, 0, 29621, "push"), (2, "mesh_ext2", 70, 0, 29622, "push")])
def test_ranks_sharing_one_card(world, kind, n, w, port, exchange):
    """The N>1 pipeline on real HIP kernels: `world` ranks (processes) on cuda:0, A x, A^2 x, A^3 x and a global dot, every
    rank's slice bitwise.  exchange "torch": halos over gloo, host-staged (RCCL rejects duplicate devices).  exchange "push":
    the library's peer-push step (mi_part_spmv_push_dev) — receive windows mapped into the other PROCESSES with HIP IPC,
    kernels of one process writing into the window of another and raising its flags; gloo only carries the set-up."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    # (ranks SHARING a card is a test arrangement, not a deployment: every rank's persistent grid wants all 256 CUs and the
    # hardware arbitrates between the processes' queues as it likes, so a rank's launch can sit out long stretches of the others'.
    # The protocol itself is checked exhaustively on the host — tests/test_push_protocol.py: no interleaving of {push, flag, wait,
    # read} starves or overwrites, the two step forms mixed — so a give-up here is the card's time-slicing, not the protocol:
    # the wait gets 2^23 polls (~30 s) instead of the default 2^20 (~4 s), which round 2 saw give up once in a dozen runs of the
    # four-process case.  A wait that gives up is still loud and fails the run.)
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MI355_DIST_FORCE_SELFCHECK="1", MI355_TEST_EXCHANGE=exchange,
               MI355_PUSH_SPIN_LOG2=os.environ.get("MI355_PUSH_SPIN_LOG2", "23"))
    if kind == "sfe_ext":  # the one-launch form, forced (few workgroups here: the card has room for every rank's waiting ones)
        kind = "sfe"
        env.update(MI355_PUSH_EXT_SPLIT="0", MI355_TEST_EXPECT_FUSED="=spmv_bcsr4_fused_ext")
    elif kind == "mesh_ext":
        kind = "mesh"
        # (MI355_PUSH_FUSED_KERNEL=csr_ext: not the ring kernel's fused form, which takes such a rank when its plan serves it)
        env.update(MI355_PUSH_EXT_SPLIT="0", MI355_PUSH_FUSED_KERNEL="csr_ext", MI355_TEST_EXPECT_FUSED="=spmv_csr_fused_ext")
    elif kind == "mesh_ext2":
        kind = "mesh"
        env.pop("MI355_PUSH_EXT_SPLIT", None)
        env.update(MI355_PUSH_FUSED_KERNEL="csr_ext", MI355_TEST_EXPECT_FUSED="spmv_csr_fused_ext x2")
    elif kind == "sfe_ext2":
        kind = "sfe"
        env.pop("MI355_PUSH_EXT_SPLIT", None)
        env.update(MI355_TEST_EXPECT_FUSED="spmv_bcsr4_fused_ext x2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py"), kind, str(n), str(w)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert any(ln.startswith("DIST_GPU_RESULT ok=1") for ln in r.stdout.splitlines()), r.stdout[-2000:]


def test_fe_matrix_takes_the_blocked_kernel_with_identical_bits():
    """An FE matrix handed over as plain CSR (the reference's assemble_ns_matrix layout, src/benchmark_spmv.c:76-123)
    has exact 4x4 node-block structure: mi_csr_create keeps a BCSR copy and AUTO may run the BCSR kernel on it.
    Same terms in the same order per row, so the bits are those of the CSR fma chain."""
    p, c, v = synth.fe_matrix(14)
    n = len(p) - 1
    x = synth.x_sin(0, n)
    yr = O.spmv(p, c, v, x)
    A = mpk.csrmatrix(n, p, c, v)
    tune, _ = A.tune_detail()
    assert tune["bcsr4"] > 0.0, "blocked copy was not built / timed"
    for kernel in ("bcsr4", "stream", "auto"):
        A.set_kernel(kernel)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(y, dev(x), A)
        assert_bit_equal(y.cpu().numpy(), yr, f"FE matrix, {kernel} -> {A.kernel_name()}")
    A.set_kernel("bcsr4")
    assert A.kernel_name() == "spmv_bcsr4<2>"
    # an x that is only 8-byte aligned cannot feed the blocked kernel's paired loads: the CSR kernel steps in
    xo = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
    xo[1:] = dev(x)
    assert xo[1:].data_ptr() % 16 == 8
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(y, xo[1:], A)
    assert_bit_equal(y.cpu().numpy(), yr, "FE matrix, bcsr4 requested, misaligned x")
    # powers through the blocked kernel
    ys = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    mpk.SpMkV(ys, dev(x), A)
    Y = O.spmk_chain(3, p, c, v, x)
    for k in range(3):
        assert_bit_equal(ys[k].cpu().numpy(), Y[k], f"FE power {k + 1} via bcsr4")
    # a matrix without the structure refuses the request loudly
    p2, c2, v2 = synth.rows("s15", 4000)
    B = mpk.csrmatrix(4000, p2, c2, v2)
    _ = B.handle
    with pytest.raises(mpk.MiError):
        B.set_kernel("bcsr4")


@pytest.mark.parametrize("kind,n,w,ranks,dense,mode", [("s15", 240_000, 2000, 4, "1", "events"), ("svar", 90_000, 2000, 3, "1", "events"),
                                                         ("sfe", 64_000, 1500, 2, "1", "flags"), ("s15", 150_000, 2000, 3, "0", "flags"),
                                                         ("s15", 480_000, 2000, 8, "1", "events")])
def test_native_step_multirank_threads(kind, n, w, ranks, dense, mode):
    """The library's native multi-rank step (C++: pack + exchange on a comm stream, interior beside, boundary
    behind) with `ranks` ranks as threads on this one GPU and tests/fake_rccl in place of librccl, which refuses
    two ranks per device.  Bitwise for 4 chained powers and for 40 unsynchronised repetitions of one step."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(fake), "run __graft_entry__.build() first (it builds tests/fake_rccl)"
    # dense = "1": ghost ranges, contiguous slices of x sent in place; "0": exact ghost sets, pack kernel + send buffer
    # mode: "events" / "flags" = cross-stream hand-offs of the RCCL step (mi_part_spmv_dev).  (The peer-push step is NOT
    # run with ranks as threads: its wait kernel spins until ANOTHER rank's push kernel has run, and the streams of one
    # process share a few hardware queues, so a waiting kernel can sit in front of the kernel it waits for — observed
    # as a hang with 4 rank threads.  One process per GPU, the deployment, has its own queues per rank: the
    # process-based test_ranks_sharing_one_card[push] cases cover it.)
    env = dict(os.environ, MI355_RCCL_LIBRARY=fake, OMP_NUM_THREADS="1", MI355_PART_DENSE_HALO=dense, MI355_PART_HANDOFF=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native_threads_worker.py"), kind, str(n), str(w), str(ranks)],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "NATIVE_THREADS_RESULT" in r.stdout


@pytest.mark.parametrize("kind,n,ranks", [("fe", 20, 8), ("fe", 16, 3), ("s15", 240_000, 4)])
def test_native_step_allgather_form_multirank_threads(kind, n, ranks):
    """The ALL-GATHER form of the RCCL step (mi_part_allgather_setup; BASELINE north_star's collective): every rank contributes the
    slice of its entries that anybody needs, one ncclAllGather, ghosts picked out of the gathered buffer — for the FE slab partition
    (a whole mesh plane per neighbour) over 8 rank threads with the in-process librccl stand-in, and for a band.  Every power of
    A x .. A^4 x and 40 unsynchronised repetitions bitwise against the oracle on every rank."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        pytest.skip("tests/fake_rccl not built")
    env = dict(os.environ, MI355_RCCL_LIBRARY=fake, OMP_NUM_THREADS="1", MI355_TEST_ALLGATHER="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native_threads_worker.py"), kind, str(n), "2000", str(ranks)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "NATIVE_THREADS_RESULT" in r.stdout


@pytest.mark.parametrize("kind,n", [("s15", 300_000), ("svar", 250_000)])
def test_ring_variants_are_bit_identical(kind, n, monkeypatch):
    """Every instantiation of the ring kernel mi_csr_create may pick — temporal / non-temporal value loads,
    plain / padded staging layout, the four block shapes — returns the same bits as the oracle's fma chain."""
    import ctypes
    p, c, v = synth.rows(kind, n)
    x = synth.x_sin(0, n)
    yr = O.spmv(p, c, v, x)
    L = mpk.lib()
    seen = set()
    for cfg in ("4", "1", "2", "3"):
        for skew in ("0", "1"):
            monkeypatch.setenv("MI355_RING_CONFIG", cfg)
            monkeypatch.setenv("MI355_RING_SKEW", skew)
            A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
            for nt in (0, 1):
                mpk.check(L.mi_csr_set_nontemporal(A.handle, nt, -1))
                y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
                mpk.SpMV_CSR(y, dev(x), A)
                assert_bit_equal(y.cpu().numpy(), yr, f"{kind} cfg {cfg} skew {skew} nt {nt}: {A.kernel_name()}")
                seen.add(A.kernel_name())
    assert len(seen) == 16, seen


def test_bcsr_matrix_powers():
    """SpM2V_BCSR (mpk/SpM2V.cpp:375-801) on the blocked FE matrix: y = A x, z = A^2 x, device and host entry
    points, bitwise against two oracle BCSR products (= two CSR fma-chain products of the same matrix)."""
    p, c, v = synth.fe_matrix(10)
    n = len(p) - 1
    bp, bc, bv = synth.csr_to_bcsr4(p, c, v)
    A = mpk.bcsr4x4_matrix(n // 4, bp, bc, bv, nbcols=n // 4)
    x = synth.x_sin(0, n)
    Y = O.spmk_chain(2, p, c, v, x)
    yb = O.spmv_bcsr4(bp, bc, bv, x)
    assert_bit_equal(yb, Y[0], "oracle BCSR product == oracle CSR product on the blocked matrix")
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    z = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    mpk.SpM2V_BCSR(z, y, dev(x), A)
    assert_bit_equal(y.cpu().numpy(), Y[0], "SpM2V_BCSR y (device)")
    assert_bit_equal(z.cpu().numpy(), Y[1], "SpM2V_BCSR z (device)")
    yh, zh = np.full(n, np.nan), np.full(n, np.nan)
    mpk.SpM2V_BCSR_FMA(zh, yh, x, A)
    assert_bit_equal(yh, Y[0], "SpM2V_BCSR y (host)")
    assert_bit_equal(zh, Y[1], "SpM2V_BCSR z (host)")


def _random_banded(rng, n, ncols, mean_len, band, len_mode):
    """Seeded CSR with a chosen row-length law and column band (columns ascending per row)."""
    if len_mode == "mult8":
        lens = 8 * rng.integers(0, max(2, mean_len // 4), n)
    elif len_mode == "const":
        lens = np.full(n, mean_len)
    elif len_mode == "spiky":
        lens = rng.integers(0, 2 * mean_len, n)
        lens[rng.integers(0, n, 3)] = rng.integers(2048, 5000, 3)   # rows longer than a block
    else:
        lens = rng.integers(0, 2 * mean_len + 1, n)
    lens = np.minimum(lens, min(ncols, 2 * band + 1)).astype(np.int64)
    cols, centre = [], np.linspace(0, ncols - 1, n).astype(np.int64)
    for i in range(n):
        lo, hi = max(0, centre[i] - band), min(ncols, centre[i] + band + 1)
        cols.append(np.sort(rng.choice(np.arange(lo, hi), int(min(lens[i], hi - lo)), replace=False)))
        lens[i] = len(cols[-1])
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    c = (np.concatenate(cols) if p[-1] else np.zeros(0)).astype(np.int32)
    return p, c, rng.uniform(-1, 1, int(p[-1]))


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_all_kernels_and_layouts(seed, monkeypatch):
    """Seeded sweep over shapes the tid-strided (unclamped) loads and the padded arrays must survive: sizes that are
    multiples of nothing, rectangular matrices, row lengths that are multiples of 8 (padded staging layout),
    rows longer than a block, narrow and wide bands, row-mapped pieces — every kernel, temporal and non-temporal,
    bitwise against the oracle."""
    import ctypes
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 40000))
    ncols = n if seed % 2 == 0 else int(rng.integers(max(1, n // 2), 2 * n + 2))
    mode = ["uniform", "mult8", "const", "spiky", "uniform", "mult8"][seed]
    band = [300, 2000, 50, 1500, 30000, 6000][seed]
    p, c, v = _random_banded(rng, n, ncols, int(rng.integers(3, 30)), band, mode)
    x = rng.uniform(-1, 1, ncols)
    yr = O.spmv(p, c, v, x)
    L = mpk.lib()
    for cfg in ("4", "2"):
        monkeypatch.setenv("MI355_RING_CONFIG", cfg)
        for mapped in (False, True, "offset"):
            # a scattered rowmap (gather per row) and a contiguous one (applied as a pointer offset)
            rowmap = (np.arange(n) + 5).astype(np.int32) if mapped == "offset" else rng.permutation(n + 7)[:n].astype(np.int32)
            A = mpk.csrmatrix(n, p, c, v, ncols=ncols, rowmap=rowmap if mapped else None)
            for kernel in ("ring", "stream", "rowpar", "auto"):
                A.set_kernel(kernel)
                for nt in (0, 1):
                    mpk.check(L.mi_csr_set_nontemporal(A.handle, nt, nt))
                    y = torch.full((n + 7,), float("nan"), dtype=torch.float64, device="cuda")
                    mpk.SpMV_CSR(y, dev(x), A)
                    got = y.cpu().numpy()
                    got = got[rowmap] if mapped else got[:n]
                    assert_bit_equal(got, yr, f"seed {seed} n={n} ncols={ncols} {mode} band={band} cfg={cfg} mapped={mapped} {kernel} nt={nt}")
                    if not mapped:
                        assert np.isnan(y.cpu().numpy()[n:]).all(), "wrote past y"


def test_unsorted_and_repeated_columns():
    """CSR rows whose columns are neither ascending nor unique (generate_CSR never produces them, a caller's arrays
    might): every kernel still evaluates the row's terms in storage order, like the reference's loop."""
    rng = np.random.default_rng(77)
    n, ncols, band = 30_000, 30_000, 1500
    lens = rng.integers(0, 24, n)
    p = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    c = np.concatenate([np.clip(rng.integers(-band, band + 1, l) + i, 0, ncols - 1) for i, l in enumerate(lens)]).astype(np.int32)
    v = rng.uniform(-1, 1, int(p[-1]))
    x = rng.uniform(-1, 1, ncols)
    yr = O.spmv(p, c, v, x)
    for kernel in KERNELS + ["auto"]:
        A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(y, dev(x), A)
        assert_bit_equal(y.cpu().numpy(), yr, f"unsorted/repeated columns, {kernel} -> {A.kernel_name()}")


@pytest.mark.parametrize("seed", range(12))
def test_random_patterns_every_kernel(seed):
    """the planner fuzz (tests/test_planner_fuzz.py) on the GPU: rows of wildly different lengths, wandering / jumping / scattered
    column clusters, unsorted rows, duplicates — every kernel against the oracle's fma chain, bit for bit"""
    from test_planner_fuzz import random_pattern
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([7, 300, 5000, 20000, 60000]))
    p, c = random_pattern(rng, n)
    v = rng.uniform(-1, 1, len(c))
    x = rng.uniform(-1, 1, n)
    yr = O.spmv(p, c, v, x)
    for kernel in KERNELS + ["auto"]:
        A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(y, dev(x), A)
        assert_bit_equal(y.cpu().numpy(), yr, f"seed {seed} n {n} {kernel} -> {A.kernel_name()}")


@pytest.mark.parametrize("force", ["0", "1"])
def test_blocked_kernel_with_and_without_the_x_tile(force, monkeypatch):
    """spmv_bcsr4 (x blocks through L1/L2) and spmv_bcsr4_tile (each workgroup's distinct block columns once into LDS): the same
    lanes and fma order, so the same bits — on an FE matrix whose block-row count is no multiple of 64, with empty block rows, and
    through the CSR entry point that runs the blocked copy."""
    monkeypatch.setenv("MI355_BCSR_TILE", force)
    p, c, v = synth.fe_matrix(9, 7, 5)                       # 480 nodes: 7.5 workgroups of 64 block rows
    n = len(p) - 1
    x = synth.x_sin(0, n)
    yr = O.spmv(p, c, v, x)
    bp, bc, bv = synth.csr_to_bcsr4(p, c, v)
    B = mpk.bcsr4x4_matrix(n // 4, bp, bc, bv, nbcols=n // 4)
    y = np.full(n, np.nan)
    mpk.SpMV_BCSR(y, x, B)
    assert_bit_equal(y, yr, f"BCSR API, tile={force}")
    A = mpk.csrmatrix(n, p, c, v).set_kernel("bcsr4")
    assert ("tile" in A.kernel_name()) == (force == "1")
    yd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(yd, dev(x), A)
    assert_bit_equal(yd.cpu().numpy(), yr, f"CSR API on the blocked copy, tile={force}")
    # empty block rows in the middle and at the end
    keep = np.ones(n // 4, bool)
    keep[[3, 100, 101, n // 4 - 1]] = False
    lens = np.diff(bp) * keep
    bp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    sel = np.repeat(keep, np.diff(bp))
    bc2, bv2 = bc[sel], bv.reshape(-1, 16)[sel].ravel()
    B2 = mpk.bcsr4x4_matrix(n // 4, bp2, bc2, bv2, nbcols=n // 4)
    mpk.SpMV_BCSR(y, x, B2)
    assert_bit_equal(y, O.spmv_bcsr4(bp2, bc2, bv2, x), f"empty block rows, tile={force}")


@pytest.mark.parametrize("kind,n,kernel,expect_fused", [("s15", 300_000, "ring", True), ("svar", 200_000, "ring", True), ("s15", 300_000, "stream", False),
                                                         ("s15", 3_000, "ring", True), ("sfe", 40_000, "auto", None),
                                                         ("s15", 300_000, "sstream", False), ("s15", 70_001, "sstream:3", False), ("s15", 2_049, "sstream:0", False)])
def test_product_with_the_dot_in_its_epilogue(kind, n, kernel, expect_fused, monkeypatch):
    """mi_spmv_dot_dev / mi_spmv_orthogonalize_dev (the f-4 pipeline SpMV -> dot + AXPY -> SpMV of mpk/SpMVmulti.cpp:563-569 with
    the dot folded into the product): every row of y is the oracle's fma chain whether or not the launch carried the dot; beta
    is a fixed-tree reduction inside the bound all reductions of the library are held to; GIVEN beta the update is the
    reference's fma bit for bit."""
    if kernel.startswith("sstream"):  # round 4: a sliced-stream handle takes product + separate dot (its own epilogue measured slower: NOTES R4.4); ":f" forces a variant
        monkeypatch.setenv("MI355_SSTREAM", "1")
        if ":" in kernel:
            monkeypatch.setenv("MI355_SSTREAM_FORM", kernel.split(":")[1])
        kernel = "sstream"
    p, c, v = synth.rows(kind, n, w=min(2000, max(8, n // 8))) if kernel == "sstream" else synth.rows(kind, n)
    x = synth.x_sin(0, n)
    b = np.cos(0.002 * np.arange(n))
    A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
    if expect_fused is not None:
        assert A.dot_in_epilogue() == expect_fused, A.kernel_name()
    yo = O.spmv(p, c, v, x)
    bound = 1e-13 * float(np.abs(b * yo).sum())
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    beta = mpk.SpMV_CSR_dot(y, dev(x), A, dev(b))
    assert_bit_equal(y.cpu().numpy(), yo, f"product with dot epilogue ({A.kernel_name()})")
    assert abs(float(beta) - O.dot(b, yo)) <= bound, (float(beta), O.dot(b, yo), bound)
    b2 = float(beta)
    beta_again = mpk.SpMV_CSR_dot(y, dev(x), A, dev(b))
    assert float(beta_again) == b2, "the reduction is deterministic run to run"
    x1 = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    x3 = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    beta3 = mpk.SpMV_CSR_orthogonalize(x1, dev(x), A, dev(b), x3, 1e-8)
    assert float(beta3) == b2
    assert_bit_equal(x1.cpu().numpy(), yo, "x1 = A x")
    assert_bit_equal(x3.cpu().numpy(), O.ortho_update(1e-8 * b2, b, yo), "x3 given the device's beta")
    # the pipeline's second product on x3
    x2 = torch.empty(n, dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(x2, x3, A)
    assert_bit_equal(x2.cpu().numpy(), O.spmv(p, c, v, x3.cpu().numpy()), "x2 = A x3")


def test_column_major_blocks_of_a_petsc_baij_matrix():
    """The PETSc seam (integration/petsc_matmult_mi355.c): MATSEQBAIJ stores a 4x4 block column-major
    (src/kernels/baij4_mad.c:73-76); mi_bcsr4_create_layout(MI_BLOCK_COLMAJOR) takes the array as it is.  The product must be
    bit-equal to the row-major handle of the same matrix and to the oracle, also after a value refresh in either form."""
    p, c, v = synth.fe_matrix(12)
    n = len(p) - 1
    bp, bc, bv = synth.csr_to_bcsr4(p, c, v)                       # row-major blocks (mpk layout)
    bv_col = np.ascontiguousarray(bv.reshape(-1, 4, 4).transpose(0, 2, 1)).reshape(-1)  # what Mat_SeqBAIJ::a holds
    x = synth.x_sin(0, n)
    yo = O.spmv_bcsr4(bp, bc, bv, x)
    A_row = mpk.bcsr4x4_matrix(n // 4, bp, bc, bv, nbcols=n // 4)
    A_col = mpk.bcsr4x4_matrix(n // 4, bp, bc, bv_col, nbcols=n // 4, layout="col")
    for A, what in ((A_row, "row-major"), (A_col, "column-major")):
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_BCSR(y, dev(x), A)
        assert_bit_equal(y.cpu().numpy(), yo, f"{what} blocks")
    yh = np.full(n, np.nan)
    mpk.SpMV_BCSR(yh, x, A_col)  # the host-vector call the PETSc glue makes
    assert_bit_equal(yh, yo, "column-major blocks, host vectors")
    # a Newton step rewrites the values: refresh from a host array and from a device array, column-major both
    bv2 = bv * np.cos(np.arange(len(bv)))
    bv2_col = np.ascontiguousarray(bv2.reshape(-1, 4, 4).transpose(0, 2, 1)).reshape(-1)
    yo2 = O.spmv_bcsr4(bp, bc, bv2, x)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    A_col.update_values(bv2_col)
    mpk.SpMV_BCSR(y, dev(x), A_col)
    assert_bit_equal(y.cpu().numpy(), yo2, "after a host refresh, column-major")
    A_col.update_values(dev(bv_col))
    mpk.SpMV_BCSR(y, dev(x), A_col)
    assert_bit_equal(y.cpu().numpy(), yo, "after a device refresh, column-major")
    # the association the PETSc AVX2 kernel uses is NOT this chain: four per-column accumulators added at the end
    # (src/kernels/baij4_avx2.c:42-66) — agreement to rounding only, documented as parity-unpinned (no PETSc here)
    acc = np.zeros((4, n))
    rows = np.repeat(np.arange(n // 4), np.diff(bp))
    blk = bv.reshape(-1, 4, 4)
    for j in range(4):
        np.add.at(acc[j].reshape(-1, 4), rows, blk[:, :, j] * x.reshape(-1, 4)[bc][:, j:j + 1])
    y_avx2_like = (acc[0] + acc[1]) + (acc[2] + acc[3])
    assert O.rel_error(yo, y_avx2_like) < 1e-14


@pytest.mark.parametrize("kind,n,w", [("s15", 300_000, 2000), ("svar", 250_000, 2000), ("s15", 1_000_000, 2000), ("s15", 40_000, 300)])
def test_matrix_powers_in_one_launch(kind, n, w, monkeypatch):
    """The one-launch k-step (spmk_ring.hpp; SpM2V_CSR mpk/SpM2V.cpp:79-112, SpM3V / SpM4V mpk/SpMVmulti0.cpp:132-221): every
    workgroup keeps its run for all powers, power p + 1 waits for the flags of the runs its columns name.  Forced on
    (MI355_SPMK_FUSED=1): every power bit-equal to the chained oracle, for k = 2..8, repeatedly on one handle (the flags count on
    from launch to launch) with a different x each time, and equal to what k launches give (MI355_SPMK_FUSED=0)."""
    p, c, v = synth.rows(kind, n, w=w)
    A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
    xs = [synth.x_sin(0, n), np.cos(0.002 * np.arange(n)), synth.x_sin(0, n) * 0.5 - 0.25]
    monkeypatch.setenv("MI355_SPMK_FUSED", "1")
    outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(8)]
    for rep, k in enumerate((4, 2, 3, 8, 4, 4, 2)):
        x = xs[rep % 3]
        for t in outs:
            t.fill_(float("nan"))
        mpk.SpMkV(outs[:k], dev(x), A)
        info = A.spmk_info(k)
        assert info["eligible"] and info["one_launch"], (info, A.kernel_name(), A.ring_shape_info())
        Y = O.spmk_chain(k, p, c, v, x)
        for q in range(k):
            assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"one launch, rep {rep}, k={k}, power {q + 1}")
    monkeypatch.setenv("MI355_SPMK_FUSED", "0")
    mpk.SpMkV(outs[:4], dev(xs[0]), A)
    Y = O.spmk_chain(4, p, c, v, xs[0])
    for q in range(4):
        assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"k launches, power {q + 1}")
    # the measured choice (no override): whichever form wins, the bits are the same and the handle says what it does
    monkeypatch.delenv("MI355_SPMK_FUSED")
    mpk.SpMkV(outs[:4], dev(xs[1]), A)
    info = A.spmk_info(4)
    assert info["us_k_launches"] > 0 and info["us_one_launch"] > 0, info
    Y = O.spmk_chain(4, p, c, v, xs[1])
    for q in range(4):
        assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"measured choice {info}, power {q + 1}")


@pytest.mark.parametrize("n,hb", [(1_000_000, 1), (300_000, 3)])
def test_matrix_powers_in_one_launch_on_a_narrow_band(n, hb, monkeypatch):
    """with a narrow band (tridiagonal at 1 M rows) a run's rows plus band span fewer columns than the window's first
    fill, which used to load all 5120 entries from its first column unconditionally — lines of y_p owned by runs that are NOT on
    the dependency list, possibly before their publication.  The fill is now clamped to the first block's new columns; every
    power of the one-launch k = 4 step must equal the chained oracle, launch after launch on one handle."""
    i = np.arange(n, dtype=np.int64)
    cols = np.stack([i + d for d in range(-hb, hb + 1)], axis=1)
    ok = (cols >= 0) & (cols < n)
    p = np.concatenate([[0], np.cumsum(ok.sum(axis=1))]).astype(np.int32)
    c = cols[ok].astype(np.int32)
    rng = np.random.default_rng(5)
    v = rng.uniform(-0.4, 0.4, len(c))
    A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
    monkeypatch.setenv("MI355_SPMK_FUSED", "1")
    outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
    for rep in range(6):
        x = np.cos(0.001 * (rep + 1) * np.arange(n))
        for t in outs:
            t.fill_(float("nan"))
        mpk.SpMkV(outs, dev(x), A)
        info = A.spmk_info(4)
        assert info["eligible"] and info["one_launch"], info
        Y = O.spmk_chain(4, p, c, v, x)
        for q in range(4):
            assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"narrow band hb={hb}, rep {rep}, power {q + 1}")


def test_matrix_powers_in_one_launch_is_refused_where_it_cannot_run(monkeypatch):
    """Handles the one-launch form is not built for (stream kernel, FE matrices on the blocked kernel, a band much wider than a
    run) take k launches even when it is forced on — same bits."""
    monkeypatch.setenv("MI355_SPMK_FUSED", "1")
    for kind, n, w, kernel in (("s15", 100_000, 2000, "stream"), ("sfe", 40_000, 1500, "auto"), ("s15", 200_000, 60_000, "auto"),
                               ("s15", 120_000, 4500, "ring")):  # (the last: ring configuration 3, 512 threads — no one-launch instantiation)
        p, c, v = synth.rows(kind, n, w=w)
        A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
        x = synth.x_sin(0, n)
        outs = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3)]
        mpk.SpMkV(outs, dev(x), A)
        assert not A.spmk_info(3)["one_launch"], (kind, A.kernel_name())
        Y = O.spmk_chain(3, p, c, v, x)
        for q in range(3):
            assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"{kind} {kernel} power {q + 1}")


@pytest.mark.parametrize("seed", range(10))
def test_random_patterns_round3_paths(seed, monkeypatch):
    """The planner fuzz's random patterns (rows of wildly different lengths, wandering / jumping column clusters, duplicates,
    unsorted rows) through round 3's entry points: the one-launch powers step forced on (it runs where the plan allows and the
    handle takes k launches elsewhere), the product with the dot in its epilogue, and the internal-numbering calls — bitwise
    against the oracle (beta inside its bound)."""
    from test_planner_fuzz import random_pattern
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([5000, 20000, 60000, 150000]))
    p, c = random_pattern(rng, n)
    v = rng.uniform(-1, 1, len(c)) / max(1.0, float(np.diff(p).max()))  # keep A^k x finite for long rows
    x = rng.uniform(-1, 1, n)
    b = rng.uniform(-1, 1, n)
    monkeypatch.setenv("MI355_SPMK_FUSED", "1")
    for kernel in ("auto", "ring"):
        A = mpk.csrmatrix(n, p, c, v).set_kernel(kernel)
        Y = O.spmk_chain(3, p, c, v, x)
        for rep in range(2):  # twice: the flags count on
            outs = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3)]
            mpk.SpMkV(outs, dev(x), A)
            for q in range(3):
                assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"seed {seed} {kernel} -> {A.kernel_name()} {A.spmk_info(3)} rep {rep} power {q + 1}")
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        beta = mpk.SpMV_CSR_dot(y, dev(x), A, dev(b))
        assert_bit_equal(y.cpu().numpy(), Y[0], f"seed {seed} {kernel}: product with dot (epilogue={A.dot_in_epilogue()})")
        assert abs(float(beta) - O.dot(b, Y[0])) <= 1e-13 * float(np.abs(b * Y[0]).sum()) + 1e-300
        xi = A.to_internal(dev(x))
        yi = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR_internal(yi, xi, A)
        assert_bit_equal(A.from_internal(yi).cpu().numpy(), Y[0], f"seed {seed} {kernel}: internal numbering")


def test_powers_step_under_hip_graph_capture_is_recorded_as_k_launches(monkeypatch):
    """A captured k-step must not bake the one-launch step's flag epoch into a graph (a replay would present the same epoch and every
    in-kernel wait would pass at once): under capture mi_spmk_dev records k plain launches.  Replayed five times with a different x
    each time: every power bitwise."""
    monkeypatch.setenv("MI355_SPMK_FUSED", "1")
    n = 300_000
    p, c, v = synth.rows("s15", n)
    A = mpk.csrmatrix(n, p, c, v).set_kernel("ring")
    xs = [synth.x_sin(0, n), np.cos(0.002 * np.arange(n)), synth.x_sin(0, n) * 0.5 - 0.25]
    xd = dev(xs[0]).clone()
    outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    mpk.SpMkV(outs, xd, A)  # eager first: one launch, plans uploaded outside any capture
    assert A.spmk_info(3)["one_launch"]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            mpk.SpMkV(outs, xd, A)
    for rep in range(5):
        xd.copy_(dev(xs[rep % 3]))
        for t in outs:
            t.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        Y = O.spmk_chain(3, p, c, v, xs[rep % 3])
        for q in range(3):
            assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"graph replay {rep}, power {q + 1}")


@pytest.mark.parametrize("which", ["sstream", "bcsr4_sell", "bcsr4_api"])
def test_graph_replay_after_a_value_update_reads_the_new_values(which, monkeypatch):
    """A product captured into a HIP graph BEFORE a value update (a Newton loop's Jacobian, src/solve_newton.c:1245-1247) must return
    A_new x when replayed after it: the graph holds only the product's node, so the sliced copies the round-4 kernels read (spmv_sstream's,
    spmv_bcsr4_sell's) have to be refilled by the update itself, on its stream — not lazily in front of the next eager product (ADVICE r4,
    high).  Update from a device array on the capture's stream, then from a host array; then a captured k = 2 step over the same handle."""
    monkeypatch.setenv("MI355_SSTREAM", "1")
    monkeypatch.setenv("MI355_BCSR_SELL", "1")
    side = torch.cuda.Stream()
    if which == "sstream":
        n = 300_000
        p, c, v = synth.rows("s15", n)
        A = mpk.csrmatrix(n, p, c, v).set_kernel("sstream")
        ref = lambda vals, x: O.spmv(p, c, vals, x)
        prod = lambda y, x: mpk.SpMV_CSR(y, x, A)
    else:
        p, c, v = synth.fe_matrix(20)
        n = len(p) - 1
        if which == "bcsr4_sell":
            A = mpk.csrmatrix(n, p, c, v).set_kernel("bcsr4")
            ref = lambda vals, x: O.spmv(p, c, vals, x)
            prod = lambda y, x: mpk.SpMV_CSR(y, x, A)
        else:
            bp, bc, v = synth.csr_to_bcsr4(p, c, v)
            A = mpk.bcsr4x4_matrix(n // 4, bp, bc, v, nbcols=n // 4)
            ref = lambda vals, x: O.spmv_bcsr4(bp, bc, vals, x)
            prod = lambda y, x: mpk.SpMV_BCSR(y, x, A)
    x = synth.x_sin(0, n)
    xd = dev(x)
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    prod(y, xd)  # eager first
    if which != "bcsr4_api":
        assert ("sstream" if which == "sstream" else "sell") in A.kernel_name(), A.kernel_name()
    assert_bit_equal(y.cpu().numpy(), ref(v, x), "eager product")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            prod(y, xd)
    v2 = v * np.cos(np.arange(len(v)))
    v3 = v * np.sin(1.0 + np.arange(len(v)))
    with torch.cuda.stream(side):
        y.fill_(float("nan"))
        A.update_values(dev(v2))  # device array, on the stream the graph is replayed on
        g.replay()
    torch.cuda.synchronize()
    assert_bit_equal(y.cpu().numpy(), ref(v2, x), "graph replay after a device-side value update")
    A.update_values(v3)           # host array (synchronous)
    with torch.cuda.stream(side):
        y.fill_(float("nan"))
        g.replay()
    torch.cuda.synchronize()
    assert_bit_equal(y.cpu().numpy(), ref(v3, x), "graph replay after a host-side value update")
    if which == "sstream":  # a captured k-step on a sliced-stream handle records k of its launches
        outs = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
        mpk.SpMkV(outs, xd, A)
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g2, stream=side):
                mpk.SpMkV(outs, xd, A)
        with torch.cuda.stream(side):
            A.update_values(dev(v2))
            for t in outs:
                t.fill_(float("nan"))
            g2.replay()
        torch.cuda.synchronize()
        Y = O.spmk_chain(2, p, c, v2, x)
        for q in range(2):
            assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"captured k-step after a value update, power {q + 1}")


def test_mring_with_second_round_workgroups_is_bitwise(monkeypatch):
    """A relabelled 0.75 M-row mesh operator through the multi-window ring kernel: its plan holds more runs than the 512 workgroups the
    GPU keeps resident (forced cuts leave short runs, dealt out behind the long ones: mring_plan.hpp), so the launch has workgroups
    that start late.  Natural order beside it (one round).  Both bitwise against the oracle, twice (the second launch warm)."""
    import ctypes
    from test_ring_plan import relabelled
    from test_mring_plan import deal
    monkeypatch.setenv("MI355_REORDER", "0")
    monkeypatch.setenv("MI355_SPMV_KERNEL", "mring")
    p, c, v = synth.pressure_matrix(120, 100, 60)
    ps, cs, vs = synth.permute_nodes(p, c, v, block=1)[:3]
    p2, c2 = relabelled(np.ascontiguousarray(ps, np.int32), np.ascontiguousarray(cs, np.int32))
    v2 = np.cos(np.arange(len(c2)) * 0.37) + 1.5
    late = 0
    for pp, cc, vv, tag in ((p, c, v, "natural"), (p2, c2, v2, "relabelled")):
        n = len(pp) - 1
        T = deal(pp, cc)
        late += int((T[:, 64:] > 0).sum())
        A = mpk.csrmatrix(n, pp, cc, vv)
        assert "mring" in A.kernel_name(), A.kernel_name()
        x = synth.x_sin(0, n)
        d = dev(x)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        yo = O.spmv(pp, cc, vv, x)
        for rep in range(2):
            mpk.SpMV_CSR(y, d, A)
            assert_bit_equal(y.cpu().numpy(), yo, f"mring {tag} launch {rep}")
            y.fill_(float("nan"))
        A.close()
    assert late > 0, "the test matrix no longer produces second-round workgroups: pick another"


@pytest.mark.parametrize("form", ["0", "1", "2", "3"])
def test_blocked_product_from_the_sliced_copy(form, monkeypatch):
    """spmv_bcsr4_sell (spmv_bcsr_sell.hpp; SpMV_BCSR*, mpk/SpMV.cpp:90-219): the sliced copy of the block values streamed by
    persistent waves.  Each variant forced (MI355_BCSR_SELL_FORM): bit-equal to the oracle's SpMV_BCSR_FMA restatement on the FE matrix
    (through the BCSR API and, as the blocked copy of a CSR handle, through SpMV_CSR), on ragged patterns — empty block rows, a row count
    that is no multiple of the slice, rows much shorter than their slice (padding places read x at node 0 and must not be multiplied: x is
    infinite there), more slices than waves and fewer —, and after value refreshes (the sliced copy follows the blocks)."""
    monkeypatch.setenv("MI355_BCSR_SELL", "1")
    monkeypatch.setenv("MI355_BCSR_SELL_FORM", form)
    L = mpk.lib()
    import ctypes

    def sell_form(handle):
        b, f = ctypes.c_int(), ctypes.c_int()
        mpk.check(L.mi_bcsr4_sell_info(handle, ctypes.byref(b), ctypes.byref(f), None, None, None))
        return b.value, f.value

    # the FE matrix, BCSR API
    p, c, v = synth.fe_matrix(20)
    n = len(p) - 1
    bp, bc, bv = synth.csr_to_bcsr4(p, c, v)
    x = synth.x_sin(0, n)
    B = mpk.bcsr4x4_matrix(n // 4, bp, bc, bv, nbcols=n // 4)
    assert sell_form(B.handle) == (1, int(form))
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    for _ in range(3):
        mpk.SpMV_BCSR(y, dev(x), B)
    assert_bit_equal(y.cpu().numpy(), O.spmv_bcsr4(bp, bc, bv, x), f"FE matrix, sliced form {form}")
    bv2 = bv * np.cos(np.arange(len(bv)))
    B.update_values(bv2)
    mpk.SpMV_BCSR(y, dev(x), B)
    assert_bit_equal(y.cpu().numpy(), O.spmv_bcsr4(bp, bc, bv2, x), "after mi_bcsr4_update_values")
    # the same matrix as a CSR handle: AUTO runs the blocked copy (forced here), values refreshed through mi_csr_update_values
    A = mpk.csrmatrix(n, p, c, v).set_kernel("bcsr4")
    y1 = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(y1, dev(x), A)
    assert "sell" in A.kernel_name(), A.kernel_name()
    assert_bit_equal(y1.cpu().numpy(), O.spmv(p, c, v, x), "CSR API over the sliced blocked copy")
    v2 = v * np.sin(1.0 + np.arange(len(v)))
    A.update_values(v2)
    mpk.SpMV_CSR(y1, dev(x), A)
    assert_bit_equal(y1.cpu().numpy(), O.spmv(p, c, v2, x), "after mi_csr_update_values")
    # ragged block patterns
    rng = np.random.default_rng(11)
    for nbr, nbc, maxlen in ((1, 7, 5), (15, 40, 9), (16, 16, 1), (17, 300, 30), (1000, 1000, 12), (70_001, 70_001, 6)):
        lens = rng.integers(0, maxlen + 1, nbr)
        lens[rng.random(nbr) < 0.2] = 0            # empty block rows
        if nbr > 40:
            lens[nbr // 2] = 4 * maxlen            # one long row: its slice is mostly padding
        lens = np.minimum(lens, nbc)
        bp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        lens = np.minimum(lens, nbc - 1)
        bp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        # no block names block column 0, and x is infinite there: padding places read x at node 0 and must not multiply it (0 * inf = NaN)
        bc = np.concatenate([1 + rng.choice(nbc - 1, size=l, replace=False) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        bv = rng.uniform(-1, 1, 16 * len(bc))
        xx = rng.uniform(-1, 1, 4 * nbc)
        xx[:4] = np.inf
        B = mpk.bcsr4x4_matrix(nbr, bp, bc, bv, nbcols=nbc)
        yy = torch.full((4 * nbr,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_BCSR(yy, dev(xx), B)
        assert sell_form(B.handle) == (1, int(form))
        assert_bit_equal(yy.cpu().numpy(), O.spmv_bcsr4(bp, bc, bv, xx), f"ragged {nbr} x {nbc}, sliced form {form}")


def _band(n, ncols, hb, per, seed, empty_every=0):
    """rows with `per` distinct columns within +-hb of the (scaled) diagonal, ascending; every `empty_every`-th row empty"""
    rng = np.random.default_rng(seed)
    ptr, col = [0], []
    for i in range(n):
        if empty_every and i % empty_every == 3:
            ptr.append(len(col))
            continue
        c0 = int(i * (ncols - 1) / max(n - 1, 1))
        lo, hi = max(0, c0 - hb), min(ncols - 1, c0 + hb)
        k = min(per, hi - lo + 1)
        col.extend(sorted(rng.choice(np.arange(lo, hi + 1), size=k, replace=False).tolist()))
        ptr.append(len(col))
    return np.array(ptr, np.int32), np.array(col, np.int32), rng.uniform(-1, 1, len(col))


@pytest.mark.parametrize("form", ["0", "1", "2", "3"])
def test_mesh_operator_through_the_cut_ring_sliced_stream(form, monkeypatch):
    """Round 5: spmv_sstream_mw (spmv_sstream_mw.hpp) — the sliced stream with its LDS ring cut into four sub-rings, for rows that name
    several column neighbourhoods: the P1 pressure operator of src/integration.c on Kuhn meshes in natural node order (a node's plane and
    the two next to it).  Forced in each of the four variants: bit-equal to the oracle's fma chain (SpMV_CSR_FMA, mpk/SpMV.cpp:41-56) at
    two sizes (the planes must lie further apart than the one-window form's ring holds, or that form takes the matrix), after a value
    refresh, as the kernel behind the k = 3 powers, through a y that is only 8-byte aligned (that product takes another kernel) and through
    row maps (a partition piece's rows: odd and even offsets, scattered)."""
    monkeypatch.setenv("MI355_SSTREAM", "1")
    monkeypatch.setenv("MI355_SSTREAM_FORM", form)
    for cells in (62, 75):
        p, c, v = synth.pressure_matrix(cells)
        n = len(p) - 1
        A = mpk.csrmatrix(n, p, c, v).set_kernel("sstream")
        assert A.kernel_name().startswith("spmv_sstream_mw<") and A.sstream_info()["form"] == int(form), (A.kernel_name(), A.sstream_info())
        x = synth.x_sin(0, n)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        for _ in range(3):
            mpk.SpMV_CSR(y, dev(x), A)
        assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v, x), f"mesh {cells}^3 {A.kernel_name()}")
    v2 = v * np.cos(np.arange(len(v)))
    A.update_values(v2)
    mpk.SpMV_CSR(y, dev(x), A)
    assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v2, x), "after mi_csr_update_values")
    outs = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3)]
    for _ in range(2):
        mpk.SpMkV(outs, dev(x), A)
    Y = O.spmk_chain(3, p, c, v2, x)
    for q in range(3):
        assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"powers over a cut-ring handle, power {q + 1}: {A.spmk_info(3)}")
    ybig = torch.full((n + 1,), float("nan"), dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(ybig[1:], dev(x), A)
    assert_bit_equal(ybig[1:].cpu().numpy(), O.spmv(p, c, v2, x), "8-byte aligned y")
    p, c, v = synth.pressure_matrix(62)
    n = len(p) - 1
    x = synth.x_sin(0, n)
    yo = O.spmv(p, c, v, x)
    rng = np.random.default_rng(5)
    for tag, rowmap in (("offset 5", (np.arange(n) + 5).astype(np.int32)), ("offset 6", (np.arange(n) + 6).astype(np.int32)),
                        ("scattered", rng.permutation(n + 9)[:n].astype(np.int32))):
        A = mpk.csrmatrix(n, p, c, v, rowmap=rowmap).set_kernel("sstream")
        assert A.kernel_name().startswith("spmv_sstream_mw<"), A.kernel_name()
        yy = torch.full((n + 9,), float("nan"), dtype=torch.float64, device="cuda")
        for _ in range(2):
            mpk.SpMV_CSR(yy, dev(x), A)
        got = yy.cpu().numpy()
        assert_bit_equal(got[rowmap], yo, f"row-mapped ({tag}) {A.kernel_name()}")
        rest = np.ones(n + 9, bool)
        rest[rowmap] = False
        assert np.isnan(got[rest]).all(), f"row-mapped ({tag}): wrote rows outside the map"


@pytest.mark.parametrize("form", ["8nt", "8t", "12nt", "12t"])
def test_sliced_stream_kernel(form, monkeypatch):
    """spmv_sstream (spmv_sstream.hpp; SpMV_CSR*, mpk/SpMV.cpp:6-85): the sliced copy streamed by one wave per SIMD, a lane per row
    pair, x in a sliding LDS ring, y parked in LDS.  Forced on (MI355_SSTREAM=1 + set_kernel) in each of its four variants: bit-equal to the
    oracle's fma chain on S15 (sizes that are and are not multiples of the 512-row round, fewer rounds than workgroups and more), on ragged
    rows (empty rows, an odd row count: the last lane's second row does not exist), on a rectangular matrix, with x infinite where only
    padding places point, after value refreshes, through a y that is only 8-byte aligned (that product falls back to the ring kernel), and
    as the kernel behind the k = 4 powers step."""
    monkeypatch.setenv("MI355_SSTREAM", "1")
    L = mpk.lib()
    import ctypes

    def force(A):
        A.set_kernel("sstream")
        h = A.handle
        mpk.check(L.mi_csr_set_nontemporal(h, -1, -1))
        return A

    def variant(A):  # pick the variant by timing-free means: the info call reports the form in use; the env forces it at create
        return A.sstream_info()["form"]

    monkeypatch.setenv("MI355_SSTREAM_FORM", {"8nt": "0", "8t": "1", "12nt": "2", "12t": "3"}[form])
    cases = [("s15", 300_000, 2000), ("s15", 1_000_000, 2000), ("s15", 70_001, 900), ("s15", 3_000, 300), ("s15", 1_001, 20)]
    for kind, n, w in cases:
        p, c, v = synth.rows(kind, n, w=w)
        A = force(mpk.csrmatrix(n, p, c, v))
        assert "sstream" in A.kernel_name() and variant(A) == int({"8nt": 0, "8t": 1, "12nt": 2, "12t": 3}[form]), (A.kernel_name(), A.sstream_info())
        x = synth.x_sin(0, n)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        for _ in range(3):
            mpk.SpMV_CSR(y, dev(x), A)
        assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v, x), f"{kind} n={n} w={w} {A.kernel_name()}")
    # value refresh; the powers step on a handle whose products run the sliced stream; an 8-byte aligned y
    n = 400_000
    p, c, v = synth.rows("s15", n)
    A = force(mpk.csrmatrix(n, p, c, v))
    x = synth.x_sin(0, n)
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(y, dev(x), A)
    v2 = v * np.cos(np.arange(len(v)))
    A.update_values(v2)
    mpk.SpMV_CSR(y, dev(x), A)
    assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v2, x), "after mi_csr_update_values")
    outs = [torch.full((n,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(4)]
    for _ in range(2):
        mpk.SpMkV(outs, dev(x), A)
    Y = O.spmk_chain(4, p, c, v2, x)
    for q in range(4):
        assert_bit_equal(outs[q].cpu().numpy(), Y[q], f"powers over a sliced-stream handle, power {q + 1}: {A.spmk_info(4)}")
    ybig = torch.full((n + 1,), float("nan"), dtype=torch.float64, device="cuda")
    mpk.SpMV_CSR(ybig[1:], dev(x), A)  # y only 8-byte aligned
    assert_bit_equal(ybig[1:].cpu().numpy(), O.spmv(p, c, v2, x), "8-byte aligned y")
    # ragged rows, odd row count, rectangular, x infinite at a column nobody names (padding places point at slot 0x8000: never read)
    for n, ncols, hb, per, ee in ((50_001, 50_001, 700, 9, 7), (20_000, 26_000, 1500, 12, 0), (777, 777, 60, 5, 5)):
        p, c, v = _band(n, ncols, hb, per, seed=n, empty_every=ee)
        e, r_, st, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_double()
        mpk.check(L.mi_sstream_plan_probe(n, ncols, p.ctypes.data, c.ctypes.data, ctypes.byref(e), ctypes.byref(r_), ctypes.byref(st), ctypes.byref(pad)))
        if not e.value:  # (row lengths vary too much: the handle must refuse the kernel, loudly)
            with pytest.raises(mpk.MiError):
                force(mpk.csrmatrix(n, p, c, v, ncols=ncols))
            continue
        A = force(mpk.csrmatrix(n, p, c, v, ncols=ncols))
        x = np.random.default_rng(1).uniform(-1, 1, ncols)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        mpk.SpMV_CSR(y, dev(x), A)
        assert_bit_equal(y.cpu().numpy(), O.spmv(p, c, v, x), f"ragged {n} x {ncols}")
    # row-mapped handles (partition pieces, relabelled twins): a scattered map -> two 8-byte stores per lane; a contiguous one -> a pointer
    # offset, odd (row pairs no longer 16-byte aligned: that product falls back) and even
    n = 30_001
    p, c, v = synth.rows("s15", n, w=400)
    x = synth.x_sin(0, n)
    yo = O.spmv(p, c, v, x)
    rng = np.random.default_rng(9)
    for tag, rowmap in (("scattered", rng.permutation(n + 9)[:n].astype(np.int32)), ("offset 5", (np.arange(n) + 5).astype(np.int32)),
                        ("offset 6", (np.arange(n) + 6).astype(np.int32))):
        A = force(mpk.csrmatrix(n, p, c, v, rowmap=rowmap))
        y = torch.full((n + 9,), float("nan"), dtype=torch.float64, device="cuda")
        for _ in range(2):
            mpk.SpMV_CSR(y, dev(x), A)
        got = y.cpu().numpy()
        assert_bit_equal(got[rowmap], yo, f"row-mapped ({tag}) {A.kernel_name()}")
        rest = np.ones(n + 9, bool)
        rest[rowmap] = False
        assert np.isnan(got[rest]).all(), f"row-mapped ({tag}): wrote rows outside the map"
