"""Worker for tests/test_dist_single_process.py (TEST INFRASTRUCTURE): one process, N ranks behind ONE mi_dist handle.
Run in a child process because the shim reads MI355_NGPUS once and the library reads MI355_RCCL_LIBRARY at its first use.
  capi <N>          navierstokes_amd.mpk.DistMatrix (ctypes over include/mi355_spmv.h): products, powers, BLAS-1, device vectors
  shim <N>          the C++ symbols of include/SpMV.h through tests/shim_harness with MI355_NGPUS=N, against the reference-made goldens
Prints DIST_SINGLE_OK on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import assert_bit_equal  # noqa: E402
from navierstokes_amd import mpk, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)


def upwind(n, w):
    """rows that reach only DOWN the numbering: the last rank sends and receives nothing from above (asymmetric couplings)"""
    p, c, v = synth.rows("s15", n, w=w)
    keep = c <= np.repeat(np.arange(n), np.diff(p))
    rows = np.repeat(np.arange(n), np.diff(p))[keep]
    p2 = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    return p2, c[keep].copy(), v[keep].copy()


def capi(N):
    want = os.environ.get("MI355_DIST_EXCHANGE", "auto")
    cases = [("s15", 200_000, 2000), ("svar", 120_000, 900), ("sfe", 60_000, 1500), ("up", 90_000, 700), ("s15", 3 * N + 1, 2)]
    for kind, n, w in cases:
        if kind == "up":
            p, c, v = upwind(n, w)
        else:
            p, c, v = synth.rows(kind, n, w=w)
        n = len(p) - 1
        D = mpk.DistMatrix(N, n, p, c, v)
        info = D.info()
        assert info["nranks"] == N and sum(r["n_local"] for r in info["ranks"]) == n, info
        if want != "auto":
            assert info["exchange"].split("-")[0] == want, info
        x = synth.x_sin(0, n)
        y = np.full(n, np.nan)
        D.spmv(y, x)
        assert_bit_equal(y, O.spmv(p, c, v, x), f"{kind} N={N} {info['exchange']}: y = A x")
        # powers, host vectors: SpM4V's outputs (mpk/SpMVmulti0.cpp:189-221)
        Y = O.spmk_chain(4, p, c, v, x)
        outs = [np.full(n, np.nan) for _ in range(4)]
        D.spmk(outs, x)
        for q in range(4):
            assert_bit_equal(outs[q], Y[q], f"{kind} N={N}: power {q + 1}")
        # device-resident vectors: 40 back-to-back steps into the same vector (hazards between steps), then a powers chain
        vx, vy = D.vector(x), D.vector()
        for _ in range(40):
            D.spmv_dev(vy, vx)
        D.synchronize()
        assert_bit_equal(vy.get(), Y[0], f"{kind} N={N}: 40 steps back to back")
        vp = [D.vector() for _ in range(3)]
        for _ in range(5):
            D.spmk_dev(vp, vx)
        D.synchronize()
        for q in range(3):
            assert_bit_equal(vp[q].get(), Y[q], f"{kind} N={N}: device powers {q + 1}")
        # BLAS-1 across the ranks: dot inside the reduction bound, the update bit-equal GIVEN beta
        b = np.cos(0.003 * np.arange(n))
        beta = D.dot(b, Y[0])
        scale = np.abs(b * Y[0]).sum()
        assert abs(beta - O.dot(b, Y[0])) <= 1e-13 * scale + 1e-300
        x3 = np.full(n, np.nan)
        beta2 = D.orthogonalize(b, Y[0], x3, 1e-8)
        assert beta2 == beta
        assert_bit_equal(x3, O.ortho_update(1e-8 * beta, b, Y[0]), "orthogonalize: update given beta")
        vb, v3 = D.vector(b), D.vector()
        beta3 = D.orthogonalize_dev(vb, vy, v3, 1e-8)
        assert beta3 == beta
        assert_bit_equal(v3.get(), x3, "orthogonalize_dev")
        # (round 5) the update has no host in its middle: seven calls back to back on alternating vector pairs (the shared table of partial
        # dots has two parities; every rank's update kernel adds the partials in rank order itself), one synchronisation at the end
        b2 = np.sin(0.007 * np.arange(n)) + 0.25
        vb2, v4 = D.vector(b2), D.vector()
        betas = []
        for t in range(7):
            betas.append(D.orthogonalize_dev(vb if t % 2 == 0 else vb2, vy, v3 if t % 2 == 0 else v4, 1e-8, want_beta=t >= 5))
        D.synchronize()
        beta_b2 = D.dot(b2, Y[0])
        assert betas[:5] == [None] * 5 and betas[5] == beta_b2 and betas[6] == beta, (betas, beta, beta_b2)
        assert_bit_equal(v3.get(), x3, "orthogonalize_dev, repeated")
        assert_bit_equal(v4.get(), O.ortho_update(1e-8 * beta_b2, b2, Y[0]), "orthogonalize_dev, second pair")
        vb2.close()
        v4.close()
        # new coefficients, same pattern
        v2 = v * np.cos(np.arange(len(v)))
        D.update_values(v2)
        D.spmv(y, x)
        assert_bit_equal(y, O.spmv(p, c, v2, x), f"{kind} N={N}: after mi_dist_update_values")
        for t in [vx, vy, vb, v3] + vp:
            t.close()
        D.close()
        print(f"  capi {kind} n={n} N={N}: exchange={info['exchange']} devices={info['distinct_devices']} halo_max={info['halo_max']} ok", flush=True)


def shim_mode(N):
    assert os.environ.get("MI355_NGPUS") == str(N)
    import shim
    G = os.path.join(ROOT, "tests", "golden")
    for name in ("s15_n512", "svar_n400", "sfe_n268"):
        g = dict(np.load(os.path.join(G, name + ".npz")))
        p, c, v, x = g["ptrow"], g["indcol"], g["coef"], g["x"]
        for fn in shim.CSR_VARIANTS:  # SpMV_CSR, _OPT, _FMA, _AVX2, SpMV
            assert_bit_equal(shim.spmv_csr(fn, p, c, v, x), g["y_fma"], f"{fn} over {N} ranks vs reference SpMV_CSR_FMA")
        for fn in shim.SPM2V_VARIANTS:
            y, z = shim.spm2v_csr(fn, p, c, v, x)
            assert_bit_equal(y, g["m2_y_opt"], fn)
            assert_bit_equal(z, g["m2_z_opt"], fn)
        assert_bit_equal(shim.powers("SpM3V", p, c, v, x), g["pow_fused3"], "SpM3V")
        Y4 = shim.powers("SpM4V", p, c, v, x)
        assert_bit_equal(Y4[:3], g["pow_fused3"], "SpM4V")
        assert_bit_equal(Y4, shim.powers("SpM4V_AVX2", p, c, v, x), "SpM4V_AVX2")
        for k in range(4):
            assert O.rel_error(g["pow_fused4"][k], Y4[k]) <= 1e-15
        # orthogonalize(nrow, b, x1, x3, alpha) behind the product: distributed like the matrix last multiplied with
        n = len(p) - 1
        b = np.cos(0.01 * np.arange(n))
        x3 = shim.orthogonalize3(b, Y4[0], 1e-8)
        yi = shim.orthogonalize_inplace(b, Y4[0], 1e-8)
        assert_bit_equal(x3, yi)
        scale = np.abs(b * Y4[0]).sum()
        ref = O.ortho_update(1e-8 * O.dot_gccvec(b, Y4[0]), b, Y4[0])
        assert np.abs(x3 - ref).max() <= 1e-8 * 1e-13 * scale * np.abs(b).max() + 2e-16 * np.abs(Y4[0]).max()
        print(f"  shim {name} N={N} ok", flush=True)
    # the stale-copy hazard through the distributed handle: an in-place edit of ONE coefficient must be seen
    n = 20000
    p, c, v = synth.rows("s15", n, w=300)
    x = synth.x_sin(0, n)
    idx = [len(c) // 2 + 1]
    y0, y1 = shim.spmv_csr_inplace_edit(p, c, v, x, idx, [0.375])
    v2 = v.copy()
    v2[idx] = 0.375
    assert_bit_equal(y0, O.spmv(p, c, v, x))
    assert_bit_equal(y1, O.spmv(p, c, v2, x), "in-place edit seen by the distributed handle")


if __name__ == "__main__":
    mode, N = sys.argv[1], int(sys.argv[2])
    {"capi": capi, "shim": shim_mode}[mode](N)
    print("DIST_SINGLE_OK", flush=True)
    sys.stdout.flush()
    os._exit(0)
