#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference kernels.

Run in a container that has /root/reference and a built oracle/_ref/
(`make -C oracle`).  Every expected output below is produced by the
reference's own object code (mpk/SpMV.cpp, utils.cpp, SpM2V.cpp,
SpMVmulti0.cpp compiled where they lie); the inputs are seeded synthetic
matrices (navierstokes_amd/synth.py) or hand-made COO edge cases.  The .npz
files hold DATA only: inputs and expected outputs as raw float64/int32.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from navierstokes_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def csr_case(name, kind, n, w, xkind):
    p, c, v = synth.rows(kind, n, w=w)
    x = synth.x_sin(0, n) if xkind == "sin" else synth.x_ones(n)
    d = dict(kind=kind, n=n, w=w, xkind=xkind, ptrow=p, indcol=c, coef=v, x=x)
    # mpk/SpMV.cpp: the four CSR variants
    d["y_scalar"] = O.ref_spmv(p, c, v, x, "scalar")
    d["y_opt"] = O.ref_spmv(p, c, v, x, "opt")
    d["y_fma"] = O.ref_spmv(p, c, v, x, "fma")
    if np.all(np.diff(p) % 4 == 0):
        d["y_avx2"] = O.ref_spmv(p, c, v, x, "avx2")  # valid only for row lengths = 0 mod 4
    # mpk/SpM2V.cpp: first-touch table and fused 2-step
    d["end1"] = O.ref_gen_layer1(p, c)
    d["m2_y_opt"], d["m2_z_opt"] = O.ref_spm2v(p, c, v, x, "opt")
    d["m2_y_scalar"], d["m2_z_scalar"] = O.ref_spm2v(p, c, v, x, "scalar")
    # mpk/SpMVmulti0.cpp: k = 2,3,4 fused and the SpMV chain
    d["pow_chain4"] = O.ref_powers(4, p, c, v, x, fused=False)
    d["pow_fused2"] = O.ref_powers(2, p, c, v, x, fused=True)
    d["pow_fused3"] = O.ref_powers(3, p, c, v, x, fused=True)
    d["pow_fused4"] = O.ref_powers(4, p, c, v, x, fused=True)
    # mpk/SpMVmulti0.cpp:106-130, :157-187: the nested first-touch tables, flattened in traversal order
    lay = O.ref_gen_layers(p, c)
    assert np.array_equal(lay["e1"], d["end1"])
    for key in ("len2", "e2", "len3", "e3"):
        d["lay_" + key] = lay[key]
    # mpk/SpMVmulti-1.cpp:434-493: SpM4V_AVX2 (y1..y4)
    d["pow_avx2_4"] = O.ref_spm4v_avx2(p, c, v, x)
    # mpk/utils.cpp: the parity metric on a perturbed vector
    pert = d["y_scalar"] * (1.0 + 1e-9 * np.cos(np.arange(n)))
    d["pert"] = pert
    d["norm2_y"] = np.float64(O.ref("spmv").ref_norm2(n, d["y_scalar"]))
    d["rel_err_pert"] = np.float64(O.ref("spmv").ref_rel_error(n, d["y_scalar"], pert))
    np.savez(os.path.join(OUT, name + ".npz"), **d)
    print(name, "n", n, "nnz", len(c))


def coo_case(name, nrow, irow, jcol, val, x):
    d = dict(nrow=nrow, irow=irow, jcol=jcol, val=val, x=x)
    p, c, v = O.ref_coo2csr(nrow, irow, jcol, val)
    d["csr_ptrow"], d["csr_indcol"], d["csr_coef"] = p, c, v
    d["y_scalar"] = O.ref_spmv(p, c, v, x, "scalar")
    d["y_fma"] = O.ref_spmv(p, c, v, x, "fma")
    bp, bc, bv = O.ref_coo2bcsr4(nrow, irow, jcol, val)
    d["bcsr_ptrow"], d["bcsr_indcol"], d["bcsr_coef"] = bp, bc, bv
    d["end1"] = O.ref_gen_layer1(p, c)                      # mpk/SpM2V.cpp:5-26 on the duplicate-dropped CSR
    if nrow % 4 == 0:                                        # block columns must be block rows too
        d["bcsr_end1"] = O.ref_gen_layer1_bcsr4(bp, bc)     # mpk/SpM2V.cpp:28-46
    xb = x[: 4 * (nrow // 4)] if nrow % 4 else x
    # BCSR kernels index x by block column: pad x so that a block column that
    # straddles nrow stays in bounds (the reference would read past the vector)
    xpad = np.concatenate([x, np.zeros(4)])
    for var in ("scalar", "opt", "fma", "avx2"):
        d["yb_" + var] = O.ref_spmv_bcsr(bp, bc, bv, xpad, var)
    if nrow % 4 == 0:
        for var in ("opt", "fma", "avx2"):
            yb, zb = O.ref_spm2v_bcsr(bp, bc, bv, xpad, var)
            d["m2b_y_" + var], d["m2b_z_" + var] = yb, zb
    del xb
    np.savez(os.path.join(OUT, name + ".npz"), **d)
    print(name, "nrow", nrow, "coo", len(irow), "csr", len(c), "blocks", len(bc))


def blas1_case(name, n, m, alpha, seed):
    """dot + AXPY helpers between the SpMVs: orthogonalize (both forms) and orthonormalize_against_basis."""
    rng = np.random.default_rng(seed)
    b = rng.uniform(-1, 1, n)
    x1 = rng.uniform(-1, 1, n)
    basis = np.sin(0.001 * np.arange(n)[None, :] + np.arange(m)[:, None])  # v_i[j] = sin(0.001 j + i), mpk/2SpMV.cpp:110-116
    d = dict(n=n, m=m, alpha=alpha, b=b, x1=x1, basis=basis)
    d["x3_ortho3"] = O.ref_orthogonalize3(b, x1, alpha)              # mpk/old/SpMVmulti.cpp:164-169 (= mpk/SpMVmulti.cpp:146-151)
    d["y_ortho_inplace"] = O.ref_orthogonalize_inplace(b, x1, alpha)  # mpk/2SpMV.cpp:3-11
    d["y_mgs"] = O.ref_mgs(basis, x1)                               # mpk/2SpMV.cpp:13-28
    np.savez(os.path.join(OUT, name + ".npz"), **d)
    print(name, "n", n, "m", m)


def main():
    if not O.have_ref():
        sys.exit("oracle/_ref is not built: run `make -C oracle` where /root/reference exists")
    csr_case("s15_n512", "s15", 512, 40, "sin")
    csr_case("svar_n400", "svar", 400, 30, "sin")
    csr_case("sfe_n268", "sfe", 268, 40, "ones")  # 268 rows = mat/matrix1 (mpk/log/log_SPMV.txt:1)

    blas1_case("blas1_n1003", 1003, 6, 1e-8, 7)    # odd length: exercises the vectorised dot's pair + single tails
    blas1_case("blas1_n2000", 2000, 50, 0.25, 8)   # 50 basis vectors as the reference's harness builds; alpha large enough to matter

    # COO edge cases: duplicates (first-wins in CSR, last-wins in BCSR4), empty
    # rows, missing diagonal, unsorted input, nrow not a multiple of 4
    rng = np.random.default_rng(20250824)
    nrow = 37
    m = 300
    irow = rng.integers(0, nrow, m).astype(np.int32)
    jcol = rng.integers(0, nrow, m).astype(np.int32)
    keep = (irow != 5) & (irow != 17) & (irow != 36)  # rows 5, 17, 36 empty
    irow, jcol = irow[keep], jcol[keep]
    dup = rng.integers(0, len(irow), 40)
    irow = np.concatenate([irow, irow[dup]])
    jcol = np.concatenate([jcol, jcol[dup]])
    val = rng.uniform(-1, 1, len(irow))
    x = rng.uniform(-1, 1, nrow)
    coo_case("edge_coo_n37", nrow, irow, jcol, val, x)

    nrow = 40
    irow = rng.integers(0, nrow, 500).astype(np.int32)
    jcol = rng.integers(0, nrow, 500).astype(np.int32)
    val = rng.uniform(-1, 1, 500)
    x = rng.uniform(-1, 1, nrow)
    coo_case("edge_coo_n40", nrow, irow, jcol, val, x)


if __name__ == "__main__":
    main()
