"""Worker for tests/test_dist_gloo.py: world_size ranks over gloo on the CPU.
Runs the product's DistCSR (planner + torch.distributed exchange) with the local
SpMV injected from the test oracle, and checks every rank's slice bitwise against
the global oracle SpMV."""
import os
os.environ["MI355_TEST_COMPUTE_HOOK"] = "1"  # this worker injects the oracle as local compute (CPU test seam)
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from navierstokes_amd import dist as D  # noqa: E402
from navierstokes_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def oracle_compute(dc, which, x_ext, y_local):
    if not hasattr(dc, "_pieces"):
        dc._pieces = [dc.local_piece(0), dc.local_piece(1)]
    p, c, v, rmap = dc._pieces[which]
    if len(rmap):
        y_local.numpy()[rmap] = O.spmv(p, c, v, x_ext.numpy())


def main():
    kind, n, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # every rank generates ONLY its own rows (counter-based generator)
    lens = None
    if kind == "svar":
        P, _, _ = synth.rows(kind, n, w=w)
        lens = np.diff(P)
    rs = D.balanced_row_starts(n, world, lens)
    lo, hi = int(rs[rank]), int(rs[rank + 1])
    p, c, v = synth.rows(kind, n, lo, hi, w=w)
    dc = D.DistCSR(rs, p, c, v, compute=oracle_compute, device="cpu")
    x_ext = dc.new_x_ext()
    x_ext[: dc.n_local] = torch.from_numpy(synth.x_sin(lo, hi))
    y = dc.new_y()
    y.fill_(float("nan"))
    dc.spmv(x_ext, y)
    # k = 3 chained distributed SpMVs (the k-exchange matrix-powers baseline)
    ys = [t.clone() for t in dc.spmk(x_ext, dc.new_power_buffers(3))]
    assert np.array_equal(ys[0].numpy().view(np.uint64), y.numpy().view(np.uint64))
    # global reference on every rank (small n)
    Pg, Cg, Vg = synth.rows(kind, n, w=w)
    Y = O.spmk_chain(3, Pg, Cg, Vg, synth.x_sin(0, n))
    ok = all(np.array_equal(ys[k].numpy().view(np.uint64), Y[k][lo:hi].view(np.uint64)) for k in range(3))
    g = dc.dot(ys[0], ys[0])
    ok_dot = abs(float(g) - float(np.dot(Y[0], Y[0]))) <= 1e-12 * float(np.dot(Y[0], Y[0]))
    # distributed orthogonalize (mpk/SpMVmulti.cpp:146-151): beta within the reduction bound, the update the reference's fma GIVEN beta
    b_loc, x1_loc = ys[0][: min(len(ys[0]), 2000)].clone(), ys[1][: min(len(ys[1]), 2000)].clone()
    x3_loc = torch.empty_like(x1_loc)
    beta = float(dc.orthogonalize(b_loc, x1_loc, x3_loc, 1e-8))
    from fractions import Fraction
    a = -(1e-8 * beta)
    want = np.array([float(Fraction(a) * Fraction(float(b_loc[i])) + Fraction(float(x1_loc[i]))) for i in range(len(x1_loc))])
    parts = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    mine = torch.tensor([float(torch.dot(b_loc, x1_loc)), float(torch.dot(b_loc.abs(), x1_loc.abs()))], dtype=torch.float64)
    dist.all_gather(parts, mine)
    ref_beta, bound = sum(float(t[0]) for t in parts), sum(float(t[1]) for t in parts)
    ok_ortho = np.array_equal(x3_loc.numpy().view(np.uint64), want.view(np.uint64)) and abs(beta - ref_beta) <= 1e-13 * bound
    flag = torch.tensor([1 if (ok and ok_dot and ok_ortho) else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"DIST_RESULT ok={int(flag)} world={world} halo={dc.n_halo} boundary={dc.n_boundary} interior={dc.n_interior}")
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
