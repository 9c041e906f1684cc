"""Worker for test_gpu_parity.py::test_three_ranks_one_card: several ranks share cuda:0 (the dev box has
one GPU; RCCL refuses two ranks on one device, so the halos travel over gloo, staged through host
memory).  Everything else is the product path: planner, pack kernel, interior/boundary HIP kernels,
fixed-tree dot.  Checks each rank's slice of A x, A^2 x, A^3 x bitwise against the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from navierstokes_amd import dist as D  # noqa: E402
from navierstokes_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)


def upper_part(p, c, v, row0):
    """Entries at or above the diagonal only (an upwind coupling): x then flows from higher ranks to lower ones, the LAST rank
    receives nothing and only sends — the rank the one-launch step's push gate exists for (push_exchange.hpp)."""
    rows = np.repeat(np.arange(len(p) - 1, dtype=np.int64) + row0, np.diff(p))
    keep = c >= rows
    cnt = np.zeros(len(p) - 1, np.int64)
    np.add.at(cnt, rows[keep] - row0, 1)
    return np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32), c[keep].copy(), v[keep].copy()


def main():
    kind, n, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    upwind = kind.endswith("_up")
    kind = kind[:-3] if upwind else kind
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if kind == "mesh":  # the P1 pressure operator on a Kuhn mesh of n cells per edge, natural node order: wide halos (a plane each side), and
        Pg, Cg, Vg = synth.pressure_matrix(n)  # interior pieces that take the cut-ring sliced stream where they are large enough
        n = len(Pg) - 1
    else:
        Pg, Cg, Vg = synth.rows(kind, n, w=w)
    if upwind:
        Pg, Cg, Vg = upper_part(Pg, Cg, Vg, 0)
    rs = D.balanced_row_starts(n, world, np.diff(Pg), align=4 if kind == "sfe" else 1)  # FE-like: cut at node boundaries
    lo, hi = int(rs[rank]), int(rs[rank + 1])
    if kind == "mesh":
        p, c, v = (Pg[lo:hi + 1] - Pg[lo]).astype(np.int32), Cg[Pg[lo]:Pg[hi]].copy(), Vg[Pg[lo]:Pg[hi]].copy()
    else:
        p, c, v = synth.rows(kind, n, lo, hi, w=w)
    if upwind:
        p, c, v = upper_part(p, c, v, lo)
    ok = True
    exchange = os.environ.get("MI355_TEST_EXCHANGE", "torch")
    trace = os.environ.get("MI355_TEST_TRACE") == "1"

    def mark(what):  # stage markers for a post-mortem of a stalled rank (MI355_TEST_TRACE=1)
        if trace:
            print(f"[rank {rank}] {what}", file=sys.stderr, flush=True)
    for kernel in (None, "ring", "stream", "sstream"):
        mark(f"kernel={kernel}: create")
        dc = D.DistCSR(rs, p, c, v, kernel=kernel, exchange=exchange)
        mark(f"kernel={kernel}: created push={dc.push} fused={dc.push_fused} n_halo={dc.n_halo}")
        if upwind:
            assert (dc.n_halo == 0) == (rank == world - 1) and (dc.n_send == 0) == (rank == 0), (rank, dc.n_halo, dc.n_send)
        assert not dc.native  # gloo: no RCCL
        expect = os.environ.get("MI355_TEST_EXPECT_FUSED")
        if expect and kernel is None:  # the test asked for a particular form of the one-launch step
            from navierstokes_amd import mpk
            got = mpk.lib().mi_part_kernel_name(dc._h, 2).decode()
            assert dc.push_fused and (got == expect[1:] if expect.startswith("=") else expect in got), (rank, got)
        assert dc.push == (exchange in ("push", "auto")), "peer-push exchange was requested but did not come up (or vice versa)"
        x_ext = dc.new_x_ext()
        x_ext[: dc.n_local] = torch.from_numpy(synth.x_sin(lo, hi)).cuda()
        ys = dc.spmk(x_ext, dc.new_power_buffers(3))
        torch.cuda.synchronize()
        mark("powers done")
        Y = O.spmk_chain(3, Pg, Cg, Vg, synth.x_sin(0, n))
        ok = ok and all(np.array_equal(ys[k].cpu().numpy().view(np.uint64), Y[k][lo:hi].view(np.uint64)) for k in range(3))
        if dc.push:  # many unsynchronised repetitions of one step into the same buffers: the windows' two parities at work
            y = dc.new_y()
            for _ in range(50):
                dc.spmv(x_ext, y)
            torch.cuda.synchronize()
            mark("50 repetitions done")
            dc.status()
            ok = ok and np.array_equal(y.cpu().numpy().view(np.uint64), Y[0][lo:hi].view(np.uint64))
            # the step protocol under skew: 36 steps that alternate between THREE different x vectors (so a ghost taken from the
            # wrong step or parity gives a wrong y), ranks stalling at different moments, no synchronisation until the end
            import random
            import time
            rnd = random.Random(1000 + rank)
            xg = [synth.x_sin(0, n), np.cos(0.002 * np.arange(n)), synth.x_sin(0, n) * 0.5 - 0.25]
            xs = []
            for g_ in xg:
                t_ = dc.new_x_ext()
                t_[: dc.n_local] = torch.from_numpy(g_[lo:hi]).cuda()
                xs.append(t_)
            outs = [dc.new_y() for _ in range(36)]
            for t in range(36):
                if rnd.random() < 0.3:
                    time.sleep(rnd.random() * 0.004)
                dc.spmv(xs[t % 3], outs[t])
            torch.cuda.synchronize()
            mark("skew steps done")
            dc.status()
            refs = [O.spmv(Pg, Cg, Vg, g_)[lo:hi] for g_ in xg]
            ok = ok and all(np.array_equal(outs[t].cpu().numpy().view(np.uint64), refs[t % 3].view(np.uint64)) for t in range(36))
        # coefficients replaced in place (a Newton loop's Jacobian): same plan, same exchange, new bits
        v2 = v * np.cos(np.arange(len(v)) + lo)
        dc.update_values(v2)
        y2 = dc.new_y()
        dc.spmv(x_ext, y2)
        torch.cuda.synchronize()
        full_v2 = Vg.copy()
        full_v2[Pg[lo]:Pg[hi]] = v2
        ok = ok and np.array_equal(y2.cpu().numpy().view(np.uint64), O.spmv(Pg, Cg, full_v2, synth.x_sin(0, n))[lo:hi].view(np.uint64))
        dc.update_values(v)
        mark("update_values done")
        g = float(dc.dot(ys[0], ys[0]))
        ref = float(np.dot(Y[0], Y[0]))
        ok = ok and abs(g - ref) <= 1e-12 * ref
        # distributed orthogonalize (mpk/SpMVmulti.cpp:146-151): global beta inside the reduction bound, and GIVEN that beta the owned
        # slice of x3 is the reference's fused update fma(-(alpha beta), b, x1) bit for bit (oracle: orc_ortho_update)
        x3 = torch.empty_like(ys[1])
        beta = float(dc.orthogonalize(ys[0], ys[1], x3, 1e-8))
        torch.cuda.synchronize()
        ref_beta, bound = float(np.dot(Y[0], Y[1])), float(np.dot(np.abs(Y[0]), np.abs(Y[1])))
        ok = ok and abs(beta - ref_beta) <= 1e-13 * bound
        ok = ok and np.array_equal(x3.cpu().numpy().view(np.uint64), O.ortho_update(1e-8 * beta, Y[0][lo:hi], Y[1][lo:hi]).view(np.uint64))
        mark("distributed orthogonalize done")
        dc.close()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"DIST_GPU_RESULT ok={int(flag)} world={world}")
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
