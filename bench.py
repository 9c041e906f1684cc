#!/usr/bin/env python3
"""bench.py — fp64 CSR SpMV throughput of the HIP path on 1..N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c3] [--kernel ...]

With --gpus N > 1 and no torchrun environment (RANK unset) the script starts its N ranks ITSELF, as a child process
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, decided before anything touches the GPU), relays the
child's output and exits with its code; started by torch.distributed.run it is one of the ranks.

A "step" is one pass of the hot path over the whole (global) matrix: y = A x
(workloads c2, c4) or the k=4 matrix-powers chain y1..y4 (workload c3).  The
matrix is synthetic (no matrix ships with the reference):
generator S15 of SURVEY.md §8d, seed 0x5EED — c4 = 5 M rows / 75 M nnz (the
configuration BASELINE.json's target is quoted on; it fits one GPU), c2 = 1 M
rows / 15 M nnz.  For N > 1 (one rank per GPU) the SAME global matrix is
row-partitioned by nnz, each rank generates only its rows, and every step
exchanges the packed halo x entries while the interior rows compute — strong
scaling.  Which exchange drives the step is MEASURED: each candidate (peer push
over HIP IPC windows, the C++ step over RCCL send/recv, torch.distributed
all_to_all) is brought up, checked bit for bit against the torch.distributed
exchange, dry-run and timed on every rank; the fastest one that survived runs
the timed region and the line reports all of them (`halo.exchanges`), the bytes
exchanged per step and how much of the step is compute (`halo.overlap`).

Rank 0 prints ONE JSON line.  `value` = 2*nnz*k*K / wall (GFLOP/s, the
reference's convention, src/benchmark_spmv.c:234), inputs resident in HBM.
`roofline.achieved` = ALGORITHMIC bytes per launch (12*nnz + 4*(n+1) + 16*n,
SURVEY.md §8d) / mean launch duration from HIP events recorded on the launch
stream around the timed region.  `cpu_baseline` = the reference's own SpMV_CSR
object code (oracle/_ref, kind "reference") or, where that is absent, the C
restatement (kind "port"), one thread, cold cache, on this host — rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

import ctypes as _ct

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy)
WORKLOADS = {
    "c4": dict(kind="s15", n=5_000_000, k=1, desc="S15 synthetic CSR 5,000,000 rows x 15 nnz/row = 75,000,000 nnz, y=Ax"),
    "c2": dict(kind="s15", n=1_000_000, k=1, desc="S15 synthetic CSR 1,000,000 rows x 15 nnz/row = 15,000,000 nnz, y=Ax"),
    "c3": dict(kind="s15", n=1_000_000, k=4, desc="S15 synthetic CSR 1,000,000 rows x 15 nnz/row, k=4 matrix powers y1..y4"),
    "tiny": dict(kind="s15", n=200_000, k=1, desc="S15 synthetic CSR 200,000 rows x 15 nnz/row (plumbing checks only)"),
    # the reference's own matrix family: NS FE matrix, 4 dofs/node, <=60 nnz/row, 68^3 cells
    "fe": dict(kind="fe", n=4 * 69 ** 3, k=1, cells=68, desc="NS P1-P1 FE matrix (src/integration.c + benchmark_spmv.c block rule) on a 68^3-cell Kuhn box: 1,314,036 rows, CSR, y=Ax"),
    # the same FE matrix under a seeded RANDOM node numbering (4 dofs of a node kept together) — what an unstructured gmsh
    # mesh delivers (src/solve_newton.c:91-197): mi_csr_create relabels it behind the API (reorder.hpp), bits unchanged
    "fe_perm": dict(kind="fe", n=4 * 69 ** 3, k=1, cells=68, perm_block=4, desc="the fe matrix under a random node numbering (unstructured-mesh order), CSR, y=Ax"),
    # the reference's SCALAR operator shape (pressure Poisson): the pressure-pressure part of the FE matrix, a P1 Laplacian
    # on the jittered Kuhn mesh, 15 nonzeros per row like S15 but with a mesh's column sharing and a column span of two
    # mesh planes (58 k columns here) — no contiguous LDS window holds that: the tile kernel's case (spmv_tile.hpp)
    "mesh": dict(kind="mesh", n=171 ** 3, k=1, cells=170, desc="P1 pressure operator (stabilisation block of src/integration.c) on a 170^3-cell Kuhn mesh: 5,000,211 rows, 72.9 M nnz, natural node order, CSR, y=Ax"),
    "mesh_perm": dict(kind="mesh", n=171 ** 3, k=1, cells=170, perm_block=1, desc="the mesh matrix under a random node numbering (unstructured-mesh order), CSR, y=Ax"),
    "mesh_small": dict(kind="mesh", n=101 ** 3, k=1, cells=100, desc="P1 pressure operator on a 100^3-cell Kuhn mesh: 1,030,301 rows, natural node order, CSR, y=Ax"),
    "mesh_small_perm": dict(kind="mesh", n=101 ** 3, k=1, cells=100, perm_block=1, desc="the 100^3-cell mesh matrix under a random node numbering, CSR, y=Ax"),
    "c2_perm": dict(kind="s15", n=1_000_000, k=1, perm_block=1, desc="the c2 matrix under a random row/column numbering, CSR, y=Ax"),
    # multi-vector product, s = 4 columns, matrix read once (MatMatMult_SeqBAIJ_4_AVX2, src/kernels/spmm_avx2.c:7-109)
    "fe_spmm4": dict(kind="fe", n=4 * 69 ** 3, k=1, cells=68, bcsr=True, spmm=4, desc="the FE matrix as BCSR 4x4, Y = A X for 4 column vectors in one launch (spmm_avx2.c)"),
    "fe_spmm8": dict(kind="fe", n=4 * 69 ** 3, k=1, cells=68, bcsr=True, spmm=8, desc="the FE matrix as BCSR 4x4, Y = A X for 8 column vectors in one launch (spmm_avx2.c)"),
    "fe_bcsr": dict(kind="fe", n=4 * 69 ** 3, k=1, cells=68, bcsr=True, desc="same FE matrix as BCSR 4x4 (SpMV_BCSR path, mpk/SpMV.cpp:90-219), y=Ax"),
}


def algorithmic_bytes(n, nnz):
    """values(8)+colidx(4) per nnz, rowptr, x read once, y written once (src/benchmark_spmv.c:197-202)."""
    return 12 * nnz + 4 * (n + 1) + 8 * n + 8 * n


def load_traffic(workload, kernel_name):
    """HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for this very
    workload and kernel (committed under profiles/*_pmc.json with the calibration used), or None.
    Counters cannot be read from inside the timed run, so this is the last profiled value."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_pmc.json"):
                try:
                    d = json.load(open(os.path.join(pdir, f)))
                except Exception:
                    continue
                # the temporal / non-temporal instantiations of one kernel move the same bytes (measured:
                # 408.9 vs 410.4 thousand KB fetched), so a profile of either describes both
                # (round 3 added an eleventh template argument — the dot epilogue, false in a plain product — to the ring kernel)
                norm = lambda a: a.replace(", false>", ">") if a.startswith("spmv_csr_ring<") and a.count(",") == 10 else a
                same = lambda a, b: norm(a) == norm(b) or (a.startswith("spmv_csr_ring<") and norm(a).split(",")[:5] == norm(b).split(",")[:5]
                                                           and norm(a).split(",")[7:] == norm(b).split(",")[7:])
                # (the variants of the sliced kernels — prefetch depth, temporal / non-temporal values, park form — read the same sliced copy:
                # a profile of one instantiation describes the others)
                family = lambda a: next((pre for pre in ("spmv_sstream<", "spmv_bcsr4_sell<") if a.startswith(pre)), None)
                fam_same = family(kernel_name) is not None and family(kernel_name) == family(d.get("kernel", ""))
                if d.get("workload") == workload and (same(d.get("kernel", ""), kernel_name) or fam_same):
                    best = (d.get("hbm_bytes_per_launch"), "profiles/" + f)
    return best if best else (None, None)


def cpu_baseline(p, c, v, x, seconds_budget=20.0):
    """Single-thread CPU SpMV on this host, cold cache, best of a few calls (the reference's
    protocol: flush_cache() then one timed call, mpk/SpM2V.cpp:895-904)."""
    from oracle import oracle as O  # the checker/baseline, never the measured product

    n, nnz = len(p) - 1, len(c)
    out = {"unit": "GFLOP/s", "cores": 1}
    reps = 3
    if O.have_ref():
        t, _ = O.ref_time_spmv(p, c, v, x, "scalar", reps=reps, flush=True)
        out.update(kind="reference", value=2 * nnz / t / 1e9,
                   sample=f"full workload ({n} rows, {nnz} nnz), reference SpMV_CSR (x87 scalar, mpk/SpMV.cpp:6-20) "
                          f"object code, 1 thread, best of {reps} cold calls after flush_cache()")
        extra = {}
        for var in ("opt", "fma"):
            tv, _ = O.ref_time_spmv(p, c, v, x, var, reps=2, flush=True)
            extra[f"SpMV_CSR_{var.upper()}"] = round(2 * nnz / tv / 1e9, 4)
        # the same arithmetic written as a plain C loop (oracle/cpu_ref.c): what one core of this host can do
        tp, _ = O.time_spmv(p, c, v, x, reps=2, flush=True)
        extra["oracle_fma_chain_port_1_thread"] = round(2 * nnz / tp / 1e9, 4)
        out["other_variants_gflops"] = extra
    else:
        t, _ = O.time_spmv(p, c, v, x, reps=reps, flush=True)
        out.update(kind="port", value=2 * nnz / t / 1e9,
                   sample=f"full workload ({n} rows, {nnz} nnz), oracle/cpu_ref.c fma chain, 1 thread, "
                          f"best of {reps} cold calls after a 300 MiB flush")
    out["value"] = round(out["value"], 4)
    # all host cores this process may use (SURVEY.md §8d "CPU baseline" (2)): the same fma chain, rows split
    # over OpenMP threads (oracle/cpu_ref.c: orc_spmv_csr_fma_omp), warm, best of 5.  Not something the
    # reference ships (mpk/ is single-threaded): the fair upper bound for its algorithm on this host.
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    nthreads = int(os.environ.get("MI355_BENCH_CPU_THREADS", "0")) or min(avail, 64)
    best = None
    for nt in sorted({nthreads, max(1, nthreads // 2), min(nthreads, 16)}):
        ta, _ = O.time_spmv_omp(p, c, v, x, nt, reps=5)
        if best is None or ta < best[0]:
            best = (ta, nt)
    out["all_cores"] = dict(value=round(2 * nnz / best[0] / 1e9, 3), unit="GFLOP/s", cores=best[1], kind="port",
                            sample=f"full workload, oracle fma chain row-parallel over OpenMP threads (best of "
                                   f"{sorted({nthreads, max(1, nthreads // 2), min(nthreads, 16)})} threads; {avail} usable by this process), warm, best of 5")
    try:
        out["host_cpu"] = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
        out["host_cores_available"] = os.cpu_count()
    except Exception:
        pass
    return out


def self_launch(ngpus):
    """`bench.py --gpus N` started plainly: run the N ranks as a CHILD process (torch.distributed.run on 127.0.0.1, a free
    port), let it write to our stdout/stderr, return its exit code.  Nothing here touches the GPU (no torch import): a
    process that has initialised the GPU must neither exec nor be re-launched on this pool."""
    import signal
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL, the push windows)
    env.setdefault("MI355_BENCH_T0", repr(time.time()))  # the ranks' progress lines and the single-process budget count from HERE
    limit = int(os.environ.get("MI355_BENCH_LAUNCH_TIMEOUT", "1500"))
    # the child's stdout is filtered: the ONE JSON line goes to our stdout, anything else a rank or a backend prints there
    # (gloo's connection banner, for one) goes to stderr — the contract is one JSON line on stdout
    import threading
    child = subprocess.Popen(cmd, env=env, start_new_session=True, stdout=subprocess.PIPE, text=True, bufsize=1)

    def relay():
        for ln in child.stdout:
            looks_json = ln.lstrip().startswith("{") and '"metric"' in ln
            (sys.stdout if looks_json else sys.stderr).write(ln)
            (sys.stdout if looks_json else sys.stderr).flush()
    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    try:
        rc = child.wait(timeout=limit)
        pump.join(timeout=10)
        return rc
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {ngpus}-rank child did not finish within {limit} s; stopping its process group", file=sys.stderr, flush=True)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(child.pid, sig)  # exactly the group started above
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=15)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124
    except KeyboardInterrupt:
        os.killpg(child.pid, signal.SIGTERM)
        return 130


def single_process_main(args):
    """--single-process: ONE process drives the N GPUs through a mi_dist handle (include/mi355_spmv.h; what the reference's
    single-process harnesses reach through the mpk/SpMV.h shim with MI355_NGPUS=N).  Same workload, same metric, same JSON line;
    the timed region is K steps enqueued back to back on device-resident distributed vectors between two mi_dist_synchronize()."""
    import torch  # (device count only: this process never makes torch's context current on the GPUs the library drives)
    from navierstokes_amd import mpk, synth

    W = WORKLOADS[args.workload]
    n, k, kind = W["n"], W["k"], W["kind"]
    if kind not in ("s15", "fe") or W.get("bcsr") or W.get("perm_block"):
        sys.exit("--single-process runs the c2 / c3 / c4 / tiny / fe workloads")
    N = args.gpus
    t_setup = time.perf_counter()
    p, c, v = synth.fe_matrix(W["cells"]) if kind == "fe" else synth.rows(kind, n)
    nnz = len(c)
    x_host = synth.x_sin(0, n)
    D = mpk.DistMatrix(N, n, p, c, v)
    info = D.info()
    vx = D.vector(x_host)
    outs = [D.vector() for _ in range(k)]
    if k == 1:
        def step():
            D.spmv_dev(outs[0], vx)
    else:
        def step():
            D.spmk_dev(outs, vx)
    D.synchronize()
    t_setup = time.perf_counter() - t_setup
    for _ in range(args.warmup):
        step()
    D.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    D.synchronize()
    wall = time.perf_counter() - t0
    parity = None
    if not args.no_parity:
        from oracle import oracle as O  # checker only
        Y = O.spmk_chain(k, p, c, v, x_host)
        got = [t.get() for t in outs]
        parity = dict(rel_error=max(O.rel_error(Y[i], got[i]) for i in range(k)),
                      bitwise=all(np.array_equal(Y[i].view(np.uint64), got[i].view(np.uint64)) for i in range(k)),
                      against="oracle/cpu_ref.c fma chain (= reference SpMV_CSR_OPT/_FMA), full vectors gathered from the ranks")
    value = 2.0 * nnz * k * args.steps / wall / 1e9
    B_rank = max(algorithmic_bytes(r["n_local"], r["nnz_local"]) for r in info["ranks"])
    launch_s = wall / (args.steps * k)
    achieved = B_rank / launch_s / 1e9
    out = dict(metric="fp64 CSR SpMV GFLOP/s & % HBM roofline @ nnz; 1/2/4/8 GPU", value=round(value, 2), unit="GFLOP/s", n_gpus=N, steps=args.steps,
               warmup=args.warmup, ms_per_step=round(wall * 1e3 / args.steps, 5), higher_is_better=True, scaling="strong", vs_baseline=None,
               dtype="f64", data="synthetic",
               config=dict(workload=W["desc"], name=args.workload, n=n, nnz=nnz, k=k, seed="0x5EED", half_bandwidth=synth.DEFAULT_W,
                           partition=f"row-range x{N}, ONE process (mi_dist: a worker thread per rank)", cold=False, numbering="caller's"),
               roofline=dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None,
                             algorithmic_bytes_per_launch=B_rank, launch_us=round(launch_s * 1e6, 2),
                             bytes_model="CSR: 12 B per nonzero + 4 B per row pointer + 16 B per row (SURVEY.md §8d), the largest rank's share",
                             timing="host clock between two mi_dist_synchronize() around the K steps / launches (the ranks' streams live on different devices)"),
               pct_hbm_roofline=round(100 * achieved / HBM_PEAK_GBS, 2),
               halo=dict(driver="mi_dist (one process, N devices)", exchange=info["exchange"], fused_one_launch_per_rank=info["fused"], exchange_note=info["note"],
                         distinct_devices=info["distinct_devices"], devices_visible=torch.cuda.device_count(), halo_total=info["halo_total"], halo_max=info["halo_max"],
                         ranks=[dict(device=r["device"], rows=r["n_local"], halo=r["n_halo"]) for r in info["ranks"]]),
               setup_s=round(t_setup, 2))
    if parity is not None:
        out["parity"] = parity
    print(json.dumps(out), flush=True)
    for t in [vx] + outs:
        t.close()
    D.close()


def run_single_process_child(args, timeout_s=240):
    """N > 1, rank 0, after the timed region of the process-per-GPU path: the same workload through ONE process (a child of its own,
    so that a failure or a hang there costs this line nothing but the field).  Returns the child's JSON line as a dict, or a note."""
    import subprocess
    env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                  "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--single-process", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--workload", args.workload]
    if "MI355_FORCE_DEVICE" in os.environ:  # development: every rank on one card
        env["MI355_DIST_DEVICES"] = os.environ["MI355_FORCE_DEVICE"]
    try:
        # (in the caller's process group: self_launch's killpg on a timeout reaches this child too — a session of its own outlived a killed run
        # while holding all N GPUs; subprocess.run kills the child itself when its own timeout expires)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return dict(ok=False, note=f"child did not finish in {timeout_s} s")
    for line in reversed(r.stdout.splitlines()):
        if line.startswith("{"):
            try:
                d = json.loads(line)
                return dict(ok=True, value=d["value"], ms_per_step=d["ms_per_step"], frac=d["roofline"]["frac"], halo=d["halo"], parity=d.get("parity"), setup_s=d.get("setup_s"))
            except Exception:  # noqa: BLE001
                break
    return dict(ok=False, note=f"child exited {r.returncode}: {(r.stderr or r.stdout)[-300:]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "stream", "ring", "rowpar", "bcsr4", "tile", "mring", "sstream"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--exchange", default=None, choices=["auto", "native", "allgather", "push", "torch"], help="N > 1: which halo exchange drives the step")
    ap.add_argument("--cold", action="store_true", help="evict L2/Infinity Cache before every timed step")
    ap.add_argument("--no-extras", action="store_true", help="skip the cold single-shot and in-pipeline measurements")
    ap.add_argument("--plain-vectors", action="store_true", help="(the default since round 4; accepted for old command lines) x and y of the timed region are plain torch allocations")
    ap.add_argument("--placed-vectors", action="store_true", help="N = 1, y = A x on a matrix of >= 20 M nonzeros: ALSO let the library place a pair of vectors "
                    "(mi_vec_alloc_placed: candidate pairs timed, fastest kept — profiles/NOTES.md, placement) and report its rate in kernel_info.vectors; "
                    "the timed region, `value` and `roofline` stay on the plain allocations")
    ap.add_argument("--internal", action="store_true", help="N = 1, y = A x CSR workloads: x and y stay in the library's numbering "
                    "(mi_spmv_internal_dev: a relabelled matrix pays no gather and no mapped store per product; what a Krylov loop does)")
    ap.add_argument("--single-process", action="store_true", help="drive the N GPUs from ONE process through a mi_dist handle (what the reference's "
                    "single-process harnesses reach through the shim with MI355_NGPUS=N) instead of one process per GPU")
    ap.add_argument("--no-single-process-extra", action="store_true", help="N > 1: do not also time the one-process form (a child process of rank 0, "
                    "after the timed region; reported as `single_process`, never as `value`)")
    args = ap.parse_args()
    if args.single_process:
        return single_process_main(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    from navierstokes_amd import dist as D
    from navierstokes_amd import mpk, synth

    t_wall0 = time.time()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world  # under torch.distributed.run the environment says how many ranks there are
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if "MI355_FORCE_DEVICE" in os.environ:  # development only: several ranks on one card
        local_rank = int(os.environ["MI355_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    mpk.lib()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MI355_BENCH_BACKEND", "nccl")  # "gloo": development runs only
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("MI355_BENCH_COLLECTIVE_TIMEOUT", "300")))  # a stuck rank fails the run, not the day
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)

    W = WORKLOADS[args.workload]
    n, k, kind = W["n"], W["k"], W["kind"]
    is_fe = kind == "fe"
    is_mesh = kind == "mesh"
    if is_mesh and world > 1:
        sys.exit("the mesh workloads are 1-GPU configurations")
    if is_fe and world > 1 and (W.get("bcsr") or W.get("spmm")):
        sys.exit("the BCSR-API and multi-vector FE workloads are 1-GPU configurations")
    nnz_global = (int(synth._lib().fe_matrix_count(W["cells"], W["cells"], W["cells"])) // (16 if is_mesh else 1) if is_fe or is_mesh else
                  int(synth._lib().synth_count(synth.KINDS[kind], synth.DEFAULT_SEED, n, synth.DEFAULT_W, 0, n)))

    # ---- build this rank's share -------------------------------------------------------
    t_setup = time.perf_counter()
    if is_fe and world > 1:
        # every rank assembles the whole FE matrix (1 GB on the host) and keeps its rows; cuts at node boundaries, nnz-balanced
        P_, C_, V_ = synth.fe_matrix(W["cells"])
        rs = D.balanced_row_starts(n, world, np.diff(P_), align=4)
        lo, hi = int(rs[rank]), int(rs[rank + 1])
        p, c, v = (P_[lo:hi + 1] - P_[lo]).astype(np.int32), C_[P_[lo]:P_[hi]].copy(), V_[P_[lo]:P_[hi]].copy()
        del P_, C_, V_
    else:
        rs = D.balanced_row_starts(n, world)  # S15 rows all hold 15 nnz: equal rows == equal nnz
        lo, hi = int(rs[rank]), int(rs[rank + 1])
        p, c, v = (synth.fe_matrix(W["cells"]) if is_fe else synth.pressure_matrix(W["cells"]) if is_mesh
                   else synth.rows(kind, n, lo, hi))
    if W.get("perm_block"):
        if world > 1:
            sys.exit("the permuted workloads are 1-GPU configurations")
        p, c, v, _ = synth.permute_nodes(p, c, v, block=W["perm_block"], seed=synth.DEFAULT_SEED)
    x_host = synth.x_sin(lo, hi)
    vector_info = None
    bcsr = bool(W.get("bcsr"))
    if world == 1 and bcsr:
        bp, bc, bv = synth.csr_to_bcsr4(p, c, v)
        A = mpk.bcsr4x4_matrix(n // 4, bp, bc, bv, nbcols=n // 4)
        _ = A.handle
        bt = [_ct.c_int(), _ct.c_int(), _ct.c_double(), _ct.c_double()]
        mpk.check(mpk.lib().mi_bcsr4_tile_info(A.handle, _ct.byref(bt[0]), _ct.byref(bt[1]), _ct.byref(bt[2]), _ct.byref(bt[3])))
        kernel_name = "spmv_bcsr4_tile<2>" if bt[1].value else "spmv_bcsr4<2>"
        sb, sf, ss, spad = _ct.c_int(), _ct.c_int(), _ct.c_longlong(), _ct.c_double()
        sus = (_ct.c_double * 4)()
        mpk.check(mpk.lib().mi_bcsr4_sell_info(A.handle, _ct.byref(sb), _ct.byref(sf), _ct.byref(ss), _ct.byref(spad), sus))
        bcsr_forms = dict(us_row_per_quad=round(bt[2].value, 2), us_row_per_quad_x_tile=round(bt[3].value, 2), sliced_copy_built=bool(sb.value),
                          sliced_form_in_use=sf.value, sliced_steps=ss.value, sliced_padding=round(spad.value, 5),
                          us_sliced=dict(one_wave_per_simd_d8_nt=round(sus[0], 2), one_wave_per_simd_d8_temporal=round(sus[1], 2), two_waves_per_simd_d4_nt=round(sus[2], 2),
                                         one_wave_per_simd_d12_nt=round(sus[3], 2)))
        if sf.value >= 0:
            kernel_name = ["spmv_bcsr4_sell<8, true, 0, 2, 4>", "spmv_bcsr4_sell<8, false, 0, 2, 4>", "spmv_bcsr4_sell<4, true, 0, 2, 8>", "spmv_bcsr4_sell<12, true, 0, 2, 4>"][sf.value]
        ring_cfg, ring_runs, ring_bad, ring_frac = 0, 0, 0, 0.0
        x = torch.from_numpy(x_host).cuda()
        ys = [torch.empty(n, dtype=torch.float64, device="cuda")]
        nvec = int(W.get("spmm", 0))
        if nvec:
            kernel_name = (f"spmm_bcsr4_quad<{nvec}, 0, {'true' if os.environ.get('MI355_SPMM_XCD', '1' if nvec > 4 else '0') != '0' else 'false'}, {os.environ.get('MI355_SPMM_DEPTH') or (2 if nvec == 4 else 1)}>"
                           if nvec % 4 == 0 and os.environ.get("MI355_SPMM_QUAD") != "0" else None) or f"spmm_bcsr4<{nvec}, 0, {'true' if nvec <= 4 else 'false'}, {'true' if os.environ.get('MI355_SPMM_XCD', '1' if nvec > 4 else '0') != '0' else 'false'}>"
            Xh = np.stack([np.sin(0.001 * np.arange(n) + j) for j in range(nvec)])  # v_i[j] = sin(0.001 j + i), mpk/2SpMV.cpp:110-116
            Xd = torch.from_numpy(Xh).cuda()
            Yd = torch.empty((nvec, n), dtype=torch.float64, device="cuda")

            def step():
                mpk.MatMatMult_SeqBAIJ_4(A, Xd, Yd, "chain")
            step()  # the first product of the handle at this column count times the gather and the tile form and keeps the faster
            torch.cuda.synchronize()
            si = [_ct.c_int(), _ct.c_int(), _ct.c_int()]
            us3 = (_ct.c_double * 5)()
            mpk.check(mpk.lib().mi_bcsr4_spmm_info(A.handle, nvec, _ct.byref(si[0]), _ct.byref(si[1]), _ct.byref(si[2]), us3))
            spmm_info = dict(tile_built=bool(si[0].value), form_in_use=["gather", "tile", "octet tile", "octet tile, non-temporal coefficients", "sliced stream"][si[1].value], longest_list=si[2].value,
                             us_by_form=dict(gather=round(us3[0], 2), tile=round(us3[1], 2), octet_tile=round(us3[2], 2), octet_tile_nt=round(us3[3], 2), sliced_stream=round(us3[4], 2)))
            if si[1].value == 1:
                kernel_name = f"spmm_bcsr4_tile<{nvec}, 0, 3>"
            elif si[1].value in (2, 3):
                kernel_name = f"spmm_bcsr4_otile<{nvec}, 0, 4, {'true' if si[1].value == 3 else 'false'}>"
            elif si[1].value == 4:
                kernel_name = f"spmm_bcsr4_sell<{nvec}, 0, 6, true>"
        else:
            def step():
                mpk.SpMV_BCSR(ys[0], x, A)
        halo_info = None
    elif world == 1:
        t_gen = time.perf_counter() - t_setup  # (generating the synthetic matrix on the host: not the library's time)
        A = mpk.csrmatrix(n, p, c, v)
        if args.kernel != "auto":
            A.set_kernel(args.kernel)
        torch.cuda.synchronize()
        mem_before = torch.cuda.mem_get_info()[0]
        t_create = time.perf_counter()
        _ = A.handle
        t_create = time.perf_counter() - t_create
        torch.cuda.synchronize()
        handle_device_bytes = mem_before - torch.cuda.mem_get_info()[0]  # what mi_csr_create left allocated (its scratch is released before it returns)
        kernel_name = A.kernel_name()
        ring_cfg, ring_runs, ring_bad, ring_frac = A.ring_info()
        x = torch.from_numpy(x_host).cuda()
        ys = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(k)]
        if k == 1 and not args.internal and nnz_global >= 20_000_000 and len(p) - 1 == n:
            # The timed region runs on what a caller gets without doing anything: plain allocations (round 4; round 3's headline ran on a
            # best-of-8 placed pair).  --placed-vectors measures the library-placed pair as an EXTRA, never as `value`.
            vector_info = dict(placed=False, note="x and y of the timed region are plain torch allocations")
            if args.placed_vectors:
                def quick(xv, yv, reps=60):
                    for _ in range(10):
                        mpk.SpMV_CSR(yv, xv, A)
                    q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    q0.record()
                    for _ in range(reps):
                        mpk.SpMV_CSR(yv, xv, A)
                    q1.record()
                    q1.synchronize()
                    return q0.elapsed_time(q1) / reps * 1e3
                us_torch = quick(x, ys[0])
                try:
                    (xp_, yp_), cand_us = A.alloc_vectors(2, draws=8)
                    xp_.copy_(x)
                    vector_info.update(torch_allocated_us=round(us_torch, 2), extra_placed_pair_us=round(quick(xp_, yp_), 2), candidate_pairs_us=cand_us)
                    del xp_, yp_
                except Exception as e:  # noqa: BLE001
                    vector_info.update(torch_allocated_us=round(us_torch, 2), extra_placed_pair_note=f"mi_vec_alloc_placed failed ({type(e).__name__}: {str(e)[:120]})")
        if k == 1 and args.internal:
            x_caller = x
            x = A.to_internal(x_caller)  # once per solve, not once per product

            def step():
                mpk.SpMV_CSR_internal(ys[0], x, A)
        elif k == 1:
            def step():
                mpk.SpMV_CSR(ys[0], x, A)
        else:
            def step():
                mpk.SpMkV(ys, x, A)
        halo_info = None
    else:
        kernel_name = "interior+boundary pieces, kernel=" + args.kernel
        sp = mpk._stream_ptr()  # the bench stays on one stream: look it up once, not per step
        red_dev0 = "cuda" if backend == "nccl" else "cpu"

        def agree_min(v):  # collective: the smallest of every rank's integer
            t = torch.tensor([int(v)], dtype=torch.int32, device=red_dev0)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t)

        def timed_max_us(fn, reps):  # collective: microseconds per call, the slowest rank's
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0_) / reps * 1e6], dtype=torch.float64, device=red_dev0)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)

        # Every candidate exchange is brought up (collective), must equal the torch.distributed exchange bit for bit (DistCSR's
        # self-check), survive a dry run without a wait giving up on any rank, and is then timed; the fastest survivor runs the
        # timed region.  --exchange push|native|torch restricts the candidates to one (plus torch as the way out).
        want = [args.exchange] if args.exchange in ("push", "native", "allgather", "torch") else ["push", "native", "allgather", "torch"]
        if "torch" not in want:
            want.append("torch")
        probe_steps = max(5, min(args.steps, 50))
        probes, alive = {}, {}
        rccl_ranks = None
        t_bench0 = float(os.environ.get("MI355_BENCH_T0", "0")) or time.time()  # (self_launch stamps its own start: python start-up and rendezvous count too)

        def say(msg):  # progress on stderr, rank 0: the first real 8-GPU record explains itself even if it is cut short
            if rank == 0:
                print(f"[bench +{time.time() - t_bench0:6.1f}s] {msg}", file=sys.stderr, flush=True)
        say(f"{world} ranks up ({backend}); matrix generated; probing exchanges {want}")
        phase_s = {}
        for ex in want:
            t_ex = time.perf_counter()

            def done(ex=ex, t_ex=t_ex):
                phase_s[ex] = round(time.perf_counter() - t_ex, 2)
                probes[ex]["probe_s"] = phase_s[ex]
                say(f"exchange {ex}: {'ok, ' + str(probes[ex].get('step_us')) + ' us per step' if probes[ex]['ok'] else 'dropped: ' + probes[ex]['note'][:160]} ({phase_s[ex]} s)")
            try:
                dcx = D.DistCSR(rs, p, c, v, kernel=None if args.kernel == "auto" else args.kernel, exchange=ex)
            except D.DistSetupError as e:  # raised on every rank alike
                probes[ex] = dict(ok=False, note=str(e)[:200])
                done()
                continue
            t_created = time.perf_counter() - t_ex
            got = "push" if dcx.push else (("allgather" if dcx.allgather else "native") if dcx.native else "torch")
            if ex in ("native", "allgather") or dcx.rccl_ranks:
                rccl_ranks = dcx.rccl_ranks
            if got != ex:  # (a collective outcome: the same on every rank)
                probes[ex] = dict(ok=False, note=f"did not come up on every rank, or failed its bitwise self-check against the torch.distributed exchange (DistCSR fell to '{got}')")
                dcx.close()
                done()
                continue
            xx = dcx.new_x_ext()
            xx[: dcx.n_local] = torch.from_numpy(x_host).cuda()
            if k == 1:
                yy = dcx.new_y()
                pb = None

                def stepx(dcx=dcx, xx=xx, yy=yy):
                    dcx.spmv(xx, yy, sp)
            else:  # matrix powers across ranks: one halo exchange per power
                yy = None
                pb = dcx.new_power_buffers(k)

                def stepx(dcx=dcx, xx=xx, pb=pb):
                    dcx.spmk(xx, pb, sp)
            fine, why = 1, ""
            try:
                for _ in range(max(3, args.warmup)):
                    stepx()
                torch.cuda.synchronize()
                dcx.status()
            except Exception as e:  # noqa: BLE001
                fine, why = 0, f"{type(e).__name__}: {str(e)[:160]}"
            if agree_min(fine) == 0:
                probes[ex] = dict(ok=False, note="failed in the dry run" + (f" on this rank ({why})" if why else " on another rank"))
                try:
                    dcx.close()
                except Exception:  # noqa: BLE001
                    pass
                done()
                continue
            us = timed_max_us(stepx, probe_steps)
            fine = 1
            try:
                dcx.status()
            except Exception:  # noqa: BLE001
                fine = 0
            if agree_min(fine) == 0:
                probes[ex] = dict(ok=False, note="a halo wait gave up while the exchange was being timed")
                try:
                    dcx.close()
                except Exception:  # noqa: BLE001
                    pass
                done()
                continue
            probes[ex] = dict(ok=True, step_us=round(us, 2),
                              form=("ONE launch per step" if dcx.push_fused else "four launches per step") if ex == "push" else
                                   ("RCCL send/recv on the partition's comm stream, events" if ex == "native" else
                                    "ONE ncclAllGather of every rank's boundary slice on the comm stream, events" if ex == "allgather" else
                                    ("all_to_all_single over RCCL" if dcx._nccl else "host-staged (non-NCCL backend, development)")),
                              create_s=round(t_created, 2),
                              kernels=dict(interior=mpk.lib().mi_part_kernel_name(dcx._h, 0).decode(), boundary=mpk.lib().mi_part_kernel_name(dcx._h, 1).decode(),
                                           one_launch_step=mpk.lib().mi_part_kernel_name(dcx._h, 2).decode() or None))
            alive[ex] = (dcx, stepx, xx, yy, pb)
            done()
        if not alive:
            sys.exit("no halo exchange survived its checks on every rank: " + json.dumps(probes))
        chosen = args.exchange if args.exchange in alive else min(alive, key=lambda e: probes[e]["step_us"])
        say(f"chosen: {chosen}; timed region next ({args.warmup} + {args.steps} steps)")
        for ex in list(alive):
            if ex != chosen:
                alive.pop(ex)[0].close()
        dc, step, x_ext, y, pbufs = alive[chosen]
        # what the same rank's kernels cost WITHOUT any exchange (pack + interior rows + boundary rows, back to back), slowest rank
        L_ = mpk.lib()
        ybuf = dc.new_y()
        sbuf = torch.empty(max(dc.n_send, 1), dtype=torch.float64, device="cuda")
        vp_ = _ct.c_void_p

        def compute_only():
            for _ in range(k):
                mpk.check(L_.mi_part_pack_dev(dc._h, vp_(x_ext.data_ptr()), vp_(sbuf.data_ptr()), sp))
                mpk.check(L_.mi_part_spmv_interior_dev(dc._h, vp_(x_ext.data_ptr()), vp_(ybuf.data_ptr()), sp))
                mpk.check(L_.mi_part_spmv_boundary_dev(dc._h, vp_(x_ext.data_ptr()), vp_(ybuf.data_ptr()), sp))
        for _ in range(3):
            compute_only()
        us_compute = timed_max_us(compute_only, probe_steps)
        tb = torch.tensor([8.0 * dc.n_send * k, 8.0 * dc.n_halo * k], dtype=torch.float64, device=red_dev0)
        tb_max = tb.clone()
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        dist.all_reduce(tb_max, op=dist.ReduceOp.MAX)
        us_step = probes[chosen]["step_us"]
        halo_info = dict(n_halo=dc.n_halo, n_send=dc.n_send, interior_rows=dc.n_interior, boundary_rows=dc.n_boundary,
                         exchange=("peer push over HIP IPC windows, no RCCL (mi_part_spmv_push_dev), " + probes[chosen]["form"]) if chosen == "push"
                         else ("native RCCL send/recv (mi_part_spmv_dev)" if chosen == "native" else
                               "native RCCL all-gather of boundary slices (mi_part_spmv_dev, mi_part_allgather_setup)" if chosen == "allgather" else "torch.distributed " + probes[chosen]["form"]),
                         chosen=chosen, chosen_by=("--exchange" if args.exchange in alive else f"fastest of the survivors over {probe_steps} steps, slowest rank's time"),
                         exchanges=probes,
                         exchange_bytes_per_step=dict(sent_all_ranks=int(tb[0]), received_all_ranks=int(tb[1]), sent_max_rank=int(tb_max[0]),
                                                      received_max_rank=int(tb_max[1]), note="8 B per halo entry, once per product"),
                         overlap=dict(step_us=us_step, compute_only_us=round(us_compute, 2), exchange_exposed_us=round(us_step - us_compute, 2),
                                      compute_over_step=round(us_compute / us_step, 4),
                                      note="compute_only = pack + interior rows + boundary rows of the same rank with NO exchange, slowest rank; "
                                           "exposed = what the exchange adds to the step (negative: the one-launch step is cheaper than the three kernels it replaces)"),
                         rccl_ranks=rccl_ranks, torch_backend=backend, torch_world=world,
                         probe_seconds=dict(phase_s, note="wall seconds per candidate exchange on rank 0: DistCSR create (partition, pieces, their create-time "
                                                          "measurements, the exchange's set-up and bitwise self-check) + dry run + timing"))
        del ybuf, sbuf
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up, then the timed region ---------------------------------------------------
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if args.cold:
        ev_ms = 0.0
        for _ in range(args.steps):
            mpk.flush_cache(sync=False)  # eviction enqueued in front of the launch: cold caches, GPU not idle
            ev0.record()
            step()
            ev1.record()
            torch.cuda.synchronize()
            ev_ms += ev0.elapsed_time(ev1)
        wall = ev_ms / 1e3  # flush time excluded
    else:
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
    barrier()
    if world > 1:  # a hand-off / halo wait that gave up during the timed region fails the run here, loudly and on every rank
        fine = 1
        try:
            dc.status()
        except Exception as e:  # noqa: BLE001
            fine = 0
            print(f"[rank {rank}] {e}", file=sys.stderr, flush=True)
        if agree_min(fine) == 0:
            sys.exit("a halo wait gave up during the timed region: no number to report")
    # ---- the same kernel in two other regimes (N = 1, y = A x workloads only; never `value`) -------------------
    extra = {}
    if world == 1 and k == 1 and not args.cold and not args.no_extras and not W.get("spmm") and not args.internal:
        # (1) cold single shot: L2s and the 256 MiB Infinity Cache evicted before EVERY launch (the reference's own
        #     protocol: flush_cache() before each timed call, mpk/SpM2V.cpp:895-904)
        #     Two forms: the eviction ENQUEUED in front of the launch (cold caches, GPU never idle: what "cold cache" means on
        #     the CPU, where there is no idle state to wake from) and the synchronous form (the device idles while the host
        #     returns and launches: a kernel started on an idle MI355X runs 8-26 us longer whatever the caches hold,
        #     tools/cold_probe.py); plus a single warm launch from idle, which separates the two effects.
        ncold = 12

        def single_shots(prepare):
            ms = 0.0
            for _ in range(ncold):
                prepare()
                ev0.record()
                step()
                ev1.record()
                torch.cuda.synchronize()
                ms += ev0.elapsed_time(ev1)
            return ms * 1e3 / ncold
        extra["cold_us"] = single_shots(lambda: mpk.flush_cache(sync=False))
        extra["cold_idle_us"] = single_shots(lambda: mpk.flush_cache())
        extra["warm_idle_us"] = single_shots(lambda: None)
        # (2) inside the Krylov-step pipeline SpMV -> dot + AXPY (orthogonalize) -> SpMV of mpk/SpMVmulti.cpp:559-574:
        #     the vector kernels between two products compete with the matrix stream for the caches
        if not bcsr:
            bvec = torch.cos(0.002 * torch.arange(n, dtype=torch.float64, device="cuda"))
            x3 = torch.empty_like(ys[0])
            z = torch.empty_like(ys[0])

            def pipe():
                mpk.SpMV_CSR(ys[0], x, A)
                mpk.orthogonalize(n, bvec, ys[0], x3, 1e-8)
                mpk.SpMV_CSR(z, x3, A)

            def timed(fn, reps):
                for _ in range(5):
                    fn()
                ev0.record()
                for _ in range(reps):
                    fn()
                ev1.record()
                torch.cuda.synchronize()
                return ev0.elapsed_time(ev1) * 1e3 / reps
            def pipe_fused():  # the dot rides in the first product's epilogue: two launches + one instead of two + two
                mpk.SpMV_CSR_orthogonalize(ys[0], x, A, bvec, x3, 1e-8)
                mpk.SpMV_CSR(z, x3, A)
            t_pipe = timed(pipe, 50)
            t_orth = timed(lambda: mpk.orthogonalize(n, bvec, ys[0], x3, 1e-8), 50)
            extra["pipeline_fused_pass_us"] = timed(pipe_fused, 50)
            extra["pipeline_dot_in_epilogue"] = A.dot_in_epilogue()
            extra["pipeline_pass_us"] = t_pipe
            extra["pipeline_orthogonalize_us"] = t_orth
            extra["pipeline_spmv_us"] = (t_pipe - t_orth) / 2
            step()  # leave ys[0] = A x for the parity check below
            torch.cuda.synchronize()
            del bvec, x3, z
    # (3) what a Newton iteration pays beside its products (src/solve_newton.c:1245-1247: the Jacobian's VALUES change every iteration, the
    #     pattern never): mi_csr_update_values_dev on the benched handle — CSR values, the sliced copy the headline kernel streams and the
    #     blocked copy where there is one, refilled on the update's stream — and what the handle keeps allocated on the device.
    if world == 1 and not args.no_extras and k == 1 and not W.get("spmm"):
        vdev = torch.from_numpy(np.ascontiguousarray(bv if bcsr else v)).cuda()  # the same values: the parity check below still sees the timed region's result
        for _ in range(2):
            A.update_values(vdev)
        ev0.record()
        for _ in range(5):
            A.update_values(vdev)
        ev1.record()
        torch.cuda.synchronize()
        extra["update_values_us"] = ev0.elapsed_time(ev1) * 1e3 / 5
        del vdev
        step()  # leave ys[0] = A x for the parity check below (computed from the refilled copies)
        torch.cuda.synchronize()
    red_dev = "cuda" if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    if world > 1:
        tt = torch.tensor([wall, ev_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tt[0]), float(tt[1])

    # ---- parity of what was just timed (rank-local rows, against the oracle) ---------------------
    parity = None
    if not args.no_parity:
        from oracle import oracle as O  # checker only
        if world == 1 and bcsr and W.get("spmm"):
            got = Yd.cpu().numpy()
            refs = [O.spmv_bcsr4(bp, bc, bv, Xh[j]) for j in range(nvec)]
            parity = dict(rel_error=max(O.rel_error(refs[j], got[j]) for j in range(nvec)),
                          bitwise=all(np.array_equal(refs[j].view(np.uint64), got[j].view(np.uint64)) for j in range(nvec)),
                          against="oracle SpMV_BCSR_FMA restatement, every column")
        elif world == 1 and bcsr:
            Yb = O.spmv_bcsr4(bp, bc, bv, x_host)
            got = ys[0].cpu().numpy()
            parity = dict(rel_error=O.rel_error(Yb, got), bitwise=bool(np.array_equal(Yb.view(np.uint64), got.view(np.uint64))),
                          against="oracle SpMV_BCSR_FMA restatement (bit-pinned to the reference's object code)")
        elif world == 1:
            Y = O.spmk_chain(k, p, c, v, x_host)
            got = [(A.from_internal(t) if (args.internal and k == 1) else t).cpu().numpy() for t in ys]
            parity = dict(rel_error=max(O.rel_error(Y[i], got[i]) for i in range(k)),
                          bitwise=all(np.array_equal(Y[i].view(np.uint64), got[i].view(np.uint64)) for i in range(k)),
                          against="oracle/cpu_ref.c fma chain (= reference SpMV_CSR_OPT/_FMA), full vectors")
        else:
            # rank-local check needs the halo values: take them from the generator (x_j = sin(0.001 j))
            halo_ids = np.empty(dc.n_halo, np.int64)
            off = 0
            L = mpk.lib()
            import ctypes
            for q in range(world):
                if dc.recv_counts[q]:
                    mpk.check(L.mi_part_recv_ids(dc._h, q, halo_ids[off:].ctypes.data))
                    off += dc.recv_counts[q]
            if dc.n_halo:  # the generator's own sin (libm), not numpy's: bitwise on any host
                hmin, hmax = int(halo_ids.min()), int(halo_ids.max())
                xh = synth.x_sin(hmin, hmax + 1)[halo_ids - hmin]
            else:
                xh = np.zeros(0)
            xe = np.concatenate([x_host, xh])
            # relabel global columns like the planner: owned -> local, ghosts -> n_local + position
            cl = np.where((c >= lo) & (c < hi), c - lo, dc.n_local + np.searchsorted(halo_ids, c)).astype(np.int32)
            if k == 1:
                same = np.array_equal(O.spmv(p, cl, v, xe).view(np.uint64), y.cpu().numpy().view(np.uint64))
                how = "oracle fma chain on each rank's rows, halos from the generator"
            else:  # power 1 as above (checks the exchange); power i+1 from power i's exchanged buffer
                same = np.array_equal(O.spmv(p, cl, v, xe).view(np.uint64), pbufs[0][: dc.n_local].cpu().numpy().view(np.uint64))
                if dc.push_fused:  # that step reads ghosts from its window and leaves the halo parts alone: fetch them for the check
                    for i in range(k - 1):
                        dc.refresh_halo(pbufs[i])
                    torch.cuda.synchronize()
                for i in range(1, k):
                    yo = O.spmv(p, cl, v, pbufs[i - 1].cpu().numpy())
                    same = same and np.array_equal(yo.view(np.uint64), pbufs[i][: dc.n_local].cpu().numpy().view(np.uint64))
                how = ("oracle fma chain on each rank's rows: power 1 with halos from the generator, "
                       "power i+1 from power i's exchanged [owned | halo] buffer")
            bad = torch.tensor([0 if same else 1], device=red_dev)
            dist.all_reduce(bad)
            parity = dict(bitwise=bool(int(bad) == 0), against=how)
            del ctypes

    # ---- numbers -------------------------------------------------------------------------------
    nvec = int(W.get("spmm", 0)) or 1
    flops = 2.0 * nnz_global * k * nvec * args.steps
    value = flops / wall / 1e9
    launches = args.steps * k
    n_loc, nnz_loc = (n, nnz_global) if world == 1 else (hi - lo, len(c))
    B_csr = algorithmic_bytes(n_loc, nnz_loc)  # the reference's own model, CSR arrays
    # bytes of the format the launched kernel actually reads (what `frac` is priced on, so it can never exceed 1):
    # 16 values + 1 block column per block, block-row pointers, x once, y once for the BCSR kernel
    runs_blocked = bcsr or (world == 1 and "bcsr4" in kernel_name)
    B_exec = (132 * (nnz_loc // 16) + 4 * (n_loc // 4 + 1) + 16 * n_loc * nvec) if runs_blocked else B_csr  # x, y once per column
    launch_s = ev_ms / 1e3 / launches
    achieved = B_exec / launch_s / 1e9
    roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4),
                    traffic=load_traffic(args.workload, kernel_name)[0] if world == 1 else None,
                    traffic_source=load_traffic(args.workload, kernel_name)[1] if world == 1 else None,
                    traffic_note="PROFILED value, not measured in this run: HBM-side bytes per launch of this kernel on this workload from the rocprofv3 "
                                 "--pmc FETCH_SIZE / WRITE_SIZE passes committed under profiles/ (file named in traffic_source, with its round, command "
                                 "and the gfx950 x2 read correction); counters cannot be read inside the timed run",
                    kernel=kernel_name, algorithmic_bytes_per_launch=B_exec, launch_us=round(launch_s * 1e6, 2),
                    bytes_model=("BCSR 4x4: 132 B per block + 4 B per block row + 16 B per row (the format the kernel reads)" if runs_blocked
                                 else "CSR: 12 B per nonzero + 4 B per row pointer + 16 B per row (SURVEY.md §8d)"),
                    timing="HIP events on the launch stream around the timed region / launches"
                           + (" (per-rank share incl. halo exchange; max over ranks)" if world > 1 else ""))
    if runs_blocked and not bcsr:
        # the caller handed over CSR arrays; AUTO runs the BCSR kernel on a blocked copy (same bits, 8.25 instead of 12 B per
        # nonzero).  Priced in the CSR bytes the caller's format would have cost, the same launch reads "faster than HBM":
        roofline["csr_equivalent"] = dict(bytes_per_launch=B_csr, gbs=round(B_csr / launch_s / 1e9, 1),
                                          of_peak=round(B_csr / launch_s / 1e9 / HBM_PEAK_GBS, 4),
                                          note="NOT a roofline fraction: CSR-model bytes over the time of a kernel that reads the blocked copy")
    if k > 1 and world == 1 and not bcsr:
        out_spmk = A.spmk_info(k)
    if k > 1 and world == 1:
        # matrix powers: `frac` above prices each of the k launches at the un-fused traffic k*B; a fused kernel that
        # read the matrix once would need B_fused = 12 nnz + 4 (n+1) + 8 n (1 + k) for the whole step
        B_fused = 12 * nnz_loc + 4 * (n_loc + 1) + 8 * n_loc * (1 + k)
        step_s = ev_ms / 1e3 / args.steps
        roofline["fused_lower_bound"] = dict(bytes_per_step=B_fused, unfused_bytes_per_step=k * B_csr,
                                             step_us=round(step_s * 1e6, 2),
                                             frac_of_peak_if_fused_bytes=round(B_fused / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                             note="time of the whole k-step over the bytes a perfectly fused kernel would move; "
                                                  "the gap to `frac` is what fusing the k sweeps could still buy")
    if roofline["traffic"]:
        # `frac` prices the ALGORITHMIC bytes (12 B per nonzero for CSR) over the launch time; the sliced kernels stream 10 B per
        # nonzero, so it can pass 1.0 on a fast box without anything moving faster than the memory: the counter traffic over the same time says what did
        roofline["frac_on_traffic"] = round(roofline["traffic"] / launch_s / 1e9 / HBM_PEAK_GBS, 4)
        roofline["frac_note"] = ("frac = algorithmic bytes / launch time / peak (the contract's definition); frac_on_traffic = the profiled fabric traffic of this kernel "
                                 "(FETCH_SIZE x 2 + WRITE_SIZE) over the same time: what actually moved, as a fraction of the 8 TB/s peak — a frac above 1.0 means the "
                                 "kernel's format holds fewer bytes than the CSR model charges for, not that memory ran above its peak")
    if world == 1 and not args.no_extras:
        # this box's own streaming rate: a plain read sweep over as many bytes as the kernel's format holds (MI355X boxes of one
        # pool differ by 10-20 % in it); NOT the roofline's peak — `frac` stays priced against the 8 TB/s spec
        us_stream = mpk.stream_read_us(B_exec)
        roofline["this_box_stream_read"] = dict(bytes=B_exec, launch_us=round(us_stream, 2), gbs=round(B_exec / us_stream / 1e3, 1),
                                                kernel_over_stream=round(us_stream / (launch_s * 1e6), 4),
                                                note="plain non-temporal read sweep of the same number of bytes, same process, back to back: "
                                                     "kernel_over_stream = the kernel's rate as a fraction of it")
    if extra:
        if "cold_us" in extra:
            roofline["cold_single_shot"] = dict(launch_us=round(extra["cold_us"], 2),
                                                frac=round(B_exec / (extra["cold_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                protocol="mi_flush_cache_async() (512 MiB device fill + 512 MiB read sweep: L2s and Infinity Cache left full of clean "
                                                         "lines of an unused buffer) enqueued in front of EVERY launch, 12 launches, one HIP event pair each",
                                                from_idle_us=round(extra["cold_idle_us"], 2),
                                                from_idle_frac=round(B_exec / (extra["cold_idle_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                from_idle_note="same with the synchronous mi_flush_cache(): the device idles before the launch",
                                                warm_from_idle_us=round(extra["warm_idle_us"], 2),
                                                warm_from_idle_note="no eviction at all, one launch per synchronise: what starting on an idle GPU costs by itself")
        if "pipeline_spmv_us" in extra:
            roofline["in_pipeline"] = dict(spmv_us=round(extra["pipeline_spmv_us"], 2),
                                           frac=round(B_exec / (extra["pipeline_spmv_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                           pass_us=round(extra["pipeline_pass_us"], 2), orthogonalize_us=round(extra["pipeline_orthogonalize_us"], 2),
                                           pass_dot_in_epilogue_us=round(extra["pipeline_fused_pass_us"], 2), dot_in_epilogue=extra["pipeline_dot_in_epilogue"],
                                           protocol="SpMV -> orthogonalize (dot + AXPY) -> SpMV, mpk/SpMVmulti.cpp:559-574, device-resident, 50 passes; "
                                                    "spmv_us = (pass - orthogonalize alone) / 2; pass_dot_in_epilogue_us = the same pass through "
                                                    "mi_spmv_orthogonalize_dev (b . x1 accumulated in the first product's epilogue)")
        if "update_values_us" in extra:
            roofline["newton_refresh"] = dict(update_values_us=round(extra["update_values_us"], 1), products=round(extra["update_values_us"] / (launch_s * 1e6), 2),
                                              device_bytes_per_nnz=None if bcsr else round(handle_device_bytes / max(nnz_global, 1), 2), device_bytes=None if bcsr else int(handle_device_bytes),
                                              note="mi_csr_update_values_dev on the benched handle (a Newton loop's Jacobian: new values, same pattern; "
                                                   "src/solve_newton.c:1245-1247) — ONE pass reads the caller's values and writes the CSR values and the sliced copy "
                                                   "(+ the blocked copy's refill where the handle has one); `products` = that time in products of this line; "
                                                   "device_bytes = device memory mi_csr_create left allocated (hipMemGetInfo around it): CSR arrays 12 B per nonzero, "
                                                   "ring plan + 16-bit column stream ~2, the sliced copy ~10 where it won the create-time measurement (released otherwise), "
                                                   "placed x / y scratch")
    out = dict(metric="fp64 CSR SpMV GFLOP/s & % HBM roofline @ nnz; 1/2/4/8 GPU", value=round(value, 2), unit="GFLOP/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(wall * 1e3 / args.steps, 5),
               higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload=W["desc"], name=args.workload, n=n, nnz=nnz_global, k=k, seed="0x5EED",
                           half_bandwidth=synth.DEFAULT_W, partition=f"row-range x{world}", cold=bool(args.cold),
                           numbering="internal (mi_spmv_internal_dev; vectors permuted once outside the timed region)" if args.internal else "caller's"),
               roofline=roofline, pct_hbm_roofline=round(100 * achieved / HBM_PEAK_GBS, 2))
    if world == 1 and not bcsr:
        tune, nt = A.tune_detail()
        out["kernel_info"] = dict(kernel=kernel_name, ring_config=ring_cfg, runs=ring_runs, runs_on_plain_path=ring_bad,
                                  nnz_fraction_ring=round(ring_frac, 4), nontemporal_values=nt,
                                  matrix_stream_bytes_per_nnz=10 if ("ring<" in kernel_name or "sstream<" in kernel_name) else (8.25 if "bcsr4" in kernel_name else
                                                              round(10 + 4 * A.tile_info()["unique_per_nnz"], 2) if "tile" in kernel_name else 12),
                                  autotune_us={k_: round(v_, 1) for k_, v_ in tune.items()})
        if "spmv_csr_ring<" in kernel_name:
            rs = A.ring_shape_info()
            out["kernel_info"]["ring_plan"] = dict(row_blocks=rs["blocks"], lean=rs["lean"], prefetch_depth=rs["depth"], value_loads="16 B per lane (nonzero pairs)",
                                                   block_shape="cut at the nonzero count (>= 20 M nnz: insensitive to where x and y lie)" if nnz_global >= 20_000_000
                                                   else "whole waves of rows")
            if rs["us_aligned"] > 0:  # MI355_RING_SHAPE_COMPARE=1
                out["kernel_info"]["ring_plan"].update(us_blocks_of_whole_waves=round(rs["us_aligned"], 1), us_unaligned_blocks=round(rs["us_unaligned"], 1))
        mi = A.mring_info()
        if mi["built"]:
            out["kernel_info"]["mring_plan"] = dict(runs=mi["runs"], runs_on_plain_path=mi["runs_not_served"], nnz_fraction_served=round(mi["nnz_fraction"], 4))
            if "mring<" in kernel_name:
                rs = A.ring_shape_info()
                if rs["us_aligned"] > 0:  # MI355_RING_SHAPE_COMPARE=1
                    out["kernel_info"]["mring_plan"].update(us_blocks_of_whole_waves=round(rs["us_aligned"], 1), us_unaligned_blocks=round(rs["us_unaligned"], 1))
        if vector_info:
            out["kernel_info"]["vectors"] = vector_info
        pl = A.placement_info()
        if pl["values"]:
            out["kernel_info"]["placement_draws_us"] = dict(pl, note="mi_csr_create timed the chosen kernel on fresh device copies of the value array, then of the 16-bit column "
                                                                  "stream, and kept the fastest of each (first entry: as first allocated); where the arrays — the caller's x and y "
                                                                  "included — lie in device memory moves a WARM launch by up to 15 %, a cold one not at all (profiles/NOTES.md §4.12)")
        si_ = A.sstream_info()
        if si_["built"]:
            out["kernel_info"]["sliced_stream"] = {k_: (round(v_, 4) if isinstance(v_, float) else v_) for k_, v_ in si_.items()}
        ti = A.tile_info()
        if ti["built"]:
            out["kernel_info"]["tile_plan"] = dict(row_blocks=ti["nblk"], distinct_columns_per_nnz=round(ti["unique_per_nnz"], 4))
        ri = A.reorder_info()
        if ri["reordered"] or ri["block"]:
            out["reorder"] = {k_: (round(v_, 1) if isinstance(v_, float) else v_) for k_, v_ in ri.items()}
    if k > 1 and world == 1 and not bcsr:
        out["kernel_info"]["powers_step"] = dict(one_launch=out_spmk["one_launch"], eligible=out_spmk["eligible"],
                                                 us_k_launches=round(out_spmk["us_k_launches"], 2), us_one_launch=round(out_spmk["us_one_launch"], 2),
                                                 note="first k-step of the handle times both forms (same bits) and keeps the faster; MI355_SPMK_FUSED=0|1 forces")
    if world == 1 and bcsr and W.get("spmm"):
        out["kernel_info"] = dict(kernel=kernel_name, **spmm_info)
    elif world == 1 and bcsr:
        out["kernel_info"] = dict(kernel=kernel_name, forms=bcsr_forms)
    if parity is not None:
        out["parity"] = parity
    if halo_info is not None:
        out["halo"] = halo_info
    out["setup_s"] = round(t_setup, 2)
    if world == 1 and not bcsr:
        out["setup_breakdown_s"] = dict(generate_matrix_on_host=round(t_gen, 2), mi_csr_create=round(t_create, 2),
                                        note="mi_csr_create = upload + plans + the create-time measurements (kernel candidates, placement draws)")
    # (the one-process extra costs another mi_dist_create over all N devices: skipped — collectively, by rank 0's clock — when this run has
    # already used more than 120 s, so that the whole command stays well inside the driver's window)
    sp_budget_ok = True
    if world > 1:
        used = torch.tensor([time.time() - (float(os.environ.get("MI355_BENCH_T0", "0")) or t_wall0)], dtype=torch.float64, device=red_dev)
        dist.broadcast(used, src=0)
        sp_budget_ok = float(used) <= float(os.environ.get("MI355_BENCH_SP_BUDGET_S", "120"))
        if not sp_budget_ok and rank == 0 and not args.no_single_process_extra:
            out["single_process"] = dict(ok=False, skipped=True, note=f"skipped: the process-per-GPU path had used {float(used):.0f} s of wall time (> 120 s) when it was its turn")
    if world > 1 and sp_budget_ok and not args.no_single_process_extra and kind in ("s15", "fe") and not args.cold:
        # the same workload through ONE process (mi_dist_*), as a child of rank 0.  The other ranks wait on the HOST — a key in the
        # process group's own TCP store — so that no collective kernel spins on their GPUs while the child measures; everything here
        # is bounded (child: 200 s; the wait: 260 s) and a failure costs the line nothing but the field.
        torch.cuda.synchronize()
        key = "mi355_single_process_done"
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:  # noqa: BLE001
            store = None
        if rank == 0:
            try:
                out["single_process"] = run_single_process_child(args, timeout_s=200)
            except Exception as e:  # noqa: BLE001
                out["single_process"] = dict(ok=False, note=f"{type(e).__name__}: {str(e)[:200]}")
            out["single_process"]["note2"] = ("the same workload driven by ONE process over the N devices (python bench.py --gpus N --single-process: mi_dist_create, a worker "
                                              "thread per rank), run as a child of rank 0 after the timed region while the other ranks wait on the host; never `value`")
            if store is not None:
                try:
                    store.set(key, "1")
                except Exception:  # noqa: BLE001
                    pass
        elif store is not None:
            import datetime
            try:
                store.wait([key], datetime.timedelta(seconds=260))
            except Exception:  # noqa: BLE001 — rank 0 never got there: nothing to wait for any longer
                pass
    if rank == 0 and world == 1 and not bcsr and k == 1 and not args.no_cpu_baseline:
        # the reference's calling convention (host pointers): x H2D + kernel + y D2H per call; never `value`
        xh, yh = x_host, np.empty(n)
        mpk.SpMV_CSR(yh, xh, A)
        t1 = time.perf_counter()
        for _ in range(3):
            mpk.SpMV_CSR(yh, xh, A)
        t1 = (time.perf_counter() - t1) / 3
        out["pcie_inclusive"] = dict(gflops=round(2.0 * nnz_global / t1 / 1e9, 2), ms_per_call=round(t1 * 1e3, 3),
                                     note="mi_spmv with host vectors (pageable numpy): x in, y out over PCIe every call")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(p, c, v, x_host)
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        out["gpu_over_cpu_all_cores"] = round(value / out["cpu_baseline"]["all_cores"]["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
