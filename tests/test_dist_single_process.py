"""mi_dist_* (include/mi355_spmv.h, navierstokes_amd/csrc/capi_dist.hip): the multi-GPU split behind ONE handle of ONE process —
what the reference's single-process harnesses reach (mpk/2SpMV.cpp:128-141, mpk/SpMVmulti0.cpp:369-411 over the seam
mpk/SpMV.h:52-66) when the shim is told MI355_NGPUS=N.  On the one-GPU box every rank maps to device 0 and the event exchange
drives the step; the RCCL form runs over tests/fake_rccl (ranks as worker threads of one process, as mi_dist makes them)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = os.path.join(ROOT, "tests", "dist_single_worker.py")
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def run_worker(mode, N, extra_env, timeout=600):
    env = dict(os.environ)
    env.update(extra_env)
    r = subprocess.run([sys.executable, WORKER, mode, str(N)], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and "DIST_SINGLE_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 3, 4, 8])
def test_one_handle_over_n_ranks_event_exchange(N):
    """Products, k = 4 powers, dot and orthogonalize over N ranks of one process, host and device-resident vectors, 40 steps back
    to back: bitwise against the oracle (reductions inside their bound; the update bit-equal given beta)."""
    out = run_worker("capi", N, {"MI355_DIST_EXCHANGE": "event"})
    assert out.count("exchange=event") == 5, out


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 4])
def test_one_handle_over_n_ranks_rccl_exchange(N):
    """The RCCL form of the step (mi_part_comm_init + mi_part_spmv_dev per worker thread) with the in-process librccl stand-in."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl not built")
    out = run_worker("capi", N, {"MI355_DIST_EXCHANGE": "rccl", "MI355_RCCL_LIBRARY": FAKE, "MI355_PART_EXCHANGE": "sendrecv"})
    assert out.count("exchange=rccl ") == 5, out
    # ... and its all-gather form (one ncclAllGather of every rank's boundary slice), forced for every matrix
    out = run_worker("capi", N, {"MI355_DIST_EXCHANGE": "rccl", "MI355_RCCL_LIBRARY": FAKE, "MI355_PART_EXCHANGE": "allgather"})
    assert out.count("exchange=rccl-allgather") == 5, out


@pytest.mark.gpu
def test_push_is_refused_for_ranks_sharing_a_device():
    """The spinning push form needs one device (one set of hardware queues) per rank: forced on a one-GPU box the create fails
    loudly instead of risking the in-process deadlock of push_exchange.hpp; `auto` falls back to the event exchange and says why."""
    from navierstokes_amd import mpk, synth
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    p, c, v = synth.rows("s15", 50_000, w=500)
    os.environ["MI355_DIST_EXCHANGE"] = "push"
    try:
        with pytest.raises(mpk.MiError) as e:
            mpk.DistMatrix(2, 50_000, p, c, v)
        assert "share a device" in str(e.value)
    finally:
        del os.environ["MI355_DIST_EXCHANGE"]
    D = mpk.DistMatrix(2, 50_000, p, c, v)
    info = D.info()
    assert info["exchange"] == "event" and "ranks share devices" in info["note"], info
    D.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 4])
def test_shim_symbols_with_MI355_NGPUS(N):
    """SpMV_CSR*, SpM2V_CSR*, SpM3V, SpM4V*, orthogonalize of include/SpMV.h — the C++ symbols a reference driver calls — routed to
    a mi_dist handle by MI355_NGPUS=N: bitwise vs the reference-made goldens (VERDICT r3 next #1 done-criterion (a))."""
    run_worker("shim", N, {"MI355_NGPUS": str(N)})


@pytest.mark.gpu
def test_reference_2spmv_main_with_MI355_NGPUS(tmp_path):
    """The reference's own mpk/2SpMV.cpp main (oracle/_ref/2spmv_mi355: compiled where it lies, linked against the shim), unmodified,
    with MI355_NGPUS=2: its CSR variants run on two ranks of one process (done-criterion (b))."""
    from navierstokes_amd import synth
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
    env = dict(os.environ, MI355_NGPUS="2")
    r = subprocess.run([exe, str(mtx)], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    errs = [float(m) for m in re.findall(r"rel err = ([0-9.eE+-]+)", r.stdout)]
    assert len(errs) == 7 and max(errs) <= 1e-15, r.stdout


def test_dist_symbols_are_exported():
    """CPU: the mi_dist_* entry points of include/mi355_spmv.h load, and without a device create fails with MI_ERR_NODEVICE."""
    from navierstokes_amd import mpk
    import ctypes
    L = mpk.lib()
    for name in ("mi_dist_create", "mi_dist_destroy", "mi_dist_spmv", "mi_dist_spmk", "mi_dist_spmv_dev", "mi_dist_spmk_dev",
                 "mi_dist_synchronize", "mi_dist_dot", "mi_dist_orthogonalize", "mi_dist_vec_create", "mi_dist_update_values"):
        assert hasattr(L, name), name
    import torch
    if not torch.cuda.is_available():
        p = np.array([0, 1, 2], np.int32)
        c = np.array([0, 1], np.int32)
        v = np.ones(2)
        h = ctypes.c_void_p()
        rc = L.mi_dist_create(2, 2, p.ctypes.data, c.ctypes.data, v.ctypes.data, ctypes.byref(h))
        assert rc == 2, (rc, L.mi_last_error())  # MI_ERR_NODEVICE: no CPU fallback
