"""bench.py --gpus N started plainly launches its ranks itself (a child torch.distributed.run), before anything touches the
GPU, and relays the child's exit code.  On this CPU-only box the ranks stop at "bench.py needs a GPU" — which is exactly what
shows that the launcher ran them."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check of the launcher (the GPU tests run the real thing)")
def test_gpus_n_self_launches_child_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs torch.distributed.run" not in r.stderr + r.stdout
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]  # both ranks were started and said so


@pytest.mark.gpu
def test_bench_gpus_2_on_one_card_prints_one_json_line():
    """`python3 bench.py --gpus 2` exactly as the driver starts it (no torchrun around it), two ranks sharing cuda:0 over gloo (the
    development arrangement): rc 0, ONE JSON line on stdout, every exchange probed, the chosen one bitwise."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MI355_FORCE_DEVICE="0", MI355_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "10", "--warmup", "3"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity"]["bitwise"] is True and d["scaling"] == "strong"
    h = d["halo"]
    assert set(h["exchanges"]) == {"push", "native", "allgather", "torch"} and h["exchanges"][h["chosen"]]["ok"]
    assert h["exchanges"]["push"]["ok"], h["exchanges"]["push"]           # separate processes: the IPC windows must come up
    assert h["exchange_bytes_per_step"]["sent_all_ranks"] > 0 and h["overlap"]["compute_only_us"] > 0
    assert h["torch_world"] == 2 and "rccl_ranks" in h
    # (round 5) the first real multi-GPU record must explain itself: seconds per candidate exchange, which kernels each one launches, why a
    # dropped one was dropped, progress lines on stderr from the moment the ranks are up
    assert set(h["probe_seconds"]) >= set(h["exchanges"]) and all(h["probe_seconds"][e] >= 0 for e in h["exchanges"])
    for e, pr in h["exchanges"].items():
        assert "probe_s" in pr and (pr["ok"] or len(pr["note"]) > 10), (e, pr)
        if pr["ok"]:
            assert pr["create_s"] >= 0 and pr["kernels"]["interior"] and pr["kernels"]["boundary"], (e, pr)
    if h["exchanges"]["push"]["form"].startswith("ONE launch"):  # the fused step names its kernel (spmv_sstream_fused<...> or the ring's FUSED form)
        assert h["exchanges"]["push"]["kernels"]["one_launch_step"], h["exchanges"]["push"]
    assert "[bench +" in r.stderr and "probing exchanges" in r.stderr and "chosen:" in r.stderr, r.stderr[-1500:]
    # the one-process form of the same workload (mi_dist_*), timed by a child of rank 0 behind the timed region
    sp = d["single_process"]
    assert sp["ok"], sp
    assert sp["parity"]["bitwise"] is True and sp["halo"]["exchange"] == "event" and sp["value"] > 0, sp


@pytest.mark.gpu
def test_bench_single_process_prints_one_json_line():
    """`python3 bench.py --gpus 3 --single-process`: one process, three ranks behind a mi_dist handle (all on cuda:0 here), k = 4 powers."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--single-process", "--workload", "c3", "--steps", "5", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["parity"]["bitwise"] is True and d["config"]["k"] == 4
    assert d["halo"]["exchange"] in ("event", "push", "rccl") and len(d["halo"]["ranks"]) == 3


def test_roofline_traffic_comes_from_a_committed_profile_of_the_same_kernel_family():
    """bench.py's roofline.traffic is the last PROFILED value (profiles/*_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the
    same command), looked up by workload and kernel.  The variants of a sliced kernel — prefetch depth, temporal / non-temporal values —
    read the same sliced copy, so a profile of one describes the others; a kernel of another family, or another workload, gets none."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t8, src8 = bench.load_traffic("c4", "spmv_sstream<8, true, 0>")
    t12, src12 = bench.load_traffic("c4", "spmv_sstream<12, false, 0>")
    assert src8 == src12 and src8.startswith("profiles/r") and src8.endswith("_bench_c4_pmc.json") and t8 == t12  # (the newest round's profile)
    prof = json.load(open(os.path.join(ROOT, src8)))
    assert prof["workload"] == "c4" and prof["kernel"].startswith("spmv_sstream<") and t8 == prof["hbm_bytes_per_launch"]
    assert 0.8 * prof["algorithmic_bytes_per_launch"] < t8 < 1.1 * prof["algorithmic_bytes_per_launch"]  # (10 B per nonzero read where the CSR model counts 12)
    # the gfx950 read correction is stated in the file the number comes from
    assert prof["fetch_correction"] == 2.0 and prof["hbm_bytes_per_launch"] == prof["hbm_read_bytes_per_launch"] + prof["hbm_write_bytes_per_launch"]
    tb, srcb = bench.load_traffic("fe_bcsr", "spmv_bcsr4_sell<8, true, 0, 2, 4>")
    assert srcb.endswith("_bench_fe_bcsr_pmc.json") and tb > 0
    assert bench.load_traffic("c4", "some_other_kernel<1>") == (None, None)
    assert bench.load_traffic("no_such_workload", "spmv_sstream<8, true, 0>") == (None, None)
