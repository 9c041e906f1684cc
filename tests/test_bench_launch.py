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
