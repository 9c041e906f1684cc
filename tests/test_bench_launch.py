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
