"""The N>1 path end to end on the CPU: world_size 2 (and 3) over gloo, real
torch.distributed exchange, product planner, oracle-injected local compute."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def run(world, kind, n, w, port):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), kind, str(n), str(w)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIST_RESULT")]
    assert line and "ok=1" in line[0], r.stdout[-2000:]
    return line[0]


@pytest.mark.parametrize("world,kind,n,w,port", [(2, "s15", 4000, 300, 29601), (2, "sfe", 2000, 400, 29602),
                                                  (3, "svar", 3000, 2000, 29603)])
def test_dist_spmv_gloo(world, kind, n, w, port):
    line = run(world, kind, n, w, port)
    assert f"world={world}" in line and "halo=0" not in line
