"""Host-side planners under the address and undefined-behaviour sanitizers (CPU only: GPU sanitizers are not available on this pool, and
nothing here launches a kernel).  The harnesses live under tools/ (plan_asan.hip, part_asan.cpp) and say what they check."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "navierstokes_amd", "csrc")


def _run(cmd, exe, tmp_path):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


def test_partition_planner_under_host_sanitizers(tmp_path):
    """partition.hpp: PartPlan::build, build_combined, build_all_ext (the [owned | halo] piece of the staged one-launch step, round 5) on random
    banded / multi-band / node-blocked patterns cut into 1-6 ranks; every nonzero found again under the caller's order, the 4x4 structure kept."""
    if not shutil.which("g++"):
        pytest.skip("g++ not found")
    exe = str(tmp_path / "part_asan")
    out = _run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-I" + CSRC, "-o", exe, os.path.join(ROOT, "tools", "part_asan.cpp")], exe, tmp_path)
    assert "bad 0" in out, out


def test_sliced_stream_planners_under_host_sanitizers(tmp_path):
    """spmv_sstream.hpp / spmv_sstream_mw.hpp: both planners and their host replays on random multi-band patterns, both row shifts."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = str(tmp_path / "plan_asan")
    out = _run([hipcc, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-I" + CSRC,
                "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tools", "plan_asan.hip")], exe, tmp_path)
    assert "bad 0" in out and "eligible" in out, out
