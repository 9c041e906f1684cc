#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_reorder_gpu.py -x -q -m gpu > gpurun_out/t_tests.log 2>&1
rc=$?; tail -n 15 gpurun_out/t_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
run() { # tag, env..., cmd
  local tag=$1; shift
  timeout -k 10 400 env "$@" > gpurun_out/t_bench_$tag.log 2>&1
  rc=$?; echo "bench $tag rc=$rc"
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(f"gpurun_out/t_bench_{tag}.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; k = d.get("kernel_info", {})
        print(tag, "launch_us", r["launch_us"], "frac", r["frac"], r["kernel"], "cold", r.get("cold_single_shot", {}).get("launch_us"), "box", r.get("this_box_stream_read", {}).get("gbs"),
              "tune", k.get("autotune_us"), "mring", k.get("mring_plan"), "reorder", d.get("reorder"), "bitwise", d.get("parity", {}).get("bitwise"))
PY
  [ $rc -ge 124 ] && exit $rc
}
B="python bench.py --no-cpu-baseline --workload"
run mesh_small X=1 $B mesh_small
run mesh_small_perm X=1 $B mesh_small_perm
run mesh X=1 $B mesh
run mesh_perm X=1 $B mesh_perm
echo MRING_RUN_DONE
