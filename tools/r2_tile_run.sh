#!/bin/bash
# one gpurun call: parity of the tile kernel / new ring plans, then the wide-band workloads (development helper)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_reorder_gpu.py -x -q -m gpu > gpurun_out/t_tests.log 2>&1
rc=$?; tail -n 15 gpurun_out/t_tests.log; echo "tests rc=$rc"
[ $rc -ge 124 ] && exit $rc
[ $rc -ne 0 ] && exit $rc
run() { # tag, env..., -- workload
  local tag=$1; shift
  timeout -k 10 400 env "$@" > gpurun_out/t_bench_$tag.log 2>&1
  rc=$?; echo "bench $tag rc=$rc"
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(f"gpurun_out/t_bench_{tag}.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; k = d.get("kernel_info", {})
        print(tag, "launch_us", r["launch_us"], "frac", r["frac"], r["kernel"], "cold", r.get("cold_single_shot", {}).get("launch_us"),
              "tune", k.get("autotune_us"), "tile", k.get("tile_plan"), "reorder", d.get("reorder"), "bitwise", d.get("parity", {}).get("bitwise"))
PY
  [ $rc -ge 124 ] && exit $rc
}
B="python bench.py --no-cpu-baseline --workload"
run msp_xmap    MI355_X=1 $B mesh_small_perm
run msp_noxmap  MI355_TILE_XMAP=0 $B mesh_small_perm
run msp_1024    MI355_TILE_NNZB=1024 $B mesh_small_perm
run ms_1024     MI355_TILE_NNZB=1024 $B mesh_small
run mp_xmap     MI355_X=1 $B mesh_perm
run mp_1024     MI355_TILE_NNZB=1024 $B mesh_perm
run m_1024      MI355_TILE_NNZB=1024 $B mesh
echo TILE_RUN_DONE
