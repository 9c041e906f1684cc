#!/bin/bash
# tools/r4_run.sh — steps of one gpurun call (round 4).  Each step logs under gpurun_out/; a step that times out or is killed
# (rc >= 124) ends the script: no further GPU step is started behind it.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"
  tail -n "${TAILN:-12}" "gpurun_out/$name.log" | cut -c1-1500
  if [ $rc -ge 124 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
  return 0
}
for s in "$@"; do
  case $s in
    dist)     step r4_dist 600 python -m pytest tests/test_dist_single_process.py -x -q -m gpu ;;
    tests)    step r4_tests 1000 python -m pytest tests -x -q -m gpu ;;
    bench)    step r4_bench_c4 200 python bench.py ;;
    bench_fe) step r4_bench_fe 200 python bench.py --workload fe && step r4_bench_fe_bcsr 200 python bench.py --workload fe_bcsr ;;
    bench_c3) step r4_bench_c3 200 python bench.py --workload c3 ;;
    bench_sp) MI355_DIST_DEVICES=0 step r4_bench_sp2 300 python bench.py --gpus 2 --single-process --workload c2 --steps 50 --warmup 5 ;;
    bench_mp) MI355_FORCE_DEVICE=0 MI355_BENCH_BACKEND=gloo step r4_bench_mp2 400 python bench.py --gpus 2 --workload c2 --steps 50 --warmup 5 ;;
    sell)     step r4_sell 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sliced_copy or bcsr or fe_matrix" ;;
    ag)       step r4_ag 900 python -m pytest tests/test_gpu_parity.py tests/test_dist_single_process.py tests/test_bench_launch.py -x -q -m gpu -k "allgather or rccl_exchange or multirank_threads or bench_gpus_2" ;;
    spmm)     step r4_spmm 900 python -m pytest tests/test_spmm_gpu.py -x -q -m gpu && step r4_bench_spmm4 200 python bench.py --workload fe_spmm4 && step r4_bench_spmm8 200 python bench.py --workload fe_spmm8 ;;
    sstream)  step r4_sstream 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sliced_stream" && step r4_bench_c4 200 python bench.py && step r4_bench_c2 200 python bench.py --workload c2 && step r4_bench_c3 200 python bench.py --workload c3 ;;
    dot)      step r4_dot 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dot_in_its_epilogue or pipeline or random_patterns" && step r4_bench_c4 300 python bench.py ;;
    perm)     step r4_perm 600 python -m pytest tests/test_reorder_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "relabelled or reorder or sliced_stream or permuted or scrambled" && step r4_bench_c2_perm 200 python bench.py --workload c2_perm && step r4_bench_fe_perm 200 python bench.py --workload fe_perm && step r4_bench_mesh_perm 300 python bench.py --workload mesh_perm && step r4_bench_c2_perm_int 200 python bench.py --workload c2_perm --internal ;;
    *) echo "unknown step $s"; exit 2 ;;
  esac
done
