#!/bin/bash
# tools/gpu_run.sh — what one gpurun call executes on the MI355X box (development helper).
# Each step writes under gpurun_out/; a step that times out (rc>=124) stops the script.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"
  tail -n 40 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
  return 0
}
for s in "$@"; do
  case $s in
    rocsparse) step rocsparse_c4 300 ./tools/rocsparse_cmp 0 5000000 && step rocsparse_c2 200 ./tools/rocsparse_cmp 0 1000000 && step rocsparse_sfe 300 ./tools/rocsparse_cmp 2 1400000 ;;
    tests)    step tests 600 python -m pytest tests -x -q -m gpu ;;
    smoke)    step smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)    step bench 400 python bench.py ;;
    bench_c2) step bench_c2 300 python bench.py --workload c2 --no-cpu-baseline ;;
    bench_c3) step bench_c3 300 python bench.py --workload c3 --no-cpu-baseline ;;
    bench_fe) step bench_fe 300 python bench.py --workload fe --no-cpu-baseline ;;
    bench_fe_bcsr) step bench_fe_bcsr 300 python bench.py --workload fe_bcsr --no-cpu-baseline ;;
    bench_cold) step bench_cold 400 python bench.py --cold --steps 30 --warmup 3 --no-cpu-baseline ;;
    bench_gloo3) MI355_FORCE_DEVICE=0 MI355_BENCH_BACKEND=gloo step bench_gloo3 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 3 --workload c3 --steps 3 --warmup 1 --no-cpu-baseline ;;
    # ---- round 2 profile set: the SAME kernel as the unprofiled bench by construction (ring, non-temporal values forced:
    #      under a counter pass every launch takes ~2x as long and the create-time measurement would compare candidates in that regime)
    r2prof)   rm -rf gpurun_out/r2prof; step r2prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2prof -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-extras ;;
    r2prof_cold) rm -rf gpurun_out/r2prof_cold; step r2prof_cold 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2prof_cold -- python bench.py --cold --steps 30 --warmup 3 --no-cpu-baseline --no-parity ;;
    r2pmc)    for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/r2pmc_$c; MI355_SPMV_KERNEL=ring MI355_RING_NT=1 step r2pmc_$c 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r2pmc_$c -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras || exit 1; done ;;
    r2pmc_cold) for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/r2pmc_cold_$c; MI355_SPMV_KERNEL=ring MI355_RING_NT=1 step r2pmc_cold_$c 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r2pmc_cold_$c -- python bench.py --cold --steps 10 --warmup 2 --no-cpu-baseline --no-parity || exit 1; done ;;
    prof)     rm -rf gpurun_out/prof; step prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-parity ;;
    pmc_fetch) rm -rf gpurun_out/pmc_fetch; MI355_SPMV_AUTOTUNE=0 step pmc_fetch 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity ;;
    pmc_write) rm -rf gpurun_out/pmc_write; MI355_SPMV_AUTOTUNE=0 step pmc_write 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity ;;
    *) echo "unknown step $s"; exit 1 ;;
  esac
done
echo "=== done ($(date +%T))"
