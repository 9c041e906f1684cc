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
    kbench)   step kbench 300 ./tools/kbench 0 5000000 2000 5 50 ;;
    rocsparse) step rocsparse_c4 300 ./tools/rocsparse_cmp 0 5000000 && step rocsparse_c2 200 ./tools/rocsparse_cmp 0 1000000 && step rocsparse_sfe 300 ./tools/rocsparse_cmp 2 1400000 ;;
    kbench_c2) step kbench_c2 200 ./tools/kbench 0 1000000 2000 5 100 ;;
    kbench_sfe) step kbench_sfe 300 ./tools/kbench 2 1400000 2000 5 50 ;;
    kbench_svar) step kbench_svar 300 ./tools/kbench 1 5000000 2000 5 50 ;;
    kprof)    # PMC passes over a few kbench variants (KFILTER env), one rocprofv3 run per counter group
              F="${KFILTER:-stream<2048>,E4 ring<512,2048,5120> 512,E7 ring2<512,2048,5120,D3> 512,NO GATHER}"
              rm -rf gpurun_out/kprof; mkdir -p gpurun_out/kprof
              step kprof_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof/trace -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              step kprof_sq1 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d gpurun_out/kprof/sq1 -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              step kprof_sq2 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/kprof/sq2 -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              step kprof_tc 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/kprof/tc -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              step kprof_fetch 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/kprof/fetch -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              ;;
    kprof2)   F="${KFILTER:-ABL skeleton no y stores,ABL ring5a no reduce+gather+stage}"
              rm -rf gpurun_out/kprof2; mkdir -p gpurun_out/kprof2
              step kprof2_a 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/kprof2/a -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              step kprof2_b 300 rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/kprof2/b -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              step kprof2_c 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d gpurun_out/kprof2/c -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              step kprof2_d 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_TAG_STALL_sum --output-format csv -d gpurun_out/kprof2/d -- ./tools/kbench 0 5000000 2000 1 3 "$F"
              ;;
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
