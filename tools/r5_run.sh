#!/bin/bash
# tools/r5_run.sh — steps of one gpurun call (round 5).  Each step logs under gpurun_out/; a step that times out or is killed
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
  tail -n "${TAILN:-12}" "gpurun_out/$name.log" | cut -c1-1800
  if [ $rc -ge 124 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
  return 0
}
for s in "$@"; do
  case $s in
    sim)      step r5_sim8 300 python tools/sim_rank.py 8 1 && step r5_sim4 300 python tools/sim_rank.py 4 1 && step r5_sim2 300 python tools/sim_rank.py 2 1 ;;
    sim8prof) rm -rf gpurun_out/r5_sim8prof; step r5_sim8prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_sim8prof -- python tools/sim_rank.py 8 1 && (find gpurun_out/r5_sim8prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5_sim8_kernel_stats.csv; rm -rf gpurun_out/r5_sim8prof; head -12 gpurun_out/r5_sim8_kernel_stats.csv | cut -c1-200) ;;
    quick)    step r5_quick_a 200 python tools/quick_ss.py 625000 && QUICK_ROW0=625000 step r5_quick_b 200 python tools/quick_ss.py 625000 && step r5_trace_r8 200 ./tools/sstream_trace 625000 ;;
    quick_off) for o in 0 16 1722 1723; do QUICK_OFFSET=$o step r5_quick_off$o 200 python tools/quick_ss.py 625000 || exit 1; done ;;
    sp2)      MI355_DIST_DEVICES=0 step r5_bench_sp2 300 python bench.py --gpus 2 --single-process --workload c2 --steps 200 --warmup 20 ;;
    sp2prof)  rm -rf gpurun_out/r5_sp2prof; MI355_DIST_DEVICES=0 step r5_sp2prof 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_sp2prof -- python bench.py --gpus 2 --single-process --workload c2 --steps 30 --warmup 5 --no-parity && (f=$(find gpurun_out/r5_sp2prof -name "*kernel_trace.csv" | head -1); python tools/trace_tail.py "$f" 60 > gpurun_out/r5_sp2_trace_tail.txt; rm -rf gpurun_out/r5_sp2prof; cat gpurun_out/r5_sp2_trace_tail.txt | cut -c1-200) ;;
    spmm)     step r5_bench_spmm8 300 python bench.py --workload fe_spmm8 --no-cpu-baseline && MI355_SPMM_TILE_SORT=0 step r5_bench_spmm8_unsorted 300 python bench.py --workload fe_spmm8 --no-cpu-baseline && step r5_bench_spmm4 300 python bench.py --workload fe_spmm4 --no-cpu-baseline && step r5_t_spmm 600 python -m pytest tests/test_spmm_gpu.py -x -q -m gpu ;;
    c3tcc)    for c in TCC_HIT_sum TCC_MISS_sum; do rm -rf gpurun_out/r5_c3_$c; step r5_c3_$c 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r5_c3_$c -- python3 bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras || exit 1; python tools/pmc_summary.py gpurun_out/r5_c3_$c > gpurun_out/r5_c3_$c.txt 2>&1; rm -rf gpurun_out/r5_c3_$c; grep -B1 -A3 -E "spmk_|spmv_sstream|spmv_csr_ring" gpurun_out/r5_c3_$c.txt | head -30; done ;;
    c2forms)  for f in 0 1 2 3; do MI355_SSTREAM_FORM=$f step r5_bench_c2_form$f 300 python bench.py --workload c2 --kernel sstream --no-cpu-baseline || exit 1; done ;;
    t_push)   step r5_t_push 900 python -m pytest tests/test_gpu_parity.py tests/test_dist_single_process.py tests/test_bench_launch.py -x -q -m gpu -k "ranks_sharing_one_card or native_step or bench_gpus_2 or single_process" ;;
    fuzz)     step r5_fuzz 1100 python tools/gpu_fuzz.py ${FUZZ_FIRST:-600} ${FUZZ_LAST:-640} ;;
    spmm_ucap) for u in "368,190" "368,150" "368,300"; do MI355_SPMM_TILE_UCAP=$u MI355_SPMM_TILE=3 step "r5_bench_spmm8_ucap_${u#*,}" 300 python bench.py --workload fe_spmm8 --no-cpu-baseline --no-extras || exit 1; done; MI355_SPMM_TILE=3 step r5_bench_spmm8_ucap_256 300 python bench.py --workload fe_spmm8 --no-cpu-baseline --no-extras ;;
    t_upd)    step r5_t_upd 900 python -m pytest tests/test_gpu_parity.py tests/test_reorder_gpu.py tests/test_spmm_gpu.py tests/test_shim_gpu.py -x -q -m gpu -k "update or refresh or graph_replay or sliced_copy or relabelled or reorder or column_major or fe_matrix or blocked" && for w in fe fe_bcsr fe_perm; do step r5_bench_upd_$w 300 python bench.py --workload $w --no-cpu-baseline $( [ $w = mesh_perm ] && echo --internal ) || exit 1; done ;;
    leak)     step r5_leak 600 python tools/leak_check.py ;;
    simfe_d)  for d in 22 23 24 27; do MI355_PUSH_EXT_DEPTH=$d SIM_RANK_EXT_PARTS=${PARTS:-0} step r5_simfe8_d$d 300 python tools/sim_rank.py 8 1 fe || exit 1; done; for l in 0 2; do MI355_PUSH_EXT_LANES16=$l SIM_RANK_EXT_PARTS=0 step r5_simfe8_l$l 300 python tools/sim_rank.py 8 1 fe || exit 1; done ;;
    simfe_w)  for w in 4 8 12 19 38; do MI355_PUSH_EXT_WGS=$w SIM_RANK_EXT_PARTS=0 step r5_simfe8_w$w 300 python tools/sim_rank.py 8 1 fe || exit 1; done ;;
    simfe)    step r5_simfe8 300 python tools/sim_rank.py 8 1 fe && MI355_PUSH_EXT_SPLIT=1 SIM_RANK_EXT_PARTS=0 step r5_simfe8_split 300 python tools/sim_rank.py 8 1 fe && MI355_PUSH_FUSED_EXT=0 step r5_simfe8_four 300 python tools/sim_rank.py 8 1 fe ;;
    fe4)      MI355_FORCE_DEVICE=0 MI355_BENCH_BACKEND=gloo step r5_bench_fe4 500 python bench.py --gpus 4 --workload fe --steps 50 --warmup 5 --no-cpu-baseline ;;
    permprof) for w in c2_perm fe_perm; do rm -rf gpurun_out/r5_pp_$w; step r5_pp_$w 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_pp_$w -- python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-extras || exit 1; find gpurun_out/r5_pp_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r5_pp_${w}_kernel_stats.csv; rm -rf gpurun_out/r5_pp_$w; head -8 gpurun_out/r5_pp_${w}_kernel_stats.csv | cut -c1-220; done ;;
    simfe_small) for c in 40 30; do SIM_FE_CELLS=$c SIM_RANK_EXT_PARTS=0 step r5_simfe_c${c} 300 python tools/sim_rank.py 4 1 fe || exit 1; done ;;
    dfuzz)    step r5_dist_fuzz 1100 python tools/dist_fuzz.py ${FUZZ_FIRST:-1} ${FUZZ_LAST:-40} ;;
    b_mw)     step r5_bench_mesh_ss 300 python bench.py --workload mesh --kernel sstream --no-cpu-baseline --no-extras ;;
    b_mw_small) step r5_bench_mesh_small_ss 300 python bench.py --workload mesh_small --kernel sstream --no-cpu-baseline --no-extras ;;
    t_mw)     step r5_t_mw 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cut_ring or sliced_stream_kernel" && step r5_bench_mesh 300 python bench.py --workload mesh --no-cpu-baseline && step r5_bench_mesh_ss 300 python bench.py --workload mesh --kernel sstream --no-cpu-baseline --no-extras && step r5_bench_mesh_small 300 python bench.py --workload mesh_small --no-cpu-baseline --no-extras ;;
    simmesh)  for n in 8 4; do step r5_simmesh$n 400 python tools/sim_rank.py $n 1 mesh || exit 1; done ;;
    simfe_n)  for n in 8 4 2; do step r5_simfe$n 300 python tools/sim_rank.py $n 1 fe && MI355_PUSH_FUSED_EXT=0 step r5_simfe${n}_four 300 python tools/sim_rank.py $n 1 fe || exit 1; done ;;
    t_ext)    step r5_t_ext 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ranks_sharing_one_card and sfe" ;;
    t_meshd)  step r5_t_meshd 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ranks_sharing_one_card and mesh" ;;
    sim8)     step r5_sim8 300 python tools/sim_rank.py 8 1 ;;
    tests)    step r5_tests 1100 python -m pytest tests -x -q -m gpu ;;
    smoke)    step r5_smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)    step r5_bench_c4 300 python bench.py ;;
    bench_c2) step r5_bench_c2 200 python bench.py --workload c2 --no-cpu-baseline ;;
    bench_c3) step r5_bench_c3 200 python bench.py --workload c3 --no-cpu-baseline ;;
    trace)    step r5_trace_c2 200 ./tools/sstream_trace 1000000 && step r5_trace_r8 200 ./tools/sstream_trace 625000 ;;
    t_ss)     step r5_t_ss 1100 python -m pytest tests/test_gpu_parity.py tests/test_reorder_gpu.py -x -q -m gpu -k "sliced_stream or golden or graph_replay or sliced_copy or ranks_sharing_one_card or native_step or relabelled or pipeline or random_patterns" ;;
    t_new)    step r5_t_new 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or graph_replay or sliced_stream or sliced_copy" ;;
    *) echo "unknown step $s"; exit 2 ;;
  esac
done
