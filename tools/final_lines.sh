#!/bin/bash
# every bench line README / DESIGN quote, one box, one after the other -> gpurun_out/final_lines.jsonl (committed as profiles/rNN_bench_lines.jsonl)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/final_lines.jsonl; : > $out
run() { # args of bench.py
  timeout -k 10 400 python3 bench.py "$@" > gpurun_out/final_one.log 2>&1
  rc=$?
  grep -h '^{"metric"' gpurun_out/final_one.log >> $out
  echo "bench.py $* -> rc=$rc ($(grep -c . $out) lines)"
  [ $rc -ge 124 ] && exit $rc
  return 0
}
run
run --workload c2
run --workload c3
run --workload fe
run --workload fe_bcsr
run --workload fe_spmm4
run --workload fe_spmm8
run --workload mesh
run --workload c2_perm
run --workload c2_perm --internal
run --workload fe_perm
run --workload mesh_perm --internal
run --cold --steps 30 --warmup 3 --no-cpu-baseline
echo FINAL_DONE
