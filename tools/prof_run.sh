#!/bin/bash
# rocprofv3 kernel traces and PMC passes of the bench command (separate runs; --pmc only ever beside nothing else), rounds 4+.
#   bash tools/prof_run.sh <workload> [<workload> ...]      e.g. c4 fe_bcsr fe c3     (then tools/prof_collect.py per workload -> profiles/rNN_bench_*)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
COMMON="--no-cpu-baseline --no-parity --no-extras"
for W in "$@"; do
  name=prof_$W
  rm -rf gpurun_out/$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python3 bench.py --workload $W --steps 50 --warmup 5 $COMMON > gpurun_out/$name.out 2> gpurun_out/$name.err
  rc=$?; echo "$name rc=$rc"; [ $rc -ge 124 ] && exit $rc
  find gpurun_out/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${name}_kernel_stats.csv
  head -4 gpurun_out/${name}_kernel_stats.csv | cut -c1-240
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${W}_$c
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${W}_$c -- python3 bench.py --workload $W --steps 10 --warmup 2 $COMMON > gpurun_out/pmc_${W}_$c.out 2> gpurun_out/pmc_${W}_$c.err
    rc=$?; echo "pmc_${W}_$c rc=$rc"; [ $rc -ge 124 ] && exit $rc
    python tools/pmc_summary.py gpurun_out/pmc_${W}_$c > gpurun_out/pmc_${W}_$c.txt 2>&1
    rm -rf gpurun_out/pmc_${W}_$c   # (the raw per-dispatch files are large; the summary stays)
  done
  grep -h '"metric"' gpurun_out/$name.out | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('BENCH-UNDER-PROFILER', d['config']['name'], d['roofline']['launch_us'], d['roofline']['kernel'])"
  rm -rf gpurun_out/$name
done
for f in gpurun_out/pmc_*.txt; do echo "== $f"; grep -B1 -A3 -E "spmv_|spmk_|spmm_" $f | head -40; done
echo PROF_DONE
