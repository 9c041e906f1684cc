#!/bin/bash
# round 3, second session (16-byte value loads, unaligned blocks for large matrices): rocprofv3 kernel traces and PMC passes of the bench
# command (separate runs; --pmc never combined with traces other than kernel-trace), then one rank's share at N = 8 / 4 / 2
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
prof() { # name -- bench args
  local name=$1; shift
  rm -rf gpurun_out/$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python3 bench.py "$@" > gpurun_out/$name.out 2> gpurun_out/$name.err
  local rc=$?; echo "$name rc=$rc"; [ $rc -ge 124 ] && exit $rc
  find gpurun_out/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${name}_kernel_stats.csv
}
pmc() { # name counter -- bench args
  local name=$1 ctr=$2; shift 2
  rm -rf gpurun_out/${name}_$ctr
  timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/${name}_$ctr -- python3 bench.py "$@" > gpurun_out/${name}_$ctr.out 2> gpurun_out/${name}_$ctr.err
  local rc=$?; echo "${name}_$ctr rc=$rc"; [ $rc -ge 124 ] && exit $rc
  python tools/pmc_summary.py gpurun_out/${name}_$ctr > gpurun_out/${name}_$ctr.txt 2>&1
}
COMMON="--no-cpu-baseline --no-parity --no-extras"
prof r3bprof_c4 --steps 50 --warmup 5 $COMMON
prof r3bprof_c2 --workload c2 --steps 50 --warmup 5 $COMMON
prof r3bprof_c3 --workload c3 --steps 50 --warmup 5 $COMMON
prof r3bprof_mesh --workload mesh --steps 50 --warmup 5 $COMMON
export MI355_SPMV_KERNEL=ring MI355_RING_NT=1
for c in FETCH_SIZE WRITE_SIZE; do pmc r3bpmc_c4 $c --steps 10 --warmup 2 $COMMON; done
unset MI355_RING_NT MI355_SPMV_KERNEL
export MI355_SPMV_KERNEL=mring
for c in FETCH_SIZE WRITE_SIZE; do pmc r3bpmc_mesh $c --workload mesh --steps 10 --warmup 2 $COMMON; done
unset MI355_SPMV_KERNEL
for f in gpurun_out/r3bpmc_*.txt; do echo "== $f"; head -30 $f; done
for f in gpurun_out/r3bprof_*_kernel_stats.csv; do echo "== $f"; head -4 $f | cut -c1-260; done
grep -h '"metric"' gpurun_out/r3bprof_*.out | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['config']['name'], d['roofline']['launch_us'], d['roofline']['kernel'][:60])"
for N in 8 4 2; do timeout -k 10 300 python tools/sim_rank.py $N 1 2>&1 | tail -n 12; done | tee gpurun_out/r3b_sim_rank.txt
echo R3B_PROF_DONE
