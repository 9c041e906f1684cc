#!/bin/bash
# round 3 measurement set, part 2: rocprofv3 kernel traces and PMC passes (separate runs; --pmc never combined with traces other than kernel-trace)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
prof() { # name, extra env..., -- bench args
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
prof r3prof_c4 --steps 50 --warmup 5 $COMMON
prof r3prof_c3 --workload c3 --steps 50 --warmup 5 $COMMON
prof r3prof_c2 --workload c2 --steps 50 --warmup 5 $COMMON
prof r3prof_fe_spmm4 --workload fe_spmm4 --steps 30 --warmup 5 $COMMON
prof r3prof_mesh_perm_internal --workload mesh_perm --internal --steps 30 --warmup 5 $COMMON
prof r3prof_c4_pipeline --steps 5 --warmup 2 --no-cpu-baseline --no-parity
export MI355_SPMV_KERNEL=ring MI355_RING_NT=1
for c in FETCH_SIZE WRITE_SIZE; do pmc r3pmc_c4 $c --steps 10 --warmup 2 $COMMON; done
unset MI355_RING_NT
export MI355_SPMK_FUSED=1
for c in FETCH_SIZE WRITE_SIZE; do pmc r3pmc_c3 $c --workload c3 --steps 10 --warmup 2 $COMMON; done
unset MI355_SPMK_FUSED MI355_SPMV_KERNEL
export MI355_SPMM_TILE=1
for c in FETCH_SIZE WRITE_SIZE; do pmc r3pmc_fe_spmm4 $c --workload fe_spmm4 --steps 10 --warmup 2 $COMMON; done
unset MI355_SPMM_TILE
for f in gpurun_out/r3pmc_*.txt; do echo "== $f"; cat $f | head -40; done
for f in gpurun_out/r3prof_*_kernel_stats.csv; do echo "== $f"; head -6 $f; done
echo R3_PART2_DONE
