#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
export MI355_SPMM_TILE=1
for dbg in 0 4 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/r3x_${dbg}_$c
    MI355_SPMM_DBG=$dbg timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r3x_${dbg}_$c -- python3 bench.py --workload fe_spmm4 --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras > /dev/null 2>&1
    echo "dbg=$dbg $c:"; python tools/pmc_summary.py gpurun_out/r3x_${dbg}_$c | grep -A1 "spmm_bcsr4_tile"
  done
done
