#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3b_tests.log 2>&1
rc=$?; tail -n 6 gpurun_out/r3b_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3b_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/r3b_smoke.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3b_bench_c4.json 2> gpurun_out/r3b_bench_c4.err; rc=$?; echo "bench rc=$rc"
echo DONE
