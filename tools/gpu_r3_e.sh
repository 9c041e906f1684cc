#!/bin/bash
# round 3, GPU pass E: fuzz of the round-3 paths
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "round3_paths" > gpurun_out/r3e_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3e_pytest.log
timeout -k 10 900 python tools/gpu_fuzz.py 0 40 > gpurun_out/r3e_fuzz.txt 2>&1; echo "fuzz rc=$?"
tail -6 gpurun_out/r3e_fuzz.txt
grep -c spmk1 gpurun_out/r3e_fuzz.txt; grep -c dotE gpurun_out/r3e_fuzz.txt
