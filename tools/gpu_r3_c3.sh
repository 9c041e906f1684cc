#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export MI355_SPMK_NOACQ=1
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "one_launch" > gpurun_out/r3c_pytest.log 2>&1; echo "pytest (noacq) rc=$?" | tee -a gpurun_out/r3c_pytest.log
tail -5 gpurun_out/r3c_pytest.log
for args in "1000000 4 300" "1000000 8 150" "300000 4 300" "5000000 3 60" "100000 8 300"; do
  timeout -k 10 300 python tools/spmk_stress.py $args 2>&1 | tail -4
done
