#!/bin/bash
# A/B of two library builds (navierstokes_amd/csrc/ab_old/ against the tree's), tools/ab_lib.sh: the -m gpu parity file on the new build first,
# then alternating processes, several handles and x / y pairs per process
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3b_ab_tests.log 2>&1
rc=$?; tail -n 3 gpurun_out/r3b_ab_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
export MI355_AB_HANDLES=2 MI355_AB_XY=3 MI355_AB_ROUNDS=${MI355_AB_ROUNDS:-4}
bash tools/ab_lib.sh "5000000 s15" "170 mesh" "1000000 s15" "5000000 svar" 2>&1 | tee gpurun_out/r3b_ab.txt
echo DONE
