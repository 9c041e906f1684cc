#!/bin/bash
# 16-byte (paired) against 8-byte value loads (ring_pair.hpp): alternating processes, several handles and x / y pairs per process
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_AB_HANDLES=2 MI355_AB_XY=3 MI355_AB_ROUNDS=5
bash tools/ab_lib.sh "170 mesh" "5000000 s15" 2>&1 | tee gpurun_out/r3b_pair_ab4.txt
echo DONE
