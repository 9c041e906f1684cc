#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf gpurun_out/ppmc_$i
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/ppmc_$i -- python3 tools/placement_pmc.py 8 8 > gpurun_out/ppmc_$i.out 2> gpurun_out/ppmc_$i.err
  rc=$?; echo "pass $i rc=$rc"; [ $rc -ge 124 ] && exit $rc
  python3 tools/placement_pmc_read.py gpurun_out/ppmc_$i 8
done
