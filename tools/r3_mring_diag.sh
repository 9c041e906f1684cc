#!/bin/bash
# round 3: why is the relabelled 5 M-row mesh slower than the natural order under the same multi-window ring kernel?
# PMC passes (one counter set per run, no traces) on `mesh` and `mesh_perm --internal`, and the run-dealing A/B.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
COMMON="--no-cpu-baseline --no-parity --no-extras --steps 10 --warmup 2"
export MI355_SPMV_KERNEL=mring MI355_MRING_NT=1
pmc() { # name counters -- bench args
  local name=$1 ctr=$2; shift 2
  local tag=${ctr// /_}
  rm -rf gpurun_out/${name}_$tag
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/${name}_$tag -- python3 bench.py "$@" > gpurun_out/${name}_$tag.out 2> gpurun_out/${name}_$tag.err
  local rc=$?; echo "${name}_$tag rc=$rc"; [ $rc -ge 124 ] && exit $rc
  python tools/pmc_summary.py gpurun_out/${name}_$tag > gpurun_out/${name}_$tag.txt 2>&1
  grep -A12 "spmv_csr_mring" gpurun_out/${name}_$tag.txt
}
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES"; do
  pmc mdiag_nat "$set" --workload mesh $COMMON
  pmc mdiag_rcm "$set" --workload mesh_perm --internal $COMMON
done
for one in 0 1; do
  MI355_MRING_ONE_ROUND=$one python3 bench.py --workload mesh_perm --internal --no-cpu-baseline --no-parity --no-extras --steps 30 --warmup 5 > gpurun_out/mdiag_deal_$one.json 2>gpurun_out/mdiag_deal_$one.err || exit 1
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/mdiag_deal_$one.json') if l.startswith('{')][-1]); print('one_round=$one', d['ms_per_step'], d['roofline']['kernel'])"
done
echo MDIAG_DONE
