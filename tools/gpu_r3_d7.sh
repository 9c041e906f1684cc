#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3d_bench.jsonl
export MI355_SPMM_TILE=1
for dbg in 0 64 96 104 100 97; do
  MI355_SPMM_DBG=$dbg timeout -k 10 300 python bench.py --workload fe_spmm4 --steps 40 --warmup 5 --no-cpu-baseline --no-parity >> gpurun_out/r3d_bench.jsonl 2>> gpurun_out/r3d_bench.err; echo "bench fe_spmm4 dbg=$dbg rc=$?"
done
python - <<'PY'
import json
for ln in open('gpurun_out/r3d_bench.jsonl'):
    d = json.loads(ln)
    print(d['config']['name'], 'us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'], d.get('kernel_info',{}).get('longest_list'))
PY
