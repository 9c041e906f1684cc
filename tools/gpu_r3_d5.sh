#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3d_bench.jsonl
export MI355_SPMM_TILE=1
for ch in 0 1 2 4 8 16 40; do
  MI355_SPMM_TILE_XCD_CHUNK=$ch timeout -k 10 300 python bench.py --workload fe_spmm4 --steps 40 --warmup 5 --no-cpu-baseline >> gpurun_out/r3d_bench.jsonl 2>> gpurun_out/r3d_bench.err; echo "bench fe_spmm4 chunk=$ch rc=$?"
done
python - <<'PY'
import json
for ln in open('gpurun_out/r3d_bench.jsonl'):
    d = json.loads(ln)
    print(d['config']['name'], 'us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'], d['parity']['bitwise'])
PY
