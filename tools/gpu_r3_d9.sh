#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for t in 2 3; do
MI355_SPMM_TILE=$t timeout -k 10 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q > gpurun_out/r3d_pytest_tile$t.log 2>&1; echo "pytest tile=$t rc=$?"
tail -2 gpurun_out/r3d_pytest_tile$t.log
done
: > gpurun_out/r3d_bench.jsonl
for w in fe_spmm4 fe_spmm8; do
  for y16 in 0 1 0 1; do
    MI355_SPMM_TILE=3 MI355_SPMM_Y16=$y16 timeout -k 10 300 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline >> gpurun_out/r3d_bench.jsonl 2>> gpurun_out/r3d_bench.err; echo "bench $w y16=$y16 rc=$?"
  done
done
python - <<'PY'
import json
for ln in open('gpurun_out/r3d_bench.jsonl'):
    d = json.loads(ln)
    print(d['config']['name'], 'us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'], d['parity']['bitwise'])
PY
