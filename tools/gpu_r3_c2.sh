#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "one_launch" > gpurun_out/r3c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3c_pytest.log
tail -5 gpurun_out/r3c_pytest.log
: > gpurun_out/r3c_bench.jsonl
export MI355_SPMK_FUSED=1
for noacq in 0 1; do
  export MI355_SPMK_NOACQ=$noacq
  timeout -k 10 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline >> gpurun_out/r3c_bench.jsonl 2> gpurun_out/r3c_bench.err; echo "bench c3 noacq=$noacq rc=$?"
done
MI355_SPMK_FUSED=0 timeout -k 10 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline >> gpurun_out/r3c_bench.jsonl 2> gpurun_out/r3c_bench.err
python - <<'PY'
import json
for ln in open('gpurun_out/r3c_bench.jsonl'):
    d = json.loads(ln)
    print(d['config']['name'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['parity']['bitwise'], d['kernel_info'].get('powers_step')['one_launch'])
PY
