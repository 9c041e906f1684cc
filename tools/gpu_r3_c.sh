#!/bin/bash
# round 3, GPU pass C: the one-launch powers step
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "one_launch or powers or spmk or reorder" > gpurun_out/r3c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3c_pytest.log
tail -25 gpurun_out/r3c_pytest.log
: > gpurun_out/r3c_bench.jsonl
for f in 0 1 auto; do
  if [ $f = auto ]; then unset MI355_SPMK_FUSED; else export MI355_SPMK_FUSED=$f; fi
  timeout -k 10 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline >> gpurun_out/r3c_bench.jsonl 2> gpurun_out/r3c_bench.err; echo "bench c3 fused=$f rc=$?"
done
python - <<'PY'
import json
for ln in open('gpurun_out/r3c_bench.jsonl'):
    d = json.loads(ln)
    print(d['config']['name'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['parity']['bitwise'], d['kernel_info'].get('powers_step'))
PY
