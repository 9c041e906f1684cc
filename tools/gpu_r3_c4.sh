#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "one_launch or powers or reorder or internal" > gpurun_out/r3c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3c_pytest.log
tail -5 gpurun_out/r3c_pytest.log
for args in "1000000 4 300" "1000000 8 150" "300000 4 300" "5000000 3 60" "100000 8 300" "2000000 2 200"; do
  timeout -k 10 300 python tools/spmk_stress.py $args 2>&1 | grep -v amdgpu.ids | tail -3
done | tee gpurun_out/r3c_spmk_stress.txt
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
