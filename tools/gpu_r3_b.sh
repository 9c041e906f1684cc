#!/bin/bash
# round 3, GPU pass B: new tests (dot epilogue, internal numbering, column-major blocks, upwind push), pipeline numbers, --internal
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "dot_in_its_epilogue or internal_numbering or column_major or sharing_one_card" > gpurun_out/r3b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3b_pytest.log
tail -15 gpurun_out/r3b_pytest.log
: > gpurun_out/r3b_bench.jsonl
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline >> gpurun_out/r3b_bench.jsonl 2> gpurun_out/r3b_bench.err; echo "bench c4 rc=$?"
for w in mesh_perm c2_perm fe_perm mesh_small_perm; do
  timeout -k 10 400 python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-extras >> gpurun_out/r3b_bench.jsonl 2>> gpurun_out/r3b_bench.err; echo "bench $w rc=$?"
  timeout -k 10 400 python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --internal >> gpurun_out/r3b_bench.jsonl 2>> gpurun_out/r3b_bench.err; echo "bench $w --internal rc=$?"
done
for w in mesh c2 fe mesh_small; do
  timeout -k 10 400 python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-extras >> gpurun_out/r3b_bench.jsonl 2>> gpurun_out/r3b_bench.err; echo "bench $w rc=$?"
done
python - <<'PY'
import json
for ln in open('gpurun_out/r3b_bench.jsonl'):
    d = json.loads(ln)
    r = d['roofline']
    print(d['config']['name'], d['config'].get('numbering','')[:8], 'us', r['launch_us'], 'frac', r['frac'], 'bitwise', d.get('parity',{}).get('bitwise'), r.get('in_pipeline'))
PY
