#!/bin/bash
# round 3, second session: after the paired value loads and the block-shape rule — whole -m gpu suite, smoke, bench lines
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3b_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/r3b_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3b_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 gpurun_out/r3b_smoke.log
: > gpurun_out/r3b_bench_lines.jsonl
timeout -k 10 400 python bench.py --steps 20 --warmup 5 >> gpurun_out/r3b_bench_lines.jsonl 2>> gpurun_out/r3b_bench.err; echo "bench (driver form) rc=$?"
for w in c4 c2 c3 mesh mesh_small fe; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline >> gpurun_out/r3b_bench_lines.jsonl 2>> gpurun_out/r3b_bench.err; rc=$?; echo "bench $w rc=$rc"
  [ $rc -ge 124 ] && exit $rc
done
for w in mesh_perm c2_perm; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --internal >> gpurun_out/r3b_bench_lines.jsonl 2>> gpurun_out/r3b_bench.err; echo "bench $w --internal rc=$?"
done
python - <<'PY'
import json
for ln in open('gpurun_out/r3b_bench_lines.jsonl'):
    d = json.loads(ln); r = d['roofline']; c = r.get('cold_single_shot') or {}
    print(f"{d['config']['name']:16s} N={d['n_gpus']} us {r['launch_us']:8.2f} frac {r['frac']:.4f} cold {c.get('launch_us')} {c.get('frac')} GF {d['value']:9.1f} bitwise {d.get('parity',{}).get('bitwise')} {r['kernel'][:70]}")
PY
echo DONE
