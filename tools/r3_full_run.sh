#!/bin/bash
# round 3 measurement set, part 1: whole -m gpu suite, smoke, the default bench line, every workload once, the 2-rank self-launch
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests.log 2>&1
rc=$?; tail -n 6 gpurun_out/r3_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/r3_smoke.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_c4_driver_form.json 2> gpurun_out/r3_bench_c4.err; rc=$?; echo "bench (driver form) rc=$rc"
[ $rc -ge 124 ] && exit $rc
: > gpurun_out/r3_bench_lines.jsonl
timeout -k 10 400 python bench.py >> gpurun_out/r3_bench_lines.jsonl 2>> gpurun_out/r3_bench.err; echo "bench c4 rc=$?"
for w in c2 c3 fe fe_bcsr fe_spmm4 fe_spmm8 mesh mesh_small fe_perm mesh_perm c2_perm mesh_small_perm; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline >> gpurun_out/r3_bench_lines.jsonl 2>> gpurun_out/r3_bench.err; rc=$?; echo "bench $w rc=$rc"
  [ $rc -ge 124 ] && exit $rc
done
for w in fe_perm mesh_perm c2_perm mesh_small_perm; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --internal >> gpurun_out/r3_bench_lines.jsonl 2>> gpurun_out/r3_bench.err; echo "bench $w --internal rc=$?"
done
timeout -k 10 400 python bench.py --cold --steps 30 --warmup 3 --no-cpu-baseline >> gpurun_out/r3_bench_lines.jsonl 2>> gpurun_out/r3_bench.err; echo "bench c4 --cold rc=$?"
MI355_FORCE_DEVICE=0 MI355_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --workload c2 --steps 50 --warmup 5 >> gpurun_out/r3_bench_lines.jsonl 2>> gpurun_out/r3_bench.err; echo "bench --gpus 2 (one card, gloo) rc=$?"
MI355_FORCE_DEVICE=0 MI355_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 3 --workload c3 --steps 20 --warmup 3 >> gpurun_out/r3_bench_lines.jsonl 2>> gpurun_out/r3_bench.err; echo "bench --gpus 3 c3 (one card, gloo) rc=$?"
python - <<'PY'
import json
for ln in open('gpurun_out/r3_bench_lines.jsonl'):
    d = json.loads(ln); r = d['roofline']
    print(f"{d['config']['name']:16s} N={d['n_gpus']} {d['config'].get('numbering','')[:8]:8s} cold={d['config']['cold']!s:5s} us {r['launch_us']:8.2f} frac {r['frac']:.4f} GF {d['value']:9.1f} bitwise {d.get('parity',{}).get('bitwise')} {r['kernel'][:70]}")
PY
echo R3_PART1_DONE
