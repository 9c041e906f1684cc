set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_fe_matrix.py tests/test_spmm_gpu.py tests/test_reorder_gpu.py -x -q -m gpu -k "blocked or bcsr or fe or spmm or permuted_fe" > gpurun_out/t_tests.log 2>&1
rc=$?; tail -4 gpurun_out/t_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
for w in fe fe_bcsr fe_perm; do
timeout -k 10 400 python bench.py --no-cpu-baseline --workload $w > gpurun_out/line_$w.log 2>&1
python - $w <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(f"gpurun_out/line_{tag}.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; k = d.get("kernel_info", {})
        print(tag, "launch_us", r["launch_us"], "frac", r["frac"], r["kernel"], "cold", r.get("cold_single_shot", {}).get("frac"), "box", r.get("this_box_stream_read", {}).get("gbs"), "tune", k.get("autotune_us"), "bitwise", d.get("parity", {}).get("bitwise"))
PY
done
MI355_BCSR_TILE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --workload fe_bcsr 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fe_bcsr TILE=0', d['roofline']['launch_us'], d['roofline']['kernel'])"
MI355_BCSR_TILE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --workload fe_bcsr 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fe_bcsr TILE=1', d['roofline']['launch_us'], d['roofline']['kernel'])"
