set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_reorder_gpu.py -x -q -m gpu > gpurun_out/t_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/t_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/mring_ab.py > gpurun_out/mring_ab.log 2>&1; cat gpurun_out/mring_ab.log
for w in c2_perm mesh_small_perm mesh_perm; do
timeout -k 10 400 python bench.py --no-cpu-baseline --workload $w > gpurun_out/t_bench_$w.log 2>&1
python - $w <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(f"gpurun_out/t_bench_{tag}.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; k = d.get("kernel_info", {})
        print(tag, "launch_us", r["launch_us"], "frac", r["frac"], r["kernel"], "cold", r.get("cold_single_shot", {}).get("launch_us"), "box", r.get("this_box_stream_read", {}).get("gbs"),
              "tune", k.get("autotune_us"), "mring", k.get("mring_plan"), "reorder", d.get("reorder"), "bitwise", d.get("parity", {}).get("bitwise"))
PY
done
echo DONE
