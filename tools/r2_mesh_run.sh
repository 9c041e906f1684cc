set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/t_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
for w in mesh mesh_perm mesh_small mesh_small_perm c2_perm fe fe_perm c4; do
timeout -k 10 400 python bench.py --no-cpu-baseline --workload $w > gpurun_out/line_$w.log 2>&1
python - $w <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(f"gpurun_out/line_{tag}.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; k = d.get("kernel_info", {})
        print(tag, "launch_us", r["launch_us"], "frac", r["frac"], r["kernel"], "cold", r.get("cold_single_shot", {}).get("frac"), "box", r.get("this_box_stream_read", {}).get("gbs"), r.get("this_box_stream_read", {}).get("kernel_over_stream"),
              "tune", k.get("autotune_us"), "mring", k.get("mring_plan"), "reorder", d.get("reorder"), "bitwise", d.get("parity", {}).get("bitwise"))
PY
done
echo DONE
