set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/t_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/t_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/t_smoke.log
for w in c4 mesh c2; do
timeout -k 10 400 python bench.py --workload $w > gpurun_out/line_$w.log 2>&1
python - $w <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(f"gpurun_out/line_{tag}.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; k = d.get("kernel_info", {})
        print(tag, "value", d["value"], "launch_us", r["launch_us"], "frac", r["frac"], r["kernel"], "traffic", r.get("traffic"), "cold", r.get("cold_single_shot", {}).get("frac"), "pipe", r.get("in_pipeline", {}).get("frac"), "box", r.get("this_box_stream_read", {}).get("gbs"), r.get("this_box_stream_read", {}).get("kernel_over_stream"),
              "ring_plan", k.get("ring_plan"), "mring", k.get("mring_plan"), "bitwise", d.get("parity", {}).get("bitwise"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("all_cores", {}).get("value"))
PY
done
echo DONE
