"""Dev tool: turn what tools/prof_run.sh left under gpurun_out/ into the files committed under profiles/ (ROUND=NN in the environment, default 5):
   profiles/rNN_bench_<W>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/rNN_bench_<W>_pmc_raw.txt (per-kernel means of the
   FETCH_SIZE / WRITE_SIZE passes) and profiles/rNN_bench_<W>_pmc.json (what bench.py's roofline.traffic reads).
   python tools/prof_collect.py <workload> "<kernel name as bench.py prints it>" <algorithmic bytes> "<note>" """
import csv, json, os, re, sys
W, kernel, alg, note = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
RND = int(os.environ.get("ROUND", "5"))
RR = f"r{RND:02d}"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def mean_of(counter):
    """the per-kernel mean of `counter` for `kernel` out of tools/pmc_summary.py's text"""
    txt = open(os.path.join(G, f"pmc_{W}_{counter}.txt")).read()
    name = None
    for ln in txt.splitlines():
        if not ln.startswith(" "):
            name = ln.strip()
        elif counter in ln and name and name.replace("void ", "").startswith(kernel[:56]):
            m = re.search(r"([0-9.]+)\s+\(n=(\d+)\)", ln)
            return float(m.group(1)), int(m.group(2))
    raise SystemExit(f"{counter}: kernel {kernel} not found in pmc_{W}_{counter}.txt")


fetch, n_f = mean_of("FETCH_SIZE")
write, n_w = mean_of("WRITE_SIZE")
stats_src = os.path.join(G, f"prof_{W}_kernel_stats.csv")
rows = list(csv.DictReader(open(stats_src)))
row = next(r for r in rows if kernel in r["Name"])
bench_us = None
for ln in open(os.path.join(G, "prof_a.log")) if os.path.exists(os.path.join(G, "prof_a.log")) else []:
    if ln.startswith("BENCH-UNDER-PROFILER " + W + " "):
        bench_us = float(ln.split()[2])
with open(os.path.join(P, f"{RR}_bench_{W}_kernel_stats.csv"), "w") as f:
    f.write(open(stats_src).read())
with open(os.path.join(P, f"{RR}_bench_{W}_pmc_raw.txt"), "w") as f:
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f.write(f"== rocprofv3 --pmc {c} -- python3 bench.py --workload {W} --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras (per-kernel means, KB)\n")
        f.write(open(os.path.join(G, f"pmc_{W}_{c}.txt")).read() + "\n")
rd, wr = int(round(fetch * 1024 * 2.0)), int(round(write * 1024))
out = {
    "round": RND, "workload": W, "kernel": kernel,
    "command": f"rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --workload {W} --steps 10 --warmup 2 "
               f"--no-cpu-baseline --no-parity --no-extras   (tools/prof_run.sh, tools/prof_collect.py; raw per-kernel means: profiles/{RR}_bench_{W}_pmc_raw.txt)",
    "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write, "dispatches": n_f, "fetch_correction": 2.0,
    "fetch_correction_source": "MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads; calibrated in round 1 on a "
                               "kernel of known byte count (x2.000); WRITE_SIZE read as is",
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round((rd + wr) / alg, 4), "note": note,
    "kernel_trace_same_session": {"file": f"profiles/{RR}_bench_{W}_kernel_stats.csv", "name": row["Name"], "calls": int(row["Calls"]),
                                  "mean_us": round(float(row["AverageNs"]) / 1e3, 2), "min_us": round(float(row["MinNs"]) / 1e3, 2),
                                  "max_us": round(float(row["MaxNs"]) / 1e3, 2), "bench_launch_us_hip_events_under_profiler": bench_us,
                                  "includes": "the bench's 55 launches + the launches of this instantiation at create (candidates)"},
}
json.dump(out, open(os.path.join(P, f"{RR}_bench_{W}_pmc.json"), "w"), indent=1)
print(json.dumps(out["kernel_trace_same_session"]), out["hbm_bytes_per_launch"], out["traffic_over_algorithmic"])
