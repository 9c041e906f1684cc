"""Dev tool: the last N kernel dispatches of a rocprofv3 --kernel-trace CSV as a timeline (start relative to the first shown, duration, gap to
the previous dispatch on ANY queue, queue id, kernel name) — what a multi-stream step looks like on the device.   python tools/trace_tail.py <kernel_trace.csv> [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"{(st - prev_end) / 1e3:7.2f}"
    print(f"{(st - t0) / 1e3:9.2f} us  +{(en - st) / 1e3:7.2f}  gap {gap:>7s}  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'][:110]}")
    prev_end = en if prev_end is None else max(prev_end, en)
