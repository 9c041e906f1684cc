#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import csv, glob, sys, collections, re
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kprof"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per_dispatch.items():
        acc[names[d]][c].append(v)
def short(n):
    n = re.sub(r"mi355::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:60]
for k in sorted(acc):
    print(short(k))
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:36s} {sum(v)/len(v):16.1f}  (n={len(v)})")
