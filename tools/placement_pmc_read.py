import csv, glob, sys, collections
root = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "spmv_csr_ring" not in r["Kernel_Name"]: continue
        d = int(r["Dispatch_Id"])
        e = per.setdefault(d, {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ds = sorted(per)
    names = sorted(k for k in per[ds[0]] if k != "dur")
    print("handle  median_us  " + "  ".join(names))
    import statistics
    for h in range(len(ds) // reps):
        grp = [per[d] for d in ds[h * reps + 2:(h + 1) * reps]]   # skip the first two launches of a handle
        print(f"{h:6d}  {statistics.median(g['dur'] for g in grp):9.1f}  " + "  ".join(f"{statistics.median(g[k] for g in grp):14.0f}" for k in names))
