#!/usr/bin/env python3
"""Condense the output of tools/profile_all.sh: per workload the kernel-trace stats table, per-launch averages of the
SQ counters and of FETCH_SIZE / WRITE_SIZE (KiB; FETCH_SIZE doubled for the 16-byte-per-lane streaming reads of
gfx950, MI355X_MICROARCH.md HBM section) for the library's kernels -> <dir>/summary.json and <dir>/<workload>_kernel_stats.csv."""
import collections, csv, glob, json, os, shutil, sys
out = sys.argv[1]
summary = {}
for wdir in sorted(glob.glob(os.path.join(out, "*", ""))):
    w = os.path.basename(os.path.dirname(wdir))
    entry = {"kernels": {}}
    stats = glob.glob(os.path.join(wdir, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{w}_kernel_stats.csv"))
        for row in csv.DictReader(open(stats[0])):
            name = row["Name"].split("(")[0].replace("void ", "")
            if "mlmcpi::" in name:
                entry["kernels"].setdefault(name, {}).update(calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]), pct=float(row["Percentage"]))
    for sub in ("sq", "mix1", "mix2", "FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(wdir, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "mlmcpi::" in name:
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, d in acc.items():
            k = entry["kernels"].setdefault(name, {})
            for c, v in d.items():
                k[c] = sum(v) / len(v)
                k[c + "_launches"] = len(v)
    for name, k in entry["kernels"].items():
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            k["read_bytes_corrected"] = 2 * k["FETCH_SIZE"] * 1024
            k["write_bytes"] = k["WRITE_SIZE"] * 1024
            k["hbm_bytes_per_launch"] = k["read_bytes_corrected"] + k["write_bytes"]
    bp = os.path.join(wdir, "bench_profiled.json")
    if os.path.exists(bp) and os.path.getsize(bp):
        b = json.load(open(bp))
        entry["bench_profiled"] = {k: b.get(k) for k in ("value", "ms_per_step", "kernel_build", "config")}
    summary[w] = entry
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for w, e in summary.items():
    print(w)
    for name, k in sorted(e["kernels"].items(), key=lambda kv: -kv[1].get("pct", 0))[:6]:
        print(f"   {name[:60]:60s} {k.get('pct', 0):6.2f}% avg {k.get('avg_ns', 0) / 1e3:9.1f} us  VALU {k.get('SQ_INSTS_VALU', 0):.3g}  HBM {k.get('hbm_bytes_per_launch', 0):.4g} B")
