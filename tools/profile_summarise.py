#!/usr/bin/env python3
"""Condense the output of tools/profile_all.sh: per workload the kernel-trace stats table, per-launch averages of the
SQ counters and of FETCH_SIZE / WRITE_SIZE (KiB; FETCH_SIZE doubled for the 16-byte-per-lane streaming reads of
gfx950, MI355X_MICROARCH.md HBM section) for the library's kernels -> <dir>/summary.json and <dir>/<workload>_kernel_stats.csv."""
import collections, csv, glob, json, os, shutil, sys
out = sys.argv[1]
summary = {}


def grid_of(row):
    """total threads of a dispatch (rocprofv3 writes Grid_Size in the counter files, Grid_Size_X/Y/Z in the kernel trace)"""
    if row.get("Grid_Size"):
        return str(int(float(row["Grid_Size"])))
    g = 1
    for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"):
        g *= int(float(row.get(k) or 1))
    return str(g)

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
    # The HMC trajectory kernel runs with one name on lattices of every level (the untimed direct-HMC thermalisation of the fine
    # levels and the timed coarsest-level draws of quartic_mlmc_hier): keyed by grid size as well, from the kernel trace.
    per_grid = collections.defaultdict(list)
    for f in glob.glob(os.path.join(wdir, "stats", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "hmc_trajectory_kernel" in row["Kernel_Name"]:
                per_grid[row["Kernel_Name"].split("(")[0].replace("void ", "") + " grid=" + grid_of(row)].append(
                    float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    total_ns = sum(k["avg_ns"] * k["calls"] for k in entry["kernels"].values() if "avg_ns" in k) or 1.0
    for name, v in per_grid.items():
        entry["kernels"].setdefault(name, {}).update(calls=len(v), avg_ns=sum(v) / len(v), pct=100.0 * sum(v) / total_ns, by_grid=True)
    for sub in ("sq", "mix1", "mix2", "FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(wdir, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "mlmcpi::" in name:
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if "hmc_trajectory_kernel" in name:
                        acc[name + " grid=" + grid_of(row)][row["Counter_Name"]].append(float(row["Counter_Value"]))
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
