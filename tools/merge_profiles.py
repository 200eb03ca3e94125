#!/usr/bin/env python3
"""Merge the summaries of several tools/profile_all.sh calls (the eight workloads do not fit one gpurun call) into the
first directory:   python tools/merge_profiles.py gpurun_out/prof_r04 gpurun_out/prof_r04b gpurun_out/prof_r04c
then               python tools/make_traffic_json.py gpurun_out/prof_r04 r04"""
import glob, json, os, shutil, sys
dst = sys.argv[1]
S = json.load(open(os.path.join(dst, "summary.json")))
for src in sys.argv[2:]:
    S.update(json.load(open(os.path.join(src, "summary.json"))))
    for f in glob.glob(os.path.join(src, "*_kernel_stats.csv")):
        shutil.copy(f, dst)
json.dump(S, open(os.path.join(dst, "summary.json"), "w"), indent=1)
print(sorted(S), {w: (e.get("bench_profiled") or {}).get("kernel_build") for w, e in S.items()})
