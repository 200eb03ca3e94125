#!/usr/bin/env python3
"""Merge the second profile_all.sh call (tag Tb) into the first (tag T): summary.json (dict update) and *_kernel_stats.csv.
   tools/merge_profiles.py gpurun_out/prof_T gpurun_out/prof_Tb"""
import glob, json, os, shutil, sys
a_dir, b_dir = sys.argv[1], sys.argv[2]
a = json.load(open(os.path.join(a_dir, "summary.json")))
a.update(json.load(open(os.path.join(b_dir, "summary.json"))))
json.dump(a, open(os.path.join(a_dir, "summary.json"), "w"), indent=1)
for f in glob.glob(os.path.join(b_dir, "*_kernel_stats.csv")):
    shutil.copy(f, a_dir)
print("workloads:", ", ".join(a))
