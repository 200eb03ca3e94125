#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage for one .hip file (gfx950)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "TotalSGPRs", "VGPRs Spill"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
print(f"{'kernel':70s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>6s}")
for r in rows:
    print(f"{r['name'][:70]:70s} {r.get('VGPRs',0):5d} {r.get('TotalSGPRs',0):5d} {r.get('ScratchSize [bytes/lane]',0):8d} {r.get('Occupancy [waves/SIMD]',0):4d} {r.get('LDS Size [bytes/block]',0):6d}")
