#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE per launch of each kernel from rocprofv3 --pmc CSV output.

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read (16 B per lane, our double2 tile loads), so
it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, counter, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k][counter] = (sum(v) / len(v), len(v))
summary = []
for k, d in res.items():
    if not any(t in k for t in ("sweep_kernel", "or_kernel", "or_patch_kernel", "hmc")):
        continue
    fetch = d.get("FETCH_SIZE", (0, 0))
    write = d.get("WRITE_SIZE", (0, 0))
    summary.append({"kernel": k.split("(")[0], "launches": fetch[1], "FETCH_SIZE_KiB_raw": fetch[0],
                    "WRITE_SIZE_KiB_raw": write[0], "read_bytes_corrected": 2 * fetch[0] * 1024,
                    "write_bytes": write[0] * 1024, "hbm_bytes_per_launch": (2 * fetch[0] + write[0]) * 1024})
json.dump(summary, open(os.path.join(out, "traffic_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
