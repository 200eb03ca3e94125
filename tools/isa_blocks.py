#!/usr/bin/env python3
"""Static ISA map of one kernel: VALU instructions per basic block (labels, branches, barriers).
   tools/isa_blocks.py <file.hip> <kernel-name-substring> [min_valu]"""
import collections, re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 30
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + sys.argv[4:] + [
                       "-Iinclude", "-o", "/tmp/_isa.s", src], stderr=subprocess.DEVNULL)
lines = open("/tmp/_isa.s").read().split("\n")
start, funcs = None, {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        start = (m.group(1), i)
    if l.strip().startswith("s_endpgm") and start:
        funcs.setdefault(start[0], (start[1], i))
for name, (a, b) in funcs.items():
    if pat not in name:
        continue
    body = lines[a:b]
    ops = collections.Counter(t.split()[0] for t in (l.strip() for l in body) if re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_)", t))
    print(name, "static insts", sum(ops.values()), "VALU", sum(c for o, c in ops.items() if o.startswith("v_")))
    print("  top:", ", ".join(f"{o} {c}" for o, c in ops.most_common(14)))
    v = last = 0
    for i, l in enumerate(body):
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith(("s_cbranch", "s_branch", "s_barrier")):
            if v - last >= minv:
                print(f"{i:6d} valu_block={v - last:5d} total={v:6d}  {t[:70]}")
            last = v
        if t.startswith("v_"):
            v += 1
