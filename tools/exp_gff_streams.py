#!/usr/bin/env python3
"""GFF 512^2, 10 + 1 sweeps + QoI: the chains of the batch split over several HIP streams.  The draw is two launches -- an
HBM-bound one (gff_or_block_kernel<5>) and a vector-issue-bound one (gff_or_heat_kernel<5>); chains are independent, so the
launches of different sub-batches may run side by side."""
import sys, time
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
M, BT = 512, 1024
act = abi.lattice_action(abi.GFF, M, M, mass=10.0)


def run(nstreams, offset, steps=12):
    B = BT // nstreams
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    st = []
    for k in range(nstreams):
        with torch.cuda.stream(streams[k]):
            x = ops.lattice_initialise(act, B, 3, k * B)
            st.append({"x": x, "w": torch.empty_like(x), "s": 0, "acc": torch.zeros((B, 5), dtype=torch.float64, device="cuda")})

    def step(k):
        d = st[k]
        with torch.cuda.stream(streams[k]):
            d["x"], d["w"], _ = ops.lattice_sweep_draw_qoi(act, d["x"], d["w"], d["x"], 10, 1, 3, k * B, d["s"], 3, 0, acc=d["acc"])
            d["s"] += 11
    for _ in range(3):
        for k in range(nstreams):
            step(k)
    torch.cuda.synchronize()
    if offset and nstreams > 1:   # start the streams half a draw apart: an extra overrelaxation-only draw on the odd ones
        for k in range(1, nstreams, 2):
            d = st[k]
            with torch.cuda.stream(streams[k]):
                ops.lattice_sweep_draw(act, d["x"], d["w"], 5, 0, 3, k * B, d["s"])
                d["s"] += 5
    t0 = time.perf_counter()
    for _ in range(steps):
        for k in range(nstreams):
            step(k)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    return el


for rep in range(2):
    for ns, off in ((1, False), (2, False), (2, True), (4, False), (4, True), (8, True)):
        el = run(ns, off)
        print(f"streams {ns} offset {off}: {el*1e3:.3f} ms per step of {BT} chains = {M*M*11*BT/el/1e9:.0f} G site-updates/s", flush=True)
