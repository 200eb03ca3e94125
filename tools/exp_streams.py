"""Experiment: the step (10 OR + 1 HB + QoI) on one stream over 32 chains vs two staggered streams of 16 chains each
(the VALU-bound heat bath of one half overlapping the LDS/latency-bound overrelaxation of the other)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlmcpathintegral_amd import abi, ops
abi.load()
size, seed = 1024, 2481317
act = abi.lattice_action(abi.SCHWINGER, size, size, beta=1.0)

class Half:
    def __init__(self, B, chain0, stream):
        self.B, self.chain0, self.stream = B, chain0, stream
        with torch.cuda.stream(stream):
            self.x = ops.lattice_initialise(act, B, seed, chain0)
            self.s = torch.empty_like(self.x)
            self.acc = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
        self.sweep = 0
    def draw(self, n_or=10, n_hb=1):
        with torch.cuda.stream(self.stream):
            self.x, self.s = ops.lattice_sweep_draw_pingpong(act, self.x, self.s, n_or, n_hb, seed, self.chain0, self.sweep, 0)
            self.sweep += n_or + n_hb
            ops.stats_accumulate(self.acc, ops.qoi_avg_plaquette(self.x, size, size))

def run(halves, steps, stagger):
    for h in halves:
        for _ in range(20): h.draw()
    torch.cuda.synchronize()
    if stagger and len(halves) > 1:
        halves[1].draw(10, 0)   # put the second half out of phase: one extra block of overrelaxation sweeps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for h in halves: h.draw()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tot = sum(h.B for h in halves)
    return 1e3 * el / steps, 2 * size * size * 11 * tot * steps / el

for B in (32, 64):
    ms, rate = run([Half(B, 0, torch.cuda.current_stream())], 20, False)
    print(f"1 stream  x {B} chains: {ms:.3f} ms/step  {rate/1e9:.1f} G/s")
    for stag in (False, True):
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        ms, rate = run([Half(B // 2, 0, s1), Half(B // 2, B // 2, s2)], 20, stag)
        print(f"2 streams x {B//2} chains (stagger={stag}): {ms:.3f} ms/step  {rate/1e9:.1f} G/s")
    s = [torch.cuda.Stream() for _ in range(4)]
    ms, rate = run([Half(B // 4, i * B // 4, s[i]) for i in range(4)], 20, False)
    print(f"4 streams x {B//4} chains: {ms:.3f} ms/step  {rate/1e9:.1f} G/s")
