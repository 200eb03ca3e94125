import sys, time, torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
M, B = 512, 1024
act = abi.lattice_action(abi.GFF, M, M, mass=10.0)
x = ops.lattice_initialise(act, B, 3, 0); w = torch.empty_like(x)
def run(a, b, steps=10):
    global x, w
    s = 0
    def step():
        global x, w
        nonlocal s
        if a:
            ops.lattice_sweep_draw(act, x, w, a, 0, 3, 0, s, fuse=a)
        x, w, _ = ops.lattice_sweep_draw_qoi(act, x, w, x, b, 1, 3, 0, s + a, 3, b)
        s += a + b + 1
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
for rep in range(2):
    for a, b in ((5, 5), (6, 4), (4, 5), (6, 3), (0, 5), (5, 0), (6, 0), (4, 0), (0, 4), (0, 3)):
        if b == 0:
            s = 0
            def st():
                global s
                ops.lattice_sweep_draw(act, x, w, a, 0, 3, 0, 0, fuse=a)
            for _ in range(3): st()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): st()
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
        else:
            el = run(a, b)
        print(f"{a} OR launch + ({b} OR + HB + QoI) launch: {el*1e3:.3f} ms", flush=True)
