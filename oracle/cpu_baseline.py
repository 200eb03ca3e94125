"""CPU baseline leg of bench.py: the oracle's reference-order samplers timed on the host cores.

TEST/BENCH INFRASTRUCTURE ONLY.  One independent chain per process (the reference's own parallel
model: one chain per MPI rank), steady-state draws only (constructor work -- state initialisation,
burn-in, HMC auto-tuning -- is excluded).  Prints one JSON object.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _worker(args):
    import numpy as np
    import oracle as O
    workload, size, draws, n_or, n_hb, nt, dt, seconds = args
    L = O.lib()
    if workload == "schwinger":
        A = O.Action(O.SCHWINGER, Mt=size, Mx=size, beta=1.0)
        units = 2 * size * size * (n_or + n_hb)
        s = L.orc_heatbath_new(A.h, n_hb, n_or, 0, 0)
        draw = lambda x: L.orc_heatbath_draw(s, x)
    elif workload == "gff":
        A = O.Action(O.GFF, Mt=size, Mx=size, mass=10.0)
        units = size * size * (n_or + n_hb)
        s = L.orc_heatbath_new(A.h, n_hb, n_or, 0, 0)
        draw = lambda x: L.orc_heatbath_draw(s, x)
    elif workload == "rotor_sweep":
        A = O.Action(O.ROTOR, M=size, T_final=size / 8.0, m0=0.25)
        units = size * (n_or + n_hb)
        s = L.orc_heatbath_new(A.h, n_hb, n_or, 0, 0)
        draw = lambda x: L.orc_heatbath_draw(s, x)
    else:  # HMC, fixed dt (no auto-tune)
        if workload == "quartic":
            A = O.Action(O.QUARTIC, M=size, T_final=size / 8.0, m0=1.0, mu2=1.0, lam=1.0, x0=1.0)
        elif workload == "harmonic":
            A = O.Action(O.HARMONIC, M=size, T_final=4.0, m0=1.0, mu2=1.0)
        else:
            A = O.Action(O.ROTOR, M=size, T_final=size / 8.0, m0=0.25)
        units = size * (nt + 1)
        s = L.orc_hmc_new(A.h, nt, dt, 1, 0, 0, 0, 0)
        draw = lambda x: L.orc_hmc_draw(s, x)
    # start where the reference starts (rotor / Schwinger: U(-pi, pi) per entry, rotoraction.cc:82-89,
    # quenchedschwingeraction.cc:198-204; others zero) and move off it before timing: the rejection rate of the heat
    # bath and the HMC acceptance depend on the state
    if workload in ("schwinger", "rotor_sweep", "rotor"):
        x = np.random.default_rng(12345).uniform(-np.pi, np.pi, A.size)
    else:
        x = np.zeros(A.size)
    for _ in range(2):
        draw(x)  # warm-up (page in, first touch, a first pass of thermalisation)
    # at least `draws` draws and at least `seconds` of them: the sample carries an error bar (spread over the cores)
    t0 = time.perf_counter()
    done = 0
    while True:
        draw(x)
        done += 1
        el = time.perf_counter() - t0
        if done >= draws and el >= seconds:
            break
    return units * done, el, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="schwinger")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--draws", type=int, default=4, help="draws per core at least")
    ap.add_argument("--seconds", type=float, default=20.0, help="timed seconds per core at least (BASELINE.md section 3: 10-30 s of CPU work)")
    ap.add_argument("--cores", type=int, default=0)
    ap.add_argument("--n-overrelax", type=int, default=10)
    ap.add_argument("--n-heatbath", type=int, default=1)
    ap.add_argument("--nt", type=int, default=100)
    ap.add_argument("--dt", type=float, default=0.1)
    a = ap.parse_args()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # One chain per core the job may use: the scheduler affinity, cut down to the cgroup CPU quota when there is one
    # ("1 chain per core on all host cores").  A second point on 16 cores -- the CPU share of a
    # 1-GPU job on the GPU boxes, whose affinity mask shows the whole host -- is taken when more than 16 are available.
    quota = avail
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    cores = a.cores or min(avail, quota)
    import oracle as O
    O.build()  # compile once, before forking
    job = (a.workload, a.size, a.draws, a.n_overrelax, a.n_heatbath, a.nt, a.dt, a.seconds)

    def run(n):
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(n) as pool:
            res = pool.map(_worker, [job] * n, chunksize=1)
        wall = time.perf_counter() - t0
        rates = [u / el for u, el, _ in res]
        rate = sum(rates)  # all processes run concurrently: aggregate rate
        mean = rate / n
        std = (sum((r - mean) ** 2 for r in rates) / max(1, n - 1)) ** 0.5
        return {"value": rate, "per_core": mean, "per_core_min": min(rates), "per_core_max": max(rates), "per_core_std": std,
                "value_error": std * n ** 0.5, "cores": n, "wall_s": wall, "timed_s_per_core": sum(el for _, el, _ in res) / n,
                "draws_per_core": [d for _, _, d in res]}

    out = run(cores)
    out.update(cores_available=avail, cpu_quota=quota,
               sample=f"{min(out['draws_per_core'])}-{max(out['draws_per_core'])} draws per core (>= {a.seconds:g} s timed) of "
                      f"{a.workload} {a.size}, one chain per core")
    if cores > 16 and not a.cores:
        job = job[:-1] + (min(a.seconds, 5.0),)   # the second point: a short sample
        out["point_16"] = run(16)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
