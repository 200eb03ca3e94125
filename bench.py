#!/usr/bin/env python3
"""bench.py -- lattice-site-updates/s of the MI355X sweep engine, with roofline and CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (default): BASELINE.json configs[3], the configuration the north-star target is quoted on --
quenched Schwinger model, 1024 x 1024, beta = 1, one "step" = one OverrelaxedHeatBathSampler::draw
= 10 overrelaxation + 1 heat-bath sweep (parameters_qft_template.in) over `chains` independent
chains per GPU.  Chains are sharded over ranks by global chain index (weak scaling: fixed chains per
GPU); the only collective is the packed statistics all-reduce after the timed region.
One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="schwinger", choices=["schwinger", "gff", "rotor_hmc", "quartic_hmc", "ho_hmc", "quartic_mlmc", "rotor_sweep"])
    ap.add_argument("--size", type=int, default=0, help="lattice extent (default: BASELINE size of the workload)")
    ap.add_argument("--chains", type=int, default=0, help="independent chains per GPU (default per workload)")
    ap.add_argument("--fuse", type=int, default=0, help="sweeps fused per launch (0 = library default)")
    ap.add_argument("--n-overrelax", type=int, default=10)
    ap.add_argument("--n-heatbath", type=int, default=1)
    ap.add_argument("--nt", type=int, default=100)
    ap.add_argument("--dt", type=float, default=0.0, help="HMC step size (default: 0.05 rotor, 0.02 quartic)")
    ap.add_argument("--seed", type=int, default=2481317)
    ap.add_argument("--thermalise", type=int, default=30,
                    help="untimed sampler draws before the warm-up (sweep workloads): the timed steps run on a "
                         "thermalised state -- heat-bath rejection rates depend on it -- and at steady clocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-draws", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(a, size):
    """Reference-order oracle on the host cores; run BEFORE this process touches the GPU."""
    wl = {"schwinger": "schwinger", "gff": "gff", "rotor_hmc": "rotor", "quartic_hmc": "quartic", "ho_hmc": "harmonic",
          "quartic_mlmc": "quartic", "rotor_sweep": "rotor_sweep"}[a.workload]
    draws = a.cpu_draws or {"schwinger": 5, "gff": 40, "rotor": 30, "quartic": 200, "harmonic": 100000, "rotor_sweep": 150}[wl]
    dt = a.dt or {"rotor": 0.05, "harmonic": 0.0558}.get(wl, 0.02)
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--workload", wl, "--size", str(size),
           "--draws", str(draws), "--n-overrelax", str(a.n_overrelax), "--n-heatbath", str(a.n_heatbath),
           "--nt", str(a.nt), "--dt", str(dt)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        return {"value": None, "error": out.stderr[-300:]}
    r = json.loads(out.stdout.strip().splitlines()[-1])
    return {"value": r["value"], "unit": "updates/s", "cores": r["cores"], "kind": "port",
            "per_core": r["per_core"], "sample": r["sample"] + " (reference-order sequential sweeps, mt19937_64)"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    size = a.size or {"schwinger": 1024, "gff": 512, "rotor_hmc": 65536, "quartic_hmc": 32768, "ho_hmc": 128,
                      "quartic_mlmc": 32768, "rotor_sweep": 65536}[a.workload]
    B = a.chains or {"schwinger": 32, "gff": 1024, "rotor_hmc": 1024, "quartic_hmc": 2048, "ho_hmc": 8192,
                     "quartic_mlmc": 512, "rotor_sweep": 1024}[a.workload]

    cpu = None
    if world == 1 and a.gpus == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a, size)

    import torch
    import torch.distributed as dist
    from mlmcpathintegral_amd import abi, chains, ops

    abi.load()  # no CPU fallback: raises when the HIP extension is missing
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # MLMCPI_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then share
    # devices and the collectives go through host memory); the driver's runs use RCCL ("nccl"), one GPU per rank.
    backend = os.environ.get("MLMCPI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    coll_device = "cuda" if backend == "nccl" else "cpu"
    chain0 = rank * B  # global chain indices of this rank: [chain0, chain0 + B)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    or_events, hb_events = [], []

    if a.workload in ("schwinger", "gff"):
        if a.workload == "schwinger":
            act = abi.lattice_action(abi.SCHWINGER, size, size, beta=1.0)
            sites = 2 * size * size
        else:
            act = abi.lattice_action(abi.GFF, size, size, mass=10.0)
            sites = size * size
        x = ops.lattice_initialise(act, B, a.seed, chain0)
        scratch = torch.empty_like(x)
        units_per_step = sites * (a.n_overrelax + a.n_heatbath) * B
        fuse = a.fuse or 4  # library default
        state = {"sweep": 0, "x": x, "scratch": scratch}

        def step(record):
            s = state["sweep"]
            if record:
                e0, e1, e2 = ev(), ev(), ev()
                e0.record()
            # same arithmetic as one call with (n_overrelax, n_heatbath); split only to time the two kernels
            # (ping-pong form: the buffers swap roles instead of being copied back).  The overrelaxation
            # sweeps are issued as full launches of `fuse` sweeps (timed: the dominant kernel) + a remainder.
            n_full = (a.n_overrelax // fuse) * fuse
            cur, oth = ops.lattice_sweep_draw_pingpong(act, state["x"], state["scratch"], n_full, 0, a.seed,
                                                       chain0, s, fuse)
            if record:
                e1.record()
            if a.n_overrelax - n_full:
                cur, oth = ops.lattice_sweep_draw_pingpong(act, cur, oth, a.n_overrelax - n_full, 0, a.seed, chain0,
                                                           s + n_full, fuse)
            if record:
                e1b = ev()
                e1b.record()
            state["x"], state["scratch"] = ops.lattice_sweep_draw_pingpong(act, cur, oth, 0, a.n_heatbath, a.seed, chain0,
                                                                           s + a.n_overrelax, fuse)
            if record:
                e2.record()
                or_events.append((e0, e1))
                hb_events.append((e1b, e2))
            state["sweep"] = s + a.n_overrelax + a.n_heatbath

        def qoi():
            if a.workload == "schwinger":
                return ops.qoi_avg_plaquette(state["x"], size, size)
            return ops.qoi_phi_squared(state["x"])
        bytes_per_unit = 16.0  # each entry read once and written once per sweep
    elif a.workload == "rotor_sweep":
        # OverrelaxedHeatBathSampler::draw on the rotor action, M_lat = 65536, a = 0.125
        act = abi.path_action(abi.ROTOR, size, size / 8.0, 0.25)
        x = ops.path_initialise(act, B, a.seed, chain0)
        scratch = torch.empty_like(x)
        units_per_step = size * (a.n_overrelax + a.n_heatbath) * B
        fuse = 1
        state = {"sweep": 0}

        def step(record):
            if record:
                e0, e1 = ev(), ev()
                e0.record()
            ops.path_sweep_draw(act, x, scratch, a.n_overrelax, a.n_heatbath, a.seed, chain0, state["sweep"])
            state["sweep"] += a.n_overrelax + a.n_heatbath
            if record:
                e1.record()
                or_events.append((e0, e1))

        def qoi():
            return ops.qoi_susceptibility(x, size / 8.0)
        bytes_per_unit = 16.0
    elif a.workload == "quartic_mlmc":
        # BASELINE configs[4]: quartic double well, 5 levels, finest M_lat = 32768, a = 0.125;
        # level l on rank l % world, every level instance runs B chains; a step = one Y sample per chain on every
        # level instance (2 trajectories of the level's sampler + the two-level step).
        from mlmcpathintegral_amd import mlmc
        n_level = 5
        est = mlmc.PathMLMC(abi.QUARTIC, size, size / 8.0, n_level, B, nt=a.nt, dt0=a.dt or 0.02, seed=a.seed, rank=rank,
                            world=world, n_sub=2, params=dict(lam=1.0, x0=1.0))
        est.thermalise(64)
        # site-steps per step, all levels (the same on every rank count): HMC of the feeding level + two-level pass
        units_per_step = 0
        for l in range(n_level):
            src = l if l == n_level - 1 else l + 1
            units_per_step += (2 * (a.nt + 1) * (size >> src) + (0 if l == n_level - 1 else (size >> l))) * B
        fuse = 1

        def step(record):
            if record:
                e0, e1 = ev(), ev()
                e0.record()
            est.pass_(1)
            if record:
                e1.record()
                or_events.append((e0, e1))

        def qoi():
            lv = est.levels[min(est.levels)]
            return ops.qoi_xsquared(lv.x)
        bytes_per_unit = 32.0
    else:
        kind = {"rotor_hmc": abi.ROTOR, "quartic_hmc": abi.QUARTIC, "ho_hmc": abi.HARMONIC}[a.workload]
        # a = 0.125 at the BASELINE sizes; config 1 (HO, M_lat = 128) keeps T_final = 4
        T_final = 4.0 if a.workload == "ho_hmc" else size / 8.0
        act = abi.path_action(kind, size, T_final, 0.25 if kind == abi.ROTOR else 1.0, 1.0, 1.0, 1.0)
        x = ops.path_initialise(act, B, a.seed, chain0)
        dt = a.dt or {abi.ROTOR: 0.05, abi.HARMONIC: 0.0558}.get(kind, 0.02)
        draws_per_step = 10 if a.workload == "ho_hmc" else 1  # short paths: several draws per launch
        hmc = ops.PathHMC(act, B, a.nt, dt, seed=a.seed, chain0=chain0)
        # untimed thermalisation from the reference's cold / random start with small steps, so that the
        # timed trajectories run at a realistic acceptance rate (reported as p_accept)
        ops.hmc_thermalise(hmc, x, 32)
        units_per_step = size * (a.nt + 1) * B * draws_per_step  # site-steps: one site x one force evaluation
        fuse = 1

        def step(record):
            if record:
                e0, e1 = ev(), ev()
                e0.record()
            if draws_per_step > 1:
                q, cnt = ops.path_hmc_run(hmc, x, draws_per_step, 1)  # draws + QoIs in one launch
                hmc.n_total += draws_per_step
                hmc.n_accepted += cnt
            else:
                hmc.draw(x)
            if record:
                e1.record()
                or_events.append((e0, e1))

        def qoi():
            return ops.qoi_susceptibility(x, T_final) if kind == abi.ROTOR else ops.qoi_xsquared(x)
        bytes_per_unit = 32.0  # x, p read and written once per leapfrog step

    if a.workload in ("schwinger", "gff", "rotor_sweep"):
        for _ in range(a.thermalise):
            step(False)
    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # the one collective: packed per-chain moments of a QoI, summed over ranks (RCCL)
    acc = torch.zeros((B, chains.N_MOMENTS), dtype=torch.float64, device="cuda")
    ops.stats_accumulate(acc, qoi())
    packed = chains.allreduce_moments(chains.pack_moments(acc).to(coll_device))
    if a.workload == "quartic_mlmc":
        mlmc_q, mlmc_e, mlmc_t = est.estimate(device=coll_device)  # the level-table exchange (RCCL when world > 1)
        mlmc_t = mlmc_t.cpu()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    qoi_mean = float(packed[1] / packed[0])

    if rank == 0:
        total_units = units_per_step * a.steps * (1 if a.workload == "quartic_mlmc" else world)
        ms = lambda pairs: sum(p[0].elapsed_time(p[1]) for p in pairs)
        or_ms = ms(or_events)
        result = {
            "metric": "lattice-site-updates/sec",
            "value": total_units / elapsed,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
        }
        if a.workload in ("schwinger", "gff"):
            n_launch = a.n_overrelax // fuse  # full launches of `fuse` overrelaxation sweeps
            result["config"] = {"workload": f"{a.workload} {size}x{size}, {a.n_overrelax} overrelaxation + "
                                            f"{a.n_heatbath} heat-bath sweeps per step, multicolour order",
                                "chains_per_gpu": B, "chains_total": B * world, "fuse": fuse,
                                "parallelism": f"chains sharded over {world} GPU(s), no data-path collective"}
            if n_launch:
                launch_ms = or_ms / (a.steps * n_launch)
                alg = bytes_per_unit * sites * B * fuse  # algorithmic bytes per launch (fuse sweeps)
                achieved = alg / (launch_ms * 1e-3) / 1e9
                special = size % 64 == 0 and fuse <= (6 if a.workload == "schwinger" else 4)
                kname = ((f"schwinger_or_patch_kernel<{fuse}>" if fuse <= 4 and os.environ.get("MLMCPI_OR_KERNEL") != "lds"
                          else f"schwinger_or_kernel<64,32,{fuse},{1024 if fuse >= 4 else 512}>") if a.workload == "schwinger" and special
                         else (f"gff_or_patch_kernel<{fuse}>" if os.environ.get("MLMCPI_OR_KERNEL") != "lds"
                               else f"gff_or_kernel<64,32,{fuse},256>") if special else f"{a.workload}_sweep_kernel<false,256>")
                result["roofline"] = {"kernel": kname + f" ({fuse} fused overrelaxation sweeps per launch)", "bound": "hbm",
                                      "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(a, B, fuse),
                                      "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg,
                                      "updates_per_s": sites * B * fuse / (launch_ms * 1e-3)}
            hb_ms = ms(hb_events)
            if n_launch:
                result["roofline"]["share_of_step"] = or_ms / a.steps / (1e3 * elapsed / a.steps)
            if a.n_heatbath:
                result["heatbath"] = {"launch_ms": hb_ms / (a.steps * a.n_heatbath),
                                      "updates_per_s": sites * B * a.n_heatbath * a.steps / (hb_ms * 1e-3),
                                      "share_of_step": hb_ms / a.steps / (1e3 * elapsed / a.steps),
                                      "algorithmic_GBps": 16.0 * sites * B * a.n_heatbath * a.steps / (hb_ms * 1e-3) / 1e9,
                                      "note": "fp64 VALU bound (Philox + von Mises rejection sampler), not HBM bound "
                                              "(SURVEY F9)"}
                insts = load_valu(a, B)
                if insts:
                    # vector-ALU issue roofline: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz lane-operations/s; every
                    # fp64 / int32 VALU instruction of a wave64 occupies its SIMD for 4 cycles
                    hb_s = result["heatbath"]["launch_ms"] * 1e-3
                    peak = 256 * 4 * 16 * 2.4e9
                    result["heatbath"]["valu"] = {"bound": "valu", "wave_insts_per_launch": insts,
                                                  "achieved": insts * 64 / hb_s / 1e12, "peak": peak / 1e12,
                                                  "unit": "T lane-ops/s", "frac": insts * 64 / hb_s / peak,
                                                  "source": "SQ_INSTS_VALU, profiles/traffic.json"}
        elif a.workload == "rotor_sweep":
            launch_ms = or_ms / a.steps
            alg = bytes_per_unit * units_per_step
            achieved = alg / (launch_ms * 1e-3) / 1e9
            result["config"] = {"workload": f"rotor M_lat={size}, {a.n_overrelax} overrelaxation + {a.n_heatbath} heat-bath "
                                            "sweeps per step, even/odd order", "chains_per_gpu": B, "chains_total": B * world,
                                "parallelism": f"chains sharded over {world} GPU(s), no data-path collective"}
            result["roofline"] = {"kernel": "rotor_sweep_kernel (all sweeps of a step)", "bound": "hbm", "achieved": achieved,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                  "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg,
                                  "note": "the heat-bath sweep is VALU bound (von Mises sampler); overrelaxation sweeps are "
                                          "fused on LDS-resident segments"}
        elif a.workload == "quartic_mlmc":
            result["scaling"] = "strong"
            result["config"] = {"workload": f"quartic MLMC, 5 levels, finest M_lat={size}, a=0.125, nt={a.nt}, one Y sample per "
                                            "chain and level per step (2 sampler trajectories + two-level step)",
                                "chains_per_level": B, "levels_on_rank0": sorted(est.levels),
                                "parallelism": f"level l on rank l % {world}; per pass one all-reduce of the [5, 5] level table"}
            result["mlmc"] = {"estimate": mlmc_q, "error": mlmc_e, "level_means": mlmc_t[:, 1].tolist(),
                              "level_variances": mlmc_t[:, 2].tolist(),
                              "acceptance_rank0": {str(k): v for k, v in est.p_accept().items()}}
            launch_ms = or_ms / a.steps
            alg = bytes_per_unit * units_per_step / world
            achieved = alg / (launch_ms * 1e-3) / 1e9
            result["roofline"] = {"kernel": "hmc_trajectory_kernel (the level samplers; > 99 % of the site-steps)", "bound": "hbm",
                                  "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                  "traffic": None, "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg,
                                  "note": "one step of all level instances of rank 0; states and momenta stay in "
                                          "registers for a whole trajectory, so frac may exceed 1"}
        else:
            launch_ms = or_ms / a.steps
            alg = bytes_per_unit * units_per_step
            achieved = alg / (launch_ms * 1e-3) / 1e9
            result["p_accept"] = float(hmc.n_accepted.double().mean()) / max(1, hmc.n_total)
            result["config"] = {"workload": f"{a.workload} M_lat={size}, nt={a.nt}, dt={hmc.dt}, fused trajectories",
                                "chains_per_gpu": B, "chains_total": B * world,
                                "parallelism": f"chains sharded over {world} GPU(s), no data-path collective"}
            result["roofline"] = {"kernel": "hmc_trajectory_kernel", "bound": "hbm", "achieved": achieved,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                  "traffic": None, "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg,
                                  "note": "state and momenta stay in registers for the whole trajectory: HBM "
                                          "sees 16 B per site per trajectory, so frac may exceed 1"}
        result["qoi_mean"] = qoi_mean
        if cpu is not None:
            result["cpu_baseline"] = cpu
            if cpu.get("value"):
                result["gpu_over_cpu"] = result["value"] / cpu["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def load_valu(a, B):
    """VALU wave-instructions per heat-bath launch from the committed SQ counter profile (scaled with the chains)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        for e in t.get("valu", []):
            if (e["workload"], e["size"]) == (a.workload, a.size or 1024):
                return e["SQ_INSTS_VALU_per_launch"] * B / e["chains"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def load_traffic(a, B, fuse):
    """HBM bytes per launch from the committed PMC profile of the same configuration, else None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        for e in t.get("entries", []):
            if (e["workload"], e["size"], e["chains"], e["fuse"]) == (a.workload, a.size or 1024, B, fuse):
                return e["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


if __name__ == "__main__":
    main()
