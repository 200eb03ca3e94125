#!/usr/bin/env python3
"""bench.py -- lattice-site-updates/s of the MI355X sweep engine, with roofline and CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (default): BASELINE.json configs[3], the configuration the north-star target is quoted on --
quenched Schwinger model, 1024 x 1024, beta = 1.  One "step" is one pass of the reference's sampling loop
(montecarlo/montecarlosinglelevel.cc:59-77) over `chains` independent chains per GPU:
    sampler->draw      10 overrelaxation + 1 heat-bath sweep (parameters_qft_template.in)
    qoi->evaluate      average plaquette, one reduction pass over the state
    record_sample      per-chain moment sums (mlmcpi_stats_accumulate)
Chains are sharded over ranks by global chain index (weak scaling: fixed chains per GPU); the only collective is
the packed statistics all-reduce after the timed region.  `--gpus N` without a launcher starts the N ranks itself.
One JSON line is printed by rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4   # 256 CUs x 4 SIMDs, one fp64-class wave64 instruction per 4 cycles, 2.4 GHz

WORKLOADS = ["schwinger", "gff", "rotor_hmc", "quartic_hmc", "ho_hmc", "quartic_mlmc", "quartic_mlmc_hier", "rotor_sweep"]
DEFAULT_SIZE = {"schwinger": 1024, "gff": 512, "rotor_hmc": 65536, "quartic_hmc": 32768, "ho_hmc": 128,
                "quartic_mlmc": 32768, "quartic_mlmc_hier": 32768, "rotor_sweep": 65536}
DEFAULT_CHAINS = {"schwinger": 32, "gff": 1024, "rotor_hmc": 1024, "quartic_hmc": 2048, "ho_hmc": 8192,
                  # quartic_mlmc_hier: the plateau of tools/scan_hier_chains.sh (512: 1.04, 1024: 1.24, 2048: 1.35, 4096: 1.35 T
                  # site-steps/s: the coarsest-level HMC kernel is one 256-thread workgroup per chain, 2048 chains = 8 waves per SIMD)
                  "quartic_mlmc": 512, "quartic_mlmc_hier": 2048, "rotor_sweep": 1024}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="schwinger", choices=WORKLOADS)
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
    ap.add_argument("--no-fused-qoi", action="store_true",
                    help="Schwinger: evaluate the average plaquette in a pass of its own instead of inside the heat-bath launch")
    ap.add_argument("--no-extra-points", action="store_true",
                    help="skip the single-chain and 128-chain side measurements of the default workload")
    ap.add_argument("--probes", action="store_true",
                    help="schwinger: time the HBM-bound kernels of the path after the timed steps even with --no-extra-points")
    ap.add_argument("--cpu-draws", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="timed seconds per core of the CPU baseline, at least")
    ap.add_argument("--epsilon", type=float, default=2e-3, help="quartic_mlmc_hier: tolerance of the untimed run to convergence")
    ap.add_argument("--hier-sub-factor", type=float, default=1.0,
                    help="quartic_mlmc_hier: draws between coarse samples = this x the reference's ceil(2 tau_int) (experiment)")
    ap.add_argument("--allow-variant", action="store_true",
                    help="tools only (tools/ab.sh): accept MLMCPI_LIB_VARIANT and mark the line as not a record")
    ap.add_argument("--t-final", type=float, default=0.0, help="quartic_mlmc_hier: T_final (default size / 8, i.e. a = 0.125 on the finest level)")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) before this process
    touches a GPU, pass their output through, exit with their status.  Never prints an n_gpus that did not run."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # The host driver of this pool supports dmabuf IPC only: with the legacy IPC mode RCCL's intra-node transport (and
    # any sharing of device memory between the ranks' processes) fails in hipIpcGetMemHandle.  The image exports it
    # already; set here too so that a shell without it still starts ranks that can talk.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def build_id():
    """Identifies the kernel build the committed PMC figures belong to: hash of the kernel sources."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mlmcpathintegral_amd", "csrc")
    for f in ("device_common.hpp", "internal.hpp", "lattice2d.hip", "path1d.hip", "runtime.hip", "gff_levels.hip", "comm_rccl.cc"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def lib_sha256(path):
    """SHA-256 of the shared library this process loaded (abi.LIB_PATH): ties a line to a binary, as kernel_build ties it to
    the sources."""
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def pmc_entry(section, **match):
    """Per-launch PMC figures (profiles/traffic.json) of the CURRENT kernel build, else None: a figure measured on an
    older build is not quoted."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    bid = build_id()
    for e in t.get(section, []):
        if e.get("build") == bid and all(e.get(k) == v for k, v in match.items()):
            return e
    return None


def pmc_prefix_entry(section, kernel_prefix, **match):
    """as pmc_entry, the kernel given by the start of its name"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    bid = build_id()
    for e in t.get(section, []):
        if e.get("build") == bid and e.get("kernel", "").replace(" ", "").startswith(kernel_prefix.replace(" ", "")) and all(e.get(k) == v for k, v in match.items()):
            return e
    return None


def under_profiler():
    env = os.environ
    return bool(env.get("ROCP_TOOL_LIBRARIES") or env.get("ROCPROFILER_REGISTER_FORCE_LOAD")
                or "rocprof" in env.get("LD_PRELOAD", "") or env.get("ROCPROF_OUTPUT_PATH"))


def cpu_baseline(a, size):
    """Reference-order oracle on the host cores; run BEFORE this process touches the GPU."""
    wl = {"schwinger": "schwinger", "gff": "gff", "rotor_hmc": "rotor", "quartic_hmc": "quartic", "ho_hmc": "harmonic",
          "quartic_mlmc": "quartic", "quartic_mlmc_hier": "quartic", "rotor_sweep": "rotor_sweep"}[a.workload]
    draws = a.cpu_draws or {"schwinger": 5, "gff": 40, "rotor": 30, "quartic": 200, "harmonic": 100000, "rotor_sweep": 150}[wl]
    dt = a.dt or {"rotor": 0.05, "harmonic": 0.0558}.get(wl, 0.02)
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--workload", wl, "--size", str(size),
           "--draws", str(draws), "--n-overrelax", str(a.n_overrelax), "--n-heatbath", str(a.n_heatbath),
           "--nt", str(a.nt), "--dt", str(dt), "--seconds", str(a.cpu_seconds)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        return {"value": None, "error": out.stderr[-300:]}
    r = json.loads(out.stdout.strip().splitlines()[-1])
    out16 = None
    if r.get("point_16"):
        out16 = {"value": r["point_16"]["value"], "unit": "updates/s", "cores": 16, "per_core": r["point_16"]["per_core"],
                 "note": "the CPU share of a 1-GPU job on the GPU boxes; same sample"}
    return {"value": r["value"], "unit": "updates/s", "cores": r["cores"], "cores_available": r.get("cores_available"),
            "cpu_quota": r.get("cpu_quota"), "kind": "port", "per_core": r["per_core"], "per_core_min": r.get("per_core_min"),
            "per_core_max": r.get("per_core_max"), "per_core_std": r.get("per_core_std"), "value_error": r.get("value_error"),
            "timed_s_per_core": r.get("timed_s_per_core"), "wall_s": r.get("wall_s"), "point_16": out16,
            "sample": r["sample"] + " (reference-order sequential sweeps, mt19937_64)",
            "note": "the port runs ~2.8x faster per core than the reference itself measured in SURVEY 6.2 "
                    "(12 M link-updates/s/core for 10 OR + 1 HB), so gpu_over_cpu understates the gap to the reference"}


def cxx_path(a, size):
    """The same sampling loop through the C++ host layer (include/mlmcpi/*.hh: OverrelaxedHeatBathSampler::draw_with_qoi +
    mlmcpi_stats_accumulate, the loop of montecarlosinglelevel.cc:59-77), run by host/driver BEFORE this process touches the
    GPU: north_star's host is the C++ one, so its rate is recorded next to the Python-driven one of the same session."""
    exe = os.path.join(ROOT, "host", "driver")
    if not os.path.exists(exe):
        return {"error": "host/driver not built"}
    out = {}
    # (enough samples for the clocks to settle: the driver is the first process on the GPU; with 40 draws -- 28 ms -- the C++
    # loop read 3-4 % slower than the Python-driven one that follows it)
    for name, batch, samples in (("chains_32", 32, 400), ("single_chain", 1, 2000)):
        cmd = [exe, "--method", "throughput", "--action", "schwinger", "--Mt_lat", str(size), "--sampler", "heatbath",
               "--batch", str(batch), "--n_samples", str(samples), "--n_burnin", "30", "--seed", str(a.seed),
               "--n_sweep_overrelax", str(a.n_overrelax), "--n_sweep_heatbath", str(a.n_heatbath)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            out[name] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        d = json.loads(lines[-1])
        out[name] = {"chains_per_gpu": batch, "samples": samples, "ms_per_sample": d["ms_per_sample"],
                     "value_per_gpu": d["updates_per_s"], "unit": "updates/s", "qoi_mean": d["qoi_mean"]}
    out["command"] = "host/driver --method throughput --action schwinger --Mt_lat %d --sampler heatbath --batch B" % size
    return out


def hbm_bound_probes(torch, ops, W, a, launches=20):
    """north_star asks for >= 60 % of the HBM roofline; the sweeps of the timed step are vector-issue bound (Philox, von Mises
    sampler) or temporally blocked, so the kernels of the path that ARE HBM bound are timed here, one launch at a
    time on the same state (1024^2 x B chains, HBM resident), with events on the stream they are launched on:
      a single overrelaxation sweep   quenchedschwingeraction.cc:57-65    state read + written once
      Action::evaluate                quenchedschwingeraction.cc:7-22     state read once
      QoIAvgPlaquette::evaluate       qoi/qft/qoiavgplaquette.cc:8-27     state read once (band reduction)
      Action::force                   quenchedschwingeraction.cc:68-89    state read, force written
    floor = the bytes the launch cannot avoid; counter = FETCH_SIZE (doubled, gfx950) + WRITE_SIZE of the same launch from
    the PMC passes of this kernel build (profiles/traffic.json), when there is one."""
    E = lambda: torch.cuda.Event(enable_timing=True)
    state_bytes = 8.0 * W.sites * W.B
    x, w = W.x, W.scratch
    sweep = [W.sweep]

    def or1():
        nonlocal x, w
        x, w = ops.lattice_sweep_draw_pingpong(W.act, x, w, 1, 0, a.seed, W.chain0, sweep[0], 1)
        sweep[0] += 1

    from mlmcpathintegral_amd import abi

    def or1_block():   # the same sweep by the register-block kernel: for ONE sweep it is the faster of the two (the closed form
        abi.set_option("MLMCPI_OR_KERNEL", "block")   # pays off from three sweeps per launch on, EXPERIMENTS 1.7)
        try:
            or1()
        finally:
            abi.set_option("MLMCPI_OR_KERNEL", os.environ.get("MLMCPI_OR_KERNEL", ""))

    probes = [("schwinger_perm_kernel" if W.perm else "schwinger_or_block_kernel<1>", "mlmcpi_lattice_sweep_draw (1 overrelaxation sweep, one launch)",
               "quenchedschwingeraction.cc:57-65", or1, 2.0 * state_bytes)]
    if W.perm:
        probes.append(("schwinger_or_block_kernel<1>", "mlmcpi_lattice_sweep_draw (1 overrelaxation sweep, one launch, MLMCPI_OR_KERNEL=block)",
                       "quenchedschwingeraction.cc:57-65", or1_block, 2.0 * state_bytes))
    probes += [
              ("schwinger_reduce_band_kernel", "mlmcpi_lattice_evaluate", "quenchedschwingeraction.cc:7-22",
               lambda: ops.lattice_evaluate(W.act, x), state_bytes),
              ("schwinger_reduce_band_kernel", "mlmcpi_qoi_avg_plaquette", "qoi/qft/qoiavgplaquette.cc:8-27",
               lambda: ops.qoi_avg_plaquette(x, W.size, W.size), state_bytes),
              ("schwinger_force_kernel", "mlmcpi_lattice_force", "quenchedschwingeraction.cc:68-89",
               lambda: ops.lattice_force(W.act, x), 2.0 * state_bytes)]
    out = []
    for kernel, entry, ref, fn, floor in probes:
        for _ in range(3):
            fn()
        e0, e1 = E(), E()
        e0.record()
        for _ in range(launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / launches
        rec = {"kernel": kernel, "entry_point": entry, "reference": ref, "launch_ms": ms, "floor_bytes": floor,
               "floor_GBps": floor / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": floor / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "reaches_60_percent": floor / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS >= 0.60, "counter_bytes": None, "counter_GBps": None}
        # (the two reductions are instances schwinger_reduce_band_kernel<op, .>: 2 = action, 3 = plaquette)
        tag = {"mlmcpi_lattice_evaluate": "schwinger_reduce_band_kernel<2,", "mlmcpi_qoi_avg_plaquette": "schwinger_reduce_band_kernel<3,"}.get(entry, kernel)
        pm = pmc_prefix_entry("probes", tag, chains=W.B, size=W.size)
        if pm:
            rec["counter_bytes"] = pm["hbm_bytes_per_launch"]
            rec["counter_GBps"] = pm["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9
        out.append(rec)
    W.x, W.scratch, W.sweep = x, w, sweep[0]
    return {"state": f"schwinger {W.size}x{W.size}, {W.B} chains ({2 * state_bytes / 2**30:.2f} GiB with scratch)", "hbm_peak_GBps": HBM_PEAK_GBS,
            "timing": f"{launches} launches between two events on the launch stream, per-launch average", "probes": out}


def fast_path_cliff(torch, abi, ops, a, rank, headline_rate):
    """The fused fast path (closed-form overrelaxation + step-envelope heat bath in one launch) needs 2 beta <= 16 (r05; 4 before)
    and a lattice of at least 128 x 128; elsewhere other kernels run (VERDICT r03 missing #5: no recorded rate there).  One
    point per side of the cliff, same step (10 + 1 sweeps + QoI + record_sample, one ABI call), rate against the headline."""
    pts = []
    for name, kind, Mt, Mx, beta, B, path in (
            ("beta = 4 (2 beta = 8)", "schwinger", 1024, 1024, 4.0, 32,
             "r05: the one-launch draw with the step-envelope heat bath, which serves concentrations up to 16 now (its tables' "
             "acceptance falls from 0.78 per attempt at 2 beta = 2 to 0.67 at 8: twice the cells on the list pass); r04: the "
             "wrapped-Cauchy instance, 0.58-0.66 of the headline"),
            ("beta = 6 (2 beta = 12)", "schwinger", 1024, 1024, 6.0, 32,
             "as beta = 4; the step envelope's eight classes are at their widest in kappa here (acceptance 0.59 per attempt, "
             "a fifth of the cells on the list pass)"),
            ("beta = 10 (2 beta = 20 > 16)", "schwinger", 1024, 1024, 10.0, 32,
             "the one-launch draw with the wrapped-Cauchy heat bath (schwinger_perm_heat_kernel<512, false>): that sampler costs "
             "1.3 x the step envelope per cell plus a pool round per colour phase, and the overrelaxation it stands beside got cheap"),
            ("960 x 960", "schwinger", 960, 960, 1.0, 32, "the one-launch draw (64 x 64 tiles divide the lattice)"),
            ("1024 x 992", "schwinger", 1024, 992, 1.0, 32,
             "r05: the one-launch draw on 64 x 64 tiles with the last tile row masked (r04: closed-form launch on 64 x 32 tiles + "
             "a heat-bath launch, 0.89-0.91)"),
            ("1000 x 1000 (no tile divides it)", "schwinger", 1000, 1000, 1.0, 32,
             "r05: the one-launch draw on 16 x 16 tiles of 64 x 64, last tile row and column masked (the plane wraps arbitrarily; "
             "1.05 x the lattice's work); r04: generic sweep kernels"),
            ("130 x 70 (no tile divides it)", "schwinger", 130, 70, 1.0, 4096,
             "r05: closed-form overrelaxation launch on 3 x 3 tiles of 64 x 32 with masked edges (2.0 x the lattice's work) + generic "
             "heat-bath kernel; r04: generic sweep kernels"),
            ("192 x 96", "schwinger", 192, 96, 1.0, 1024, "as 1024 x 992"),
            ("64 x 64", "schwinger", 64, 64, 1.0, 4096,
             "closed-form overrelaxation launch + generic heat-bath kernel (one 64 x 64 tile is the lattice: the fused "
             "launch's image would wrap around it twice)"),
            ("gff 1000 x 1000 (no tile divides it)", "gff", 1000, 1000, None, 256,
             "r05: gff_or_block_kernel<5, 32> + gff_or_heat_kernel<5, 32> on 32 x 32 tiles with masked edge tiles (1.05 x the lattice's "
             "work); r04: generic sweep kernels"),
            ("gff 96 x 96", "gff", 96, 96, None, 8192,
             "gff_or_block_kernel<5, 32> + gff_or_heat_kernel<5, 32>: register blocks on 32 x 32 tiles (r04; generic tiles before: 339 G/s)")):
        act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=beta) if kind == "schwinger" else abi.lattice_action(abi.GFF, Mt, Mx, mass=10.0)
        sites = (2 if kind == "schwinger" else 1) * Mt * Mx
        st = {"x": ops.lattice_initialise(act, B, a.seed, rank * B), "s": 0}
        st["w"] = torch.empty_like(st["x"])
        acc = torch.zeros((B, 5), dtype=torch.float64, device="cuda")

        def step():
            st["x"], st["w"], _ = ops.lattice_sweep_draw_qoi(act, st["x"], st["w"], st["x"], a.n_overrelax, a.n_heatbath, a.seed,
                                                             rank * B, st["s"], 1 if kind == "schwinger" else 3, 0, acc=acc)
            st["s"] += a.n_overrelax + a.n_heatbath
        for _ in range(12):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 10
        rate = sites * (a.n_overrelax + a.n_heatbath) * B / el
        pts.append({"point": name, "action": kind, "Mt": Mt, "Mx": Mx, "chains_per_gpu": B, "ms_per_step": 1e3 * el,
                    "value_per_gpu": rate, "unit": "updates/s", "kernels": path,
                    "over_headline": rate / headline_rate if kind == "schwinger" else None})
        if kind == "schwinger":
            # work of the closed-form launch / work of the lattice: edge tiles of a lattice the tiles do not divide are computed
            # whole and written in part (lattice2d.hip, sweep_draw_impl: 64 x 64 tiles where they divide Mx or both extents
            # reach 128, else 64 x 32)
            th = 64 if (Mx % 64 == 0 or (Mt >= 128 and Mx >= 128)) else 32
            pts[-1]["padding_factor"] = (-(-Mt // 64) * 64) * (-(-Mx // th) * th) / (Mt * Mx)
        if kind == "gff":   # against the committed line of the GFF workload (512 x 512, 1024 chains), when there is one
            try:
                tag = open(os.path.join(ROOT, "profiles", "FINAL")).read().strip()
                ref = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_gff.json")))["value"]
                pts[-1]["over_committed_gff_512_line"] = rate / ref
            except (OSError, ValueError, KeyError):
                pass
        del st, acc
    return pts


def random_order_rate(torch, abi, ops, a, rank):
    """The reference's DEFAULT sweep order (random_order = true, overrelaxedheatbathsampler.hh:27): every sweep walks a freshly
    shuffled index list through the site-at-a-time entry point (mlmcpi_lattice_site_updates: one thread per chain, sequential
    within a chain -- exact semantics).  Recorded so that a user who ports a reference parameter file knows the cost before
    running it (VERDICT r04 missing #5): same draw (10 + 1 sweeps) on a small lattice with many chains, beside the multicolour
    rate of the same shape."""
    Mt = Mx = 64
    B = 4096
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=1.0)
    n = 2 * Mt * Mx
    x = ops.lattice_initialise(act, B, a.seed, rank * B)
    gen = torch.Generator(device="cpu").manual_seed(871417)

    def draw(step0):
        for s in range(a.n_overrelax + a.n_heatbath):
            idx = torch.randperm(n, generator=gen, dtype=torch.int32).cuda()
            ops.lattice_site_updates(act, x, idx, s >= a.n_overrelax, a.seed, rank * B, step0 + s)
    draw(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for k in range(reps):
        draw((k + 1) * 11)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    w = torch.empty_like(x)
    st = {"x": x, "w": w}

    def colour():
        st["x"], st["w"], _ = ops.lattice_sweep_draw_qoi(act, st["x"], st["w"], st["x"], a.n_overrelax, a.n_heatbath, a.seed, rank * B, 100, 1, 0)
    for _ in range(3):
        colour()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        colour()
    torch.cuda.synchronize()
    elc = (time.perf_counter() - t1) / 10
    units = n * (a.n_overrelax + a.n_heatbath) * B
    return {"Mt": Mt, "Mx": Mx, "chains_per_gpu": B, "random_order_true": {"ms_per_draw": 1e3 * el, "value_per_gpu": units / el},
            "random_order_false": {"ms_per_draw": 1e3 * elc, "value_per_gpu": units / elc}, "unit": "updates/s",
            "slowdown": el / elc,
            "note": "random_order = true (the reference's default) is served by the site-at-a-time entry points only: one thread per "
                    "chain; the library default is false (multicolour sweeps), INTEGRATION.md section 5"}


class SweepWorkload:
    """OverrelaxedHeatBathSampler::draw + QoI + record_sample on a 2-D lattice action, B chains."""

    def __init__(self, a, torch, abi, ops, kind, size, B, chain0):
        self.a, self.torch, self.ops, self.kind, self.size, self.B, self.chain0 = a, torch, ops, kind, size, B, chain0
        if kind == "schwinger":
            self.act = abi.lattice_action(abi.SCHWINGER, size, size, beta=1.0)
            self.sites = 2 * size * size
        else:
            self.act = abi.lattice_action(abi.GFF, size, size, mass=10.0)
            self.sites = size * size
        self.x = ops.lattice_initialise(self.act, B, a.seed, chain0)
        self.scratch = torch.empty_like(self.x)
        self.acc = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
        # the library's launch plan for the overrelaxation sweeps (lattice2d.hip, sweep_draw_impl): where the 4 x 4
        # register-block kernel applies, up to 6 sweeps per launch in launches of equal depth (10 -> 5 + 5); otherwise
        # launches of 4 and a remainder (10 -> 4 + 4 + 2)
        self.blocks = (size % 64 == 0 and os.environ.get("MLMCPI_OR_KERNEL", "") in ("", "block", "perm")
                       and not os.environ.get("MLMCPI_SWEEP_TILE"))
        # Schwinger, 64 x 64 tiles: overrelaxation in closed form (schwinger_perm_kernel / schwinger_perm_heat_kernel), up to
        # 10 sweeps per launch, launches of equal depth; the last one takes the heat-bath sweep and the QoI along
        self.perm = kind == "schwinger" and self.blocks and os.environ.get("MLMCPI_OR_KERNEL", "") in ("", "perm")
        self.fuse = (min(a.fuse, 10) if a.fuse else 10) if self.perm else a.fuse or (6 if self.blocks else 4)
        self.plan = or_plan(a.n_overrelax, self.fuse, self.blocks)   # [(depth, launches), ...], at most two entries
        # The last overrelaxation launch of a draw takes the heat-bath sweep (and the QoI) along -- one launch of
        # schwinger_or_heat_kernel<depth> / gff_or_heat_kernel<depth> (lattice2d.hip, sweep_draw_impl): 4 x 4 register-block
        # geometry, depth <= 5, lattices >= 128, for the Schwinger action the step-envelope sampler (2 beta <= 16; beta = 1
        # here), not switched off by MLMCPI_OR_HEAT=split.
        self.or_heat = (self.blocks and a.n_heatbath == 1 and bool(self.plan) and (self.plan[-1][0] <= 5 or self.perm)
                        and size >= 128 and os.environ.get("MLMCPI_OR_HEAT", "") != "split" and not a.no_fused_qoi)
        self.or_heat_depth = self.plan[-1][0] if self.or_heat else 0
        # ... and with at most one workgroup of that launch per CU (one chain) the library puts the WHOLE draw into it
        # (6 <= n_overrelax <= 10, default fuse; sweep_draw_impl: whole_draw)
        whole = (self.or_heat and kind == "schwinger" and not self.perm and not a.fuse and 6 <= a.n_overrelax <= 10
                 and (size // 64) ** 2 * B <= 256 and os.environ.get("MLMCPI_OR_HEAT", "") != "narrow")
        self.whole = whole
        if whole:
            self.plan, self.or_heat_depth = [], a.n_overrelax
        elif self.or_heat:  # the launches that stay pure overrelaxation
            d, n = self.plan[-1]
            self.plan = self.plan[:-1] + ([(d, n - 1)] if n > 1 else [])
        self.sweep = 0
        self.ev = {"or": [], "rem": [], "hb": [], "qoi": []}

    def qoi(self):
        if self.kind == "schwinger":
            return self.ops.qoi_avg_plaquette(self.x, self.size, self.size)
        return self.ops.qoi_phi_squared(self.x)

    def lean_step(self, record=False):
        """The same step -- draw + QoI + record_sample -- as ONE ABI call (mlmcpi_lattice_sweep_draw_qoi_record: what the C++
        sampler's draw_with_qoi issues when it is handed the moments): no per-kernel events, no split of the draw.  Used for
        the side measurements, where a step is ~0.07 ms and the host cost of five event records and three extra calls would
        be a tenth of it."""
        a, ops, s = self.a, self.ops, self.sweep
        self.x, self.scratch, q = ops.lattice_sweep_draw_qoi(self.act, self.x, self.scratch, self.x, a.n_overrelax, a.n_heatbath,
                                                             a.seed, self.chain0, s, 1 if self.kind == "schwinger" else 3,
                                                             a.fuse, acc=self.acc)  # record_sample in the same call
        self.sweep = s + a.n_overrelax + a.n_heatbath

    def step(self, record):
        a, ops, s = self.a, self.ops, self.sweep
        E = lambda: self.torch.cuda.Event(enable_timing=True)
        if record:
            e = [E() for _ in range(5)]
            e[0].record()
        # same arithmetic and the same launches as one call with (n_overrelax, n_heatbath); split only to time the
        # kernels (ping-pong form: the buffers swap roles instead of being copied back).  Overrelaxation sweeps: one
        # call per launch depth of the library's plan.
        cur, oth, done = self.x, self.scratch, 0
        for k, (depth, launches) in enumerate(self.plan):
            cur, oth = ops.lattice_sweep_draw_pingpong(self.act, cur, oth, depth * launches, 0, a.seed, self.chain0,
                                                       s + done, depth)
            done += depth * launches
            if record:
                e[1 + k].record()
        if record:
            for k in range(len(self.plan), 2):
                e[1 + k] = e[k]  # no launch between them: the same point of the stream (an event record costs a few us)
        # fused QoI: one pass over the state less (-0.045 ms per step at 32 chains, -0.006 ms at one chain)
        self.fused = a.n_heatbath > 0 and not a.no_fused_qoi
        if self.fused:  # sampler->draw's last launch sums the QoI of the new sample while the tile is in LDS
            d = self.or_heat_depth  # > 0: that launch also holds the last d overrelaxation sweeps (same launches as one call)
            # ... and the launch that finishes the QoI records the sample too (mlmcpi_lattice_sweep_draw_qoi_record, the call
            # the C++ sampler's draw_with_qoi makes when it is handed the moments): no launch of its own for record_sample
            self.x, self.scratch, q = ops.lattice_sweep_draw_qoi(self.act, cur, oth, cur, d, a.n_heatbath, a.seed, self.chain0,
                                                                 s + a.n_overrelax - d, 1 if self.kind == "schwinger" else 3,
                                                                 0 if self.whole else (d or self.fuse), acc=self.acc)
        else:
            self.x, self.scratch = ops.lattice_sweep_draw_pingpong(self.act, cur, oth, 0, a.n_heatbath, a.seed, self.chain0,
                                                                   s + a.n_overrelax, self.fuse)
        if record:
            e[3].record()
        if not self.fused:
            ops.stats_accumulate(self.acc, self.qoi())  # qoi->evaluate + record_sample
        if record:
            e[4].record()
            self.ev["or"].append((e[0], e[1]))
            self.ev["rem"].append((e[1], e[2]))
            self.ev["hb"].append((e[2], e[3]))
            self.ev["qoi"].append((e[3], e[4]))
        self.sweep = s + a.n_overrelax + a.n_heatbath


def or_plan(n_overrelax, fuse, blocks):
    """Launch depths of n_overrelax overrelaxation sweeps as sweep_draw_impl issues them, grouped: [(depth, launches)]."""
    depths, rem = [], n_overrelax
    while rem:
        n = min(rem, fuse)
        if blocks:
            launches = -(-rem // fuse)
            n = -(-rem // launches)
        depths.append(n)
        rem -= n
    plan = []
    for d in depths:
        if plan and plan[-1][0] == d:
            plan[-1] = (d, plan[-1][1] + 1)
        else:
            plan.append((d, 1))
    assert len(plan) <= 2, plan
    return plan


def time_steps(torch, dist, world, step, steps, warmup):
    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    size = a.size or DEFAULT_SIZE[a.workload]
    B = a.chains or DEFAULT_CHAINS[a.workload]
    # A bench line is a record of the product library.  An experiment build (MLMCPI_LIB_VARIANT, abi.py) is timed only by
    # the A/B tools, which say so (--allow-variant) and get a line marked as not a record.
    variant = os.environ.get("MLMCPI_LIB_VARIANT", "")
    if variant and not a.allow_variant:
        if rank == 0:
            print("bench.py: MLMCPI_LIB_VARIANT is set (%r): a bench line records the product library only; "
                  "unset it, or pass --allow-variant from an A/B tool" % variant, file=sys.stderr)
        sys.exit(2)

    # Under a profiler (rocprofv3 preloads its tool library, which has initialised the GPU in this process already and
    # would be inherited by the children and mix their kernels into the same output directory) no child process is started.
    profiled = under_profiler()
    cpu = None
    if world == 1 and not a.no_cpu_baseline and not profiled:
        cpu = cpu_baseline(a, size)
    cxx = None
    if world == 1 and a.workload == "schwinger" and not a.no_extra_points and a.chains == 0 and not profiled:
        cxx = cxx_path(a, size)  # child processes; this one has not touched the GPU yet

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
    elif local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU (local rank {local}, {torch.cuda.device_count()} device(s))", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    coll_device = "cuda" if backend == "nccl" else "cpu"
    chain0 = rank * B  # global chain indices of this rank: [chain0, chain0 + B)

    # The statistics exchange of an N-rank run, set up and PROVEN here, on the main thread, before anything is timed:
    # the library's own communicator (libmlmcpi_rccl.so: ncclCommInitRank / ncclAllReduce through include/mlmcpi_comm.h,
    # the calls mlmcpi::RcclExchange makes for the C++ classes).  comm.establish ends the run on EVERY rank with status 3
    # (message on stderr) if any rank fails to build it or the check all-reduce does not return what N ranks must
    # produce, and with a traceback + status 1 if a collective hangs.  No fallback path exists.
    exchange, rccl = None, None
    if world > 1:
        from mlmcpathintegral_amd import comm
        if backend == "nccl":
            exchange, chk = comm.establish(rank, world, dist, torch,
                                           lambda: comm.make_rccl_exchange(rank, world, local, dist, torch),
                                           prepare=comm.open_runtime, agree_device="cuda")
            rccl = {"ranks": chk["ranks"], "lib": comm.library_path(), "runtime": comm.runtime_path(),
                    "allreduce_check": chk["allreduce_check"], "expected": chk["expected"],
                    "rendezvous": "128-byte id broadcast with torch.distributed"}
            # two RCCL communicators live in this process (torch's process group and the library's): say whether they
            # run on one librccl file or two -- the pairing the first multi-GPU run exercises
            mapped = comm.mapped_rccl_files()
            rccl["mapped_librccl"] = mapped
            rccl["one_runtime_for_both_communicators"] = len(mapped) == 1 and os.path.realpath(comm.runtime_path() or "") == mapped[0]
            if not rccl["one_runtime_for_both_communicators"]:
                rccl["note"] = ("torch.distributed and libmlmcpi_rccl.so bind different librccl files (or more than one is "
                                "mapped): two RCCL runtimes in one process")
        else:
            exchange, chk = comm.establish(rank, world, dist, torch, lambda: TorchExchange(torch, dist))
            rccl = {"ranks": chk["ranks"], "lib": None, "runtime": None, "allreduce_check": chk["allreduce_check"],
                    "expected": chk["expected"], "rehearsal": f"torch.distributed {backend}: NOT an RCCL run"}

    ev = lambda: torch.cuda.Event(enable_timing=True)
    events = []
    extra = {}

    if a.workload in ("schwinger", "gff"):
        W = SweepWorkload(a, torch, abi, ops, a.workload, size, B, chain0)
        for _ in range(a.thermalise):
            W.step(False)
        units_per_step = W.sites * (a.n_overrelax + a.n_heatbath) * B
        step, fuse = W.step, W.fuse
        acc_of = lambda: W.acc
    elif a.workload == "rotor_sweep":
        # OverrelaxedHeatBathSampler::draw on the rotor action, M_lat = 65536, a = 0.125
        act = abi.path_action(abi.ROTOR, size, size / 8.0, 0.25)
        x = ops.path_initialise(act, B, a.seed, chain0)
        scratch = torch.empty_like(x)
        acc = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
        units_per_step = size * (a.n_overrelax + a.n_heatbath) * B
        fuse = 1
        state = {"sweep": 0}

        state.update(x=x, w=scratch)

        def step(record):
            # sampler->draw, qoi->evaluate (topological susceptibility, summed while the segments of the last launch are in
            # LDS) and stats->record_sample in one ABI call: mlmcpi_path_sweep_draw_qoi
            if record:
                e0, e1 = ev(), ev()
                e0.record()
            state["x"], state["w"], _ = ops.path_sweep_draw_qoi(act, state["x"], state["w"], state["x"], a.n_overrelax, a.n_heatbath,
                                                                 a.seed, chain0, state["sweep"], acc=acc)
            state["sweep"] += a.n_overrelax + a.n_heatbath
            if record:
                e1.record()
                events.append((e0, e1))
        for _ in range(a.thermalise):
            step(False)
        acc_of = lambda: acc
    elif a.workload == "quartic_mlmc":
        # BASELINE configs[4]: quartic double well, 5 levels, finest M_lat = 32768, a = 0.125.
        # The (level, chain) instances are cut into `world` contiguous shares of equal cost (mlmc.partition); a step =
        # one Y sample per chain of every instance (2 trajectories of the level's sampler + the two-level step).
        from mlmcpathintegral_amd import mlmc
        n_level = 5
        est = mlmc.PathMLMC(abi.QUARTIC, size, size / 8.0, n_level, B, nt=a.nt, dt0=a.dt or 0.02, seed=a.seed, rank=rank,
                            world=world, n_sub=2, params=dict(lam=1.0, x0=1.0))
        est.exchange = exchange  # the level-table all-reduce goes through the same communicator
        est.thermalise(400)      # (r02 ran 64 trajectories here: the chains were still relaxing and the estimate read 0.4575
                                 # where the thermalised value is 0.598 -- see quartic_mlmc_hier's single-level reference)
        units_per_step = 0  # site-steps per step over all ranks: HMC of the feeding level + two-level pass
        for l in range(n_level):
            src = l if l == n_level - 1 else l + 1
            units_per_step += (2 * (a.nt + 1) * (size >> src) + (0 if l == n_level - 1 else (size >> l))) * B
        fuse = 1

        def step(record):
            if record:
                e0, e1 = ev(), ev()
                e0.record()
            est.pass_(1)  # sampler draws, two-level steps, QoIs and record_sample of every owned instance
            if record:
                e1.record()
                events.append((e0, e1))
        acc_of = lambda: est.packed_finest()
    elif a.workload == "quartic_mlmc_hier":
        # BASELINE configs[4] AS THE REFERENCE RUNS IT: sampler = 'hierarchical' -- HMC only on the
        # coarsest level (M_lat = 2048, nt = 100, dt = 0.095), two-level steps up (sampler/hierarchicalsampler.cc:55-81),
        # the coarse sampler of every level sub-sampled ceil(2 tau_int) draws apart (montecarlomultilevel.cc:170-190).
        # A step = one Y sample per chain of every level instance of this rank.
        from mlmcpathintegral_amd import mlmc
        n_level = 5
        T_hier = a.t_final or size / 8.0
        # dt0 = the step size of the direct (untimed) burn-in runs on the fine levels: the value the direct workload uses
        est = mlmc.PathMLMC(abi.QUARTIC, size, T_hier, n_level, B, nt=a.nt, dt0=0.02, seed=a.seed, rank=rank, world=world,
                            params=dict(lam=1.0, x0=1.0), hierarchical=True, dt_coarse=a.dt or 0.095)
        est.exchange = exchange
        est.thermalise(400, a.hier_sub_factor)   # untimed: every chain starts from an equilibrium sample of its level (direct HMC, once)
        sub = {l: lv.n_sub for l, lv in est.levels.items()}
        # site-steps per step of THIS rank's instances (HMC force evaluations on the coarsest level + two-level passes)
        units_rank = sum((lv.n_sub * lv.sampler.cost + (0 if lv.coarsest else lv.step.fine.M)) * lv.B for lv in est.levels.values())
        units_per_step = units_rank
        if world > 1:
            units_per_step = sum(exchange.allreduce_sum_host([float(units_rank)]))
        fuse = 1

        hier_work = {"start": None}   # site-steps at the first timed step: the draws between coarse samples follow the running tau_int

        def step(record):
            if record:
                if hier_work["start"] is None:
                    hier_work["start"] = (sum(lv.site_steps for lv in est.levels.values()), {l: lv.n_sub_sum for l, lv in est.levels.items()})
                e0, e1 = ev(), ev()
                e0.record()
            est.pass_(1)
            if record:
                e1.record()
                events.append((e0, e1))
        acc_of = lambda: est.packed_finest()
    else:
        kind = {"rotor_hmc": abi.ROTOR, "quartic_hmc": abi.QUARTIC, "ho_hmc": abi.HARMONIC}[a.workload]
        # a = 0.125 at the BASELINE sizes; config 1 (HO, M_lat = 128) keeps T_final = 4
        T_final = 4.0 if a.workload == "ho_hmc" else size / 8.0
        act = abi.path_action(kind, size, T_final, 0.25 if kind == abi.ROTOR else 1.0, 1.0, 1.0, 1.0)
        x = ops.path_initialise(act, B, a.seed, chain0)
        acc = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
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
                if record:
                    e1.record()
                ops.stats_accumulate(acc, q[:, -1].contiguous() if q.dim() > 1 else q)
            else:
                hmc.draw(x)
                if record:
                    e1.record()
                ops.stats_accumulate(acc, ops.qoi_susceptibility(x, T_final) if kind == abi.ROTOR else ops.qoi_xsquared(x))
            if record:
                events.append((e0, e1))
        acc_of = lambda: acc

    elapsed = time_steps(torch, dist, world, step, a.steps, a.warmup)
    if a.workload == "quartic_mlmc_hier":
        # the work actually done in the timed steps (the sub-sampling follows the running tau_int)
        done = sum(lv.site_steps for lv in est.levels.values()) - hier_work["start"][0]
        units_per_step = done / a.steps
        if world > 1:
            units_per_step = sum(exchange.allreduce_sum_host([float(units_per_step)]))
        sub = {l: (lv.n_sub_sum - hier_work["start"][1][l]) / a.steps for l, lv in est.levels.items()}

    # (side measurements on one rank only: at N > 1 they would only stagger the ranks in front of the statistics all-reduce)
    if a.workload == "schwinger" and world == 1 and (a.probes or (not a.no_extra_points and a.chains == 0)):
        extra["hbm_bound_probes"] = hbm_bound_probes(torch, ops, W, a)

    # side measurements of the default workload (same kernels, other batch sizes), after the timed region
    if a.workload == "schwinger" and world == 1 and not a.no_extra_points and a.chains == 0:
        # (a single-chain step is ~0.06 ms: 200 of them, so that the figure does not hang on a handful of launches)
        for name, b_extra, k_extra in (("single_chain", 1, max(10 * a.steps, 200)), ("chains_128", 128, max(2, a.steps // 4))):
            Wx = SweepWorkload(a, torch, abi, ops, "schwinger", size, b_extra, rank * b_extra)
            lean = a.n_heatbath > 0 and not a.no_fused_qoi
            for _ in range(min(a.thermalise, 10)):
                Wx.step(False)
            el = time_steps(torch, dist, 1, Wx.lean_step if lean else Wx.step, k_extra, 2 if b_extra > 1 else 10)
            extra[name] = {"chains_per_gpu": b_extra, "steps": k_extra, "ms_per_step": 1e3 * el / k_extra,
                           "value_per_gpu": Wx.sites * (a.n_overrelax + a.n_heatbath) * b_extra * k_extra / el,
                           "unit": "updates/s"}
            del Wx
        extra["fast_path_cliff"] = fast_path_cliff(torch, abi, ops, a, rank, units_per_step * a.steps / elapsed)
        extra["random_order"] = random_order_rate(torch, abi, ops, a, rank)
        extra["single_chain"]["note"] = "BASELINE configs[3] read literally: one chain per GPU (16 MiB state, cache resident)"

    # the one collective: packed per-chain moments of the QoI, summed over ranks through the exchange proven above; the
    # slowest rank's time travels in the same buffer (one slot per rank, max of the sums)
    packed = chains.pack_moments(acc_of()).cpu()
    collective = "none (one rank)"
    rank_times = None
    if world > 1:
        slots = [0.0] * world
        slots[rank] = elapsed
        red = exchange.allreduce_sum_host(packed.tolist() + slots)
        packed = torch.tensor(red[:packed.numel()], dtype=torch.float64)
        rank_times = list(red[packed.numel():])   # one slot per rank: every rank's own time of the K steps
        elapsed = max(rank_times)
        collective = (f"mlmcpi_comm_allreduce_sum_host_f64 (libmlmcpi_rccl.so: ncclAllReduce, {rccl['ranks']} ranks)"
                      if backend == "nccl" else f"torch.distributed all_reduce ({backend} rehearsal, {rccl['ranks']} ranks)")
    hier_run = None
    if a.workload == "quartic_mlmc_hier":
        # untimed continuation to the tolerance (or to a bound on the work): wall time to epsilon as the reference reports it
        t_eps, extra_passes = time.perf_counter(), 0
        mlmc_q, mlmc_e, mlmc_t = est.estimate()
        while mlmc_e > a.epsilon and extra_passes < 40:
            est.pass_(4)
            extra_passes += 4
            mlmc_q, mlmc_e, mlmc_t = est.estimate()
        torch.cuda.synchronize()
        hier_run = {"epsilon": a.epsilon, "reached": bool(mlmc_e <= a.epsilon), "extra_passes": extra_passes,
                    "seconds_after_timed_steps": time.perf_counter() - t_eps,
                    "samples_per_chain_and_level": a.warmup + a.steps + extra_passes}
        if rank == 0:  # single-level HMC on the finest lattice (untimed): what the telescoping sum must reproduce
            fine_act = abi.path_action(abi.QUARTIC, size, T_hier, 1.0, 1.0, 1.0, 1.0)
            xf = ops.path_initialise(fine_act, B, a.seed + 99, 0)
            hf = ops.PathHMC(fine_act, B, a.nt, 0.02, seed=a.seed + 99)
            ops.hmc_thermalise(hf, xf, 400)
            vals = []
            for _ in range(60):
                hf.draw(xf)
                vals.append(ops.qoi_xsquared(xf))
            cm = torch.stack(vals).mean(dim=0)
            ref_m, ref_e = float(cm.mean()), float(cm.std(unbiased=True)) / (B ** 0.5)
            hier_run["single_level_fine_hmc"] = {"mean": ref_m, "error": ref_e, "chains": B, "draws": 60,
                                                 "z": (mlmc_q - ref_m) / max(1e-300, (mlmc_e ** 2 + ref_e ** 2) ** 0.5)}
    if a.workload == "quartic_mlmc":
        mlmc_q, mlmc_e, mlmc_t = est.estimate()  # the level-table exchange, through est.exchange when world > 1
    qoi_mean = float(packed[1] / packed[0]) if float(packed[0]) > 0 else None

    if rank == 0:
        total_units = units_per_step * a.steps * (1 if a.workload.startswith("quartic_mlmc") else world)
        ms = lambda pairs: sum(p[0].elapsed_time(p[1]) for p in pairs)
        step_ms = 1e3 * elapsed / a.steps
        result = {
            "metric": "lattice-site-updates/sec",
            "value": total_units / elapsed,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "step_includes": ["sampler->draw", "qoi->evaluate", "stats->record_sample"],
            "stats_collective": collective,
            "rccl": rccl,
            "kernel_build": build_id(),
            "lib": os.path.relpath(abi.LIB_PATH, ROOT),
            "lib_sha256": lib_sha256(abi.LIB_PATH),
            # experiment builds lying beside the product library (they travel to the GPU box with it); none in a record run
            "variant_libs_present": sorted(f for f in os.listdir(os.path.dirname(abi.LIB_PATH))
                                           if f.startswith("libmlmcpi_hip_") and f.endswith(".so")),
        }
        if variant:
            result["variant"] = variant
            result["not_a_record"] = "experiment build loaded through MLMCPI_LIB_VARIANT (A/B tool run)"
        # per-rank rates (filled from the exchanged times below at N > 1): the first N-GPU run yields a checkable curve
        # (the MLMC workloads cut ONE job into equal-cost shares: no per-rank rate of its own)
        result["value_per_gpu"] = (None if a.workload.startswith("quartic_mlmc") else
                                   [units_per_step * a.steps / t for t in rank_times] if rank_times else [total_units / elapsed])
        if world > 1 and backend == "nccl":
            assert rccl and rccl.get("ranks") == world, f"RCCL communicator spans {rccl and rccl.get('ranks')} ranks, launcher started {world}"
        if a.workload in ("schwinger", "gff"):
            report_sweeps(result, a, W, size, B, world, step_ms, ms)
        elif a.workload == "rotor_sweep":
            launch_ms = ms(events) / a.steps
            # one read + one write of the state per launch.  Overrelaxation in closed form (the default; path_sweep_impl): up to
            # 16 sweeps per launch, and the heat-bath sweep rides on the last of them; sweep by sweep (MLMCPI_OR_KERNEL=block): 8
            # per launch, the heat bath on the last when that holds fewer than 8.  The QoI rides on the last launch of the draw.
            closed = os.environ.get("MLMCPI_OR_KERNEL", "") != "block"
            cap = 16 if closed else 8
            split = os.environ.get("MLMCPI_OR_HEAT", "") == "split" or (not closed and a.n_overrelax % 8 == 0)
            floor = 16.0 * size * B * ((a.n_overrelax + cap - 1) // cap + (a.n_heatbath if split else max(0, a.n_heatbath - 1)))
            result["config"] = {"workload": f"rotor M_lat={size}, {a.n_overrelax} overrelaxation + {a.n_heatbath} heat-bath "
                                            "sweeps per step, even/odd order", "chains_per_gpu": B, "chains_total": B * world,
                                "parallelism": f"chains sharded over {world} GPU(s), no data-path collective"}
            result["roofline"] = register_resident_roofline(
                ("rotor_sweep_kernel<true,true> (all sweeps of a step in one launch: overrelaxation in closed form, heat bath, QoI; "
                 "record_sample)" if closed and a.n_overrelax <= 16 and not split else
                 "rotor_sweep_kernel<false> + rotor_sweep_kernel<true,true> (all sweeps of a step, QoI and record_sample)"), launch_ms, floor,
                16.0 * units_per_step, pmc_entry("kernels_valu_busy", workload=a.workload, size=size, chains=B),
                ("the overrelaxation sweeps of a launch are one closed form (a sweep permutes the differences of the path) on "
                 "LDS-resident segments" if closed else "overrelaxation sweeps are fused 8 per launch on LDS-resident segments")
                + "; the heat-bath sweep is VALU bound (von Mises sampler)")
        elif a.workload == "quartic_mlmc":
            result["scaling"] = "strong"
            result["config"] = {"workload": f"quartic MLMC, 5 levels, finest M_lat={size}, a=0.125, nt={a.nt}, one Y sample per "
                                            "chain and level per step (2 sampler trajectories + two-level step)",
                                "chains_per_level": B, "instances_on_rank0": est.describe(),
                                "parallelism": f"(level, chain) instances cut into {world} equal-cost shares; per pass one "
                                               "all-reduce of the [5, 7] level table"}
            result["mlmc"] = {"estimate": mlmc_q, "error": mlmc_e, "level_means": mlmc_t[:, 1].tolist(),
                              "level_variances": mlmc_t[:, 2].tolist(),
                              "acceptance_rank0": {str(k): v for k, v in est.p_accept().items()}}
            launch_ms = ms(events) / a.steps
            floor = 16.0 * est.state_entries() * 2  # every owned state read and written once per trajectory, 2 per step
            result["roofline"] = register_resident_roofline(
                "hmc_trajectory_kernel (the level samplers; > 99 % of the site-steps)", launch_ms, floor,
                32.0 * units_per_step / world, pmc_entry("kernels_valu_busy", workload=a.workload, size=size, chains=B),
                "one step of all level instances of rank 0")
        elif a.workload == "quartic_mlmc_hier":
            result["scaling"] = "strong"
            result["config"] = {"workload": f"quartic MLMC as the reference runs it: 5 levels, finest M_lat={size}, a={T_hier / size:g}, "
                                            f"sampler='hierarchical' (HMC nt={a.nt}, dt={a.dt or 0.095} on the coarsest level "
                                            f"M_lat={size >> 4} only, two-level steps up), coarse samplers sub-sampled "
                                            "ceil(2 tau_int) draws apart; one Y sample per chain and level per step",
                                "chains_per_level": B, "instances_on_rank0": est.describe(), "sub_sampling_rank0": {str(k): v for k, v in sub.items()},
                                "parallelism": f"(level, chain) instances cut into {world} equal-cost shares; per pass one "
                                               "all-reduce of the [5, 7] level table"}
            result["mlmc"] = {"estimate": mlmc_q, "error": mlmc_e, "level_means": mlmc_t[:, 1].tolist(),
                              "level_variances": mlmc_t[:, 2].tolist(), "level_tau_int": mlmc_t[:, 3].tolist(),
                              "acceptance_rank0": {str(k): v for k, v in est.p_accept().items()},
                              "hierarchical_acceptance_rank0": {str(l): {str(k): round(v, 4) for k, v in lv.sampler.p_accept().items()}
                                                                for l, lv in est.levels.items()},
                              "run_to_epsilon": hier_run}
            # Levels whose two-level steps never accepted in this run: the chains of such a level do not move beyond their
            # (untimed, direct-HMC) initial sample, so the estimate and its error bar rest on one effective sample per
            # chain there and the throughput counts proposals that are never accepted.  Faithful to the reference's
            # configuration (T_final = 4096, a = 0.125: the fine-level fill-in is too wide to be accepted) -- said in the line.
            if hier_run and "single_level_fine_hmc" in hier_run:
                # distance of the telescoping sum from single-level HMC on the finest lattice, in combined standard errors
                result["mlmc"]["z_vs_single_level"] = hier_run["single_level_fine_hmc"]["z"]
                result["mlmc"]["z_vs_single_level_note"] = (
                    "beyond 2 sigma at a low hierarchical acceptance is the REFERENCE'S SCHEME, not the device: its two-level steps "
                    "are exact for independent coarse proposals, and ceil(2 tau_int) draws of a rarely accepting HierarchicalSampler "
                    "(long-tailed sojourn times) are not independent.  The reference-order restatement on the CPU (oracle MlmcRefO, "
                    "mt19937_64, sequential; tools/exp_hier_bias.py) shows the same bias at -28 sigma "
                    "(profiles/r05_hier_bias_reference_order_level0.json), and the device lands on the oracle's biased value "
                    "(tests/test_gpu_statistics.py::test_device_reproduces_the_bias_of_the_reference_scheme_at_low_hierarchical_acceptance)")
            result["mlmc"]["sub_sampling"] = "running ceil(2 tau_int) of the coarse sampler's QoI, re-read before every coarse sample (montecarlomultilevel.cc:170-190), autocovariances pooled over the batch"
            hacc = result["mlmc"]["hierarchical_acceptance_rank0"]
            frozen = sorted({int(k) for acc in hacc.values() for k, v in acc.items() if v == 0.0})
            result["mlmc"]["frozen_levels"] = frozen
            if frozen:
                result["mlmc"]["frozen_levels_note"] = (
                    "two-level acceptance 0 on these levels: their chains stay at the untimed direct-HMC sample; the z-score "
                    "against single-level HMC then checks that initialisation, not the hierarchical sampler, and 'epsilon "
                    "reached' comes from independent frozen chains.  See the T_final = M_lat / 32 line for moving chains")
            launch_ms = ms(events) / a.steps
            floor = 16.0 * sum(sub[l] * lv.B * lv.sampler.acts[-1].M for l, lv in est.levels.items())  # coarsest states, once per trajectory
            result["roofline"] = register_resident_roofline(
                "hmc_trajectory_kernel<1,R> (coarsest-level HMC of every hierarchical draw, M_lat = %d: R = 16 sites per lane, 2 waves per chain at 2048 chains; > 95 %% of the site-steps)" % (size >> 4), launch_ms, floor,
                32.0 * units_per_step / world, pmc_entry("kernels_valu_busy", workload=a.workload, size=size, chains=B),
                "one step of all level instances of rank 0")
        else:
            launch_ms = ms(events) / a.steps
            floor = 16.0 * size * B * draws_per_step  # the state read and written once per trajectory
            result["p_accept"] = float(hmc.n_accepted.double().mean()) / max(1, hmc.n_total)
            result["config"] = {"workload": f"{a.workload} M_lat={size}, nt={a.nt}, dt={hmc.dt}, fused trajectories",
                                "chains_per_gpu": B, "chains_total": B * world,
                                "parallelism": f"chains sharded over {world} GPU(s), no data-path collective"}
            result["roofline"] = register_resident_roofline(
                "hmc_chain_kernel" if draws_per_step > 1 else "hmc_trajectory_kernel", launch_ms, floor,
                32.0 * units_per_step, pmc_entry("kernels_valu_busy", workload=a.workload, size=size, chains=B),
                "state and momenta stay in registers for the whole trajectory")
        result["qoi_mean"] = qoi_mean
        result.update(extra)
        if cxx is not None:
            result["cxx_path"] = cxx
            if "value_per_gpu" in cxx.get("chains_32", {}) and B == 32:
                result["cxx_path"]["over_python_driven"] = cxx["chains_32"]["value_per_gpu"] / (result["value"] / world)
        if cpu is not None:
            p16 = cpu.pop("point_16", None)
            if a.workload in ("schwinger", "rotor_sweep"):
                # the CPU port executes all sweeps one by one; the closed form is an identity a CPU code could use as well, so
                # gpu_over_cpu mixes an algebraic shortcut with hardware speed (ms_per_draw is the like-for-like figure)
                cpu["closed_form_possible"] = True
                cpu["executes"] = "every sweep, update by update (reference order)"
            result["cpu_baseline"] = cpu
            if cpu.get("value"):  # against every core the box gives (cpu_baseline.cores of them)
                result["gpu_over_cpu"] = result["value"] / cpu["value"]
            if p16:
                result["cpu_baseline_16"] = p16
                result["gpu_over_cpu_16"] = result["value"] / p16["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        if hasattr(exchange, "close"):
            exchange.close()
        dist.destroy_process_group()


class TorchExchange:
    """Stand-in for comm.Comm in rehearsals of the N-rank path on a box with fewer GPUs than ranks
    (MLMCPI_BENCH_BACKEND=gloo): the same two calls over torch.distributed.  Never used by an RCCL run."""

    def __init__(self, torch, dist):
        self.torch, self.dist = torch, dist

    def allreduce_sum_host(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.tolist()

    def size(self):
        return self.dist.get_world_size()


def register_resident_roofline(kernel, launch_ms, floor_bytes, streaming_bytes, valu, note):
    """Roofline record of a kernel that keeps its state in registers / LDS across many updates.  HBM sees the state
    once per launch (`floor_bytes`), so the HBM fraction is small by construction and the binding limit is vector
    issue; the 32 (16) B-per-unit streaming model of SURVEY 8(d) is quoted as a rate, never as a fraction."""
    achieved = floor_bytes / (launch_ms * 1e-3) / 1e9
    r = {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "hbm_frac": achieved / HBM_PEAK_GBS, "hbm_floor_GBps": achieved, "traffic": None,
         "launch_ms": launch_ms, "algorithmic_bytes_per_launch": floor_bytes, "limited_by": "valu",
         "streaming_model_GBps": streaming_bytes / (launch_ms * 1e-3) / 1e9,
         "note": note + "; hbm_* counts the bytes an ideal implementation of this launch must move (state read "
                        "and written once); streaming_model_GBps is SURVEY 8(d)'s per-unit figure x units / time, a "
                        "rate for comparison with streaming implementations, not a fraction of any roofline"}
    if valu:
        # vector-issue utilisation of the library's kernels over the rocprofv3 SQ pass of THIS kernel build (time weighted):
        # SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time); not re-measured in this run
        r["valu"] = {"frac": valu["valu_frac"], "wave_insts": valu["wave_insts"], "kernel_ns": valu["kernel_ns"],
                     "peak": VALU_PEAK_WAVE_INSTS / 1e9, "unit": "G wave-insts/s", "dominant_kernel": valu.get("dominant_kernel"),
                     "source": valu.get("source")}
        r["valu_frac"] = valu["valu_frac"]
        if valu.get("issue_frac") is not None:
            # cost-weighted vector-issue utilisation (profile, build-hash tied): class counts (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32/64,
            # CVT, INT32, INT64) x the measured issue cycles of each class / (1024 SIMDs x 2.4 GHz x kernel time)
            r["issue_frac"] = valu["issue_frac"]
            r["valu"]["issue_frac"] = valu["issue_frac"]
        r["valu"]["from"] = "profile (PMC passes of tools/profile_all.sh on this kernel build), not measured in this run"
        # the binding resource is vector issue (cost-weighted when the class counters were taken): kept under roofline.valu;
        # bound / achieved / peak / unit / frac stay on the measured HBM model in every line (schema 2)
        f = valu["issue_frac"] if valu.get("issue_frac") is not None else valu["valu_frac"]
        r["valu"].update({"binding_frac": f, "binding_achieved_Gcycle_per_s": f * 1024 * 2.4, "binding_peak_Gcycle_per_s": 1024 * 2.4})
    r["schema"] = 2
    return r


def report_sweeps(result, a, W, size, B, world, step_ms, ms):
    """Per-kernel records and the roofline of the dominant kernel for the 2-D sweep workloads."""
    fuse, sites = W.fuse, W.sites
    state_rw = 16.0 * sites * B          # one read + one write of the whole state: the HBM floor of ANY launch
    special = size % 64 == 0 and fuse <= (6 if (a.workload == "schwinger" or W.blocks) else 4) and not os.environ.get("MLMCPI_SWEEP_TILE")
    lds_kernel = os.environ.get("MLMCPI_OR_KERNEL") == "lds"

    def or_name(depth):
        if a.workload == "schwinger":
            if W.perm:
                return f"schwinger_perm_kernel (K = {depth})"
            if not special:
                return "schwinger_sweep_kernel<false,256>"
            if W.blocks:
                return f"schwinger_or_block_kernel<{depth}>"
            return (f"schwinger_or_patch_kernel<{depth}>" if depth <= 4 and not lds_kernel
                    else f"schwinger_or_kernel<64,32,{depth},{1024 if depth >= 4 else 512}>")
        if not special:
            return "gff_sweep_kernel<false,256>"
        if W.blocks:
            return f"gff_or_block_kernel<{depth}>"
        return f"gff_or_patch_kernel<{depth}>" if not lds_kernel else f"gff_or_kernel<64,32,{depth},256>"

    if a.workload == "schwinger":
        fixed = size % 64 == 0 and size >= 128 and a.n_heatbath == 1 and not os.environ.get("MLMCPI_SWEEP_TILE")
        hb_name = "schwinger_sweep_kernel<true,256,64,32>" if fixed else "schwinger_sweep_kernel<true,256,0,0>"
    else:
        fixed = size % 64 == 0 and size >= 128 and a.n_heatbath == 1 and not os.environ.get("MLMCPI_SWEEP_TILE")
        hb_name = "gff_sweep_kernel<true,256,64,32>" if fixed else "gff_sweep_kernel<true,256,0,0>"
    result["config"] = {"workload": f"{a.workload} {size}x{size}, {a.n_overrelax} overrelaxation + {a.n_heatbath} heat-bath "
                                    "sweeps + QoI + record_sample per step, multicolour order",
                        "chains_per_gpu": B, "chains_total": B * world, "fuse": fuse,
                        "overrelaxation_launches": [d for d, n in W.plan for _ in range(n)] + ([W.or_heat_depth] if W.or_heat else []),
                        "last_overrelaxation_launch_holds_the_heat_bath": W.or_heat,
                        "parallelism": f"chains sharded over {world} GPU(s), no data-path collective"}
    kernels = []

    def record(name, role, pairs, launches, sweeps, floor_bytes, traffic, valu):
        launch_ms = ms(pairs) / (a.steps * launches)
        alg = 16.0 * sites * B * sweeps  # 16 B per update x updates of one launch
        k = {"kernel": name, "role": role, "launches_per_step": launches, "sweeps_per_launch": sweeps, "launch_ms": launch_ms,
             "share_of_step": launch_ms * launches / step_ms, "updates_per_s": sites * B * sweeps / (launch_ms * 1e-3),
             "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg / (launch_ms * 1e-3) / 1e9,
             # the bound: bytes this launch cannot avoid moving (state in, state out) against the HBM peak
             "hbm_floor_bytes_per_launch": floor_bytes, "hbm_floor_GBps": floor_bytes / (launch_ms * 1e-3) / 1e9,
             "hbm_frac": floor_bytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "traffic": traffic["hbm_bytes_per_launch"] if traffic else None}
        if traffic:
            k["traffic_over_floor"] = traffic["hbm_bytes_per_launch"] / floor_bytes
            k["traffic_GBps"] = traffic["hbm_bytes_per_launch"] / (launch_ms * 1e-3) / 1e9
        if valu:
            insts = valu["SQ_INSTS_VALU_per_launch"] * B / valu["chains"]
            k["valu_wave_insts_per_launch"] = insts
            k["valu_insts_per_update"] = insts * 64 / (sites * B * sweeps)
            k["valu_frac"] = insts / (launch_ms * 1e-3) / VALU_PEAK_WAVE_INSTS   # every instruction charged 4 cycles
            k["valu_from"] = "profile (SQ counters of this kernel build, profiles/traffic.json), launch time of this run"
            if valu.get("issue"):
                # cost-weighted issue bound: the launch's instruction mix by class x the measured issue cycles of each class
                # (profiles/r02_valu_issue_cost.txt) / (1024 SIMDs x 2.4 GHz x launch time of THIS run)
                cyc = valu["issue"]["issue_cycles"] * B / valu["chains"]
                k["issue_frac"] = cyc / (launch_ms * 1e-3) / (1024 * 2.4e9)
                k["issue_cycles_per_inst"] = valu["issue"]["mean_cycles_per_inst"]
                k["valu_mix"] = {c: v * B / valu["chains"] for c, v in valu["issue"]["mix"].items()}
        kernels.append(k)
        return k

    wl = dict(workload=a.workload, size=size)
    for k, (depth, launches) in enumerate(W.plan):
        record(or_name(depth), f"{depth} fused overrelaxation sweeps", W.ev["or" if k == 0 else "rem"], launches, depth, state_rw,
               pmc_entry("entries", chains=B, fuse=depth, kind="overrelax", **wl),
               pmc_entry("valu", kind="overrelax", fuse=depth, **wl))
    fused = getattr(W, "fused", False)
    if W.or_heat:
        d = W.or_heat_depth
        record(f"schwinger_perm_heat_kernel<{1024 if (size // 64) ** 2 * B <= 256 else 512}, true> (K = {d})" if W.perm
               else f"{a.workload}_or_heat_kernel<{d}>",
               (f"{d} overrelaxation sweeps in closed form (one fixed permutation of the plaquettes)" if W.perm
                else f"{d} fused overrelaxation sweeps") + " + heat-bath sweep + qoi->evaluate in one launch "
               "(QoI summed while the tile is in LDS)", W.ev["hb"], 1, d + 1, state_rw,
               pmc_entry("entries", chains=B, fuse=d + 1, kind="or_heat", **wl), pmc_entry("valu", kind="or_heat", fuse=d + 1, **wl))
    elif a.n_heatbath:
        record(hb_name, "heat-bath sweep", W.ev["hb"], a.n_heatbath, 1, state_rw,
               pmc_entry("entries", chains=B, fuse=1, kind="heatbath", **wl), pmc_entry("valu", kind="heatbath", **wl))
        if fused:
            kernels[-1]["role"] = "heat-bath sweep + qoi->evaluate (QoI summed while the tile is in LDS)"
    if fused:
        # no launch of its own: lattice_finish_kernel (one wave per chain sums the tiles' partial QoIs and updates the chain's
        # moments) is inside the interval of the launch above
        kernels[-1]["interval_includes"] = "lattice_finish_kernel (one wave per chain: the tiles' partial QoIs summed, the chain's moments updated)"
        result["record_sample_launch"] = "none of its own: the launch that finishes the QoI updates the moments (mlmcpi_lattice_sweep_draw_qoi_record)"
    else:
        qk = record(("schwinger_reduce_band_kernel" if a.workload == "schwinger" else "lattice_reduce_kernel") + " (QoI) + stats_accumulate_kernel",
                    "qoi->evaluate + record_sample", W.ev["qoi"], 1, 1, 0.5 * state_rw, None, None)
        del qk["updates_per_s"], qk["algorithmic_bytes_per_launch"], qk["algorithmic_GBps"], qk["sweeps_per_launch"]
    result["kernels"] = kernels
    # roofline: the kernel with the largest share of the step.  `achieved` = algorithmic bytes (16 B x the updates of one
    # launch) / launch time; for a single-sweep launch that equals the HBM floor, for a fused launch the floor is used
    # (a fused launch shares one HBM round trip among its sweeps, so the per-update model is not a bound for it).
    body = kernels if fused else kernels[:-1]   # (not fused: the last entry is the QoI pass + record_sample)
    dom = max(body, key=lambda k: k["share_of_step"])
    result["qoi_fused_into_draw"] = fused
    # What binds the dominant kernel.  The heat-bath launches are vector-issue bound (Philox + von Mises sampler, SURVEY
    # A.2: the contract's "hbm" | "mfma" has no word for it, so it is called what it is): frac = issue_frac = the launch's
    # instruction mix by class x the measured issue cycles of each class / (1024 SIMDs x 2.4 GHz x launch time); the HBM
    # figures stay beside it.  Without a PMC profile of THIS kernel build the issue bound cannot be quoted, and the record
    # falls back to the HBM floor (and says so).
    roof = {"kernel": f"{dom['kernel']} ({dom['role']})", "hbm_frac": dom["hbm_frac"], "hbm_floor_GBps": dom["hbm_floor_GBps"],
            "hbm_peak_GBps": HBM_PEAK_GBS, "traffic": dom["traffic"], "launch_ms": dom["launch_ms"],
            "algorithmic_bytes_per_launch": dom["hbm_floor_bytes_per_launch"], "share_of_step": dom["share_of_step"],
            "updates_per_s": dom["updates_per_s"]}
    # The contract fields (bound, achieved, peak, unit, frac) stay on the measured HBM model in every line, so that frac is
    # comparable across rounds and with north_star's 60 %: the state read + written once per launch (the bytes no
    # implementation of this launch can avoid) / the launch time measured in this run / 8 TB/s.  What actually binds a
    # vector-issue-bound launch sits beside it under roofline.valu.
    roof.update({"bound": "hbm", "achieved": dom["hbm_floor_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["hbm_frac"],
                 "schema": 2})
    if "issue_frac" in dom:
        peak = 1024 * 2.4   # G issue cycles per second: 1024 SIMDs x 2.4 GHz
        roof["limited_by"] = "valu"
        roof["valu"] = {"frac": dom["issue_frac"], "achieved": dom["issue_frac"] * peak, "peak": peak, "unit": "Gcycle/s",
                        "issue_cycles_per_inst": dom["issue_cycles_per_inst"], "valu_frac": dom["valu_frac"],
                        "valu_insts_per_update": dom["valu_insts_per_update"], "from": dom["valu_from"],
                        "note": "vector-issue model, NOT measured in this run: instruction counts by class from the SQ counters "
                                "of this kernel build (profiles/traffic.json) x measured issue cycles per class "
                                "(profiles/r04_valu_issue_cost.txt) / (1024 SIMDs x 2.4 GHz x the launch time of this run); "
                                "valu_frac = the same with every instruction charged 4 cycles"}
        roof["note"] = ("frac is the HBM floor fraction (all sweeps of the launch share one HBM round trip); the launch is "
                        "vector-issue bound: see roofline.valu")
    elif "heat-bath sweep" in dom["role"]:
        roof["limited_by"] = "valu"
        roof["note"] = ("this launch is vector-issue bound, but profiles/traffic.json holds no PMC figures of kernel build "
                        + build_id() + ": only the HBM floor fraction can be quoted")
    result["roofline"] = roof
    # whole step: 16 B x every update of the step against the step time, and the bytes the step's launches cannot avoid
    alg_step = 16.0 * sites * B * (a.n_overrelax + a.n_heatbath)
    floor_step = state_rw * (sum(n for _, n in W.plan) + a.n_heatbath) + (0.0 if fused else 0.5 * state_rw)  # W.plan: the pure overrelaxation launches
    result["whole_step"] = {"algorithmic_bytes": alg_step, "algorithmic_GBps": alg_step / (step_ms * 1e-3) / 1e9,
                            # SURVEY 8(d)'s streaming model (16 B per update) against the HBM peak: a rate for comparison
                            # with one-pass-per-sweep implementations, NOT a roofline fraction -- fused launches share one
                            # HBM round trip among their sweeps, so it may pass 1; hbm_floor_frac below is the bounded one
                            "algorithmic_over_hbm_peak": alg_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "hbm_floor_bytes": floor_step, "hbm_floor_GBps": floor_step / (step_ms * 1e-3) / 1e9,
                            "hbm_floor_frac": floor_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "kernel_ms_sum": sum(k["launch_ms"] * k["launches_per_step"] for k in kernels)}
    if all(k["traffic"] is not None for k in body):
        tr = sum(k["traffic"] * k["launches_per_step"] for k in body)
        result["whole_step"].update({"counter_traffic_bytes": tr, "counter_traffic_GBps": tr / (step_ms * 1e-3) / 1e9,
                                     "counter_traffic_frac": tr / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    # What the headline number is.  `value` counts every link update of every sweep of the draw; where the
    # overrelaxation sweeps of a launch are applied in closed form (one fixed permutation of the plaquettes, same state to
    # 4e-14), ten of the eleven sweeps are not EXECUTED as link updates, so the comparable quantities are the time per draw
    # and the updates the launch does execute one by one (the heat-bath sweeps).
    closed_form = a.workload == "schwinger" and bool(getattr(W, "perm", False))
    executed = sites * B * (a.n_heatbath if closed_form else a.n_overrelax + a.n_heatbath)
    result["ms_per_draw"] = step_ms
    result["draws_per_s"] = B * world / (step_ms * 1e-3)
    result["executed_updates_per_s"] = executed * world / (step_ms * 1e-3)
    result["equivalent_updates"] = {"value_counts_equivalent_updates": closed_form,
                                    "sweeps_per_draw": {"overrelaxation": a.n_overrelax, "heat_bath": a.n_heatbath},
                                    "executed_as_link_updates": {"overrelaxation": 0 if closed_form else a.n_overrelax,
                                                                 "heat_bath": a.n_heatbath},
                                    "note": ("value = link updates of all %d sweeps / time; the %d overrelaxation sweeps are one "
                                             "closed-form map per launch (identical state, tests), so compare ms_per_draw / "
                                             "draws_per_s across implementations and executed_updates_per_s for sampler work"
                                             % (a.n_overrelax + a.n_heatbath, a.n_overrelax)) if closed_form else
                                            "every sweep is executed update by update"}
    result["whole_step"]["note"] = ("algorithmic_over_hbm_peak may pass 1: that is temporal blocking (the sweeps of a launch share "
                                    "one HBM round trip), not skipped work -- the state a launch writes is the state after every one "
                                    "of its sweeps: the sweep-by-sweep kernels agree bit for bit whatever the launch plan, and the "
                                    "Schwinger overrelaxation in closed form (K sweeps = one fixed permutation of the plaquettes) "
                                    "agrees with them and with the oracle's sweeps to 4e-14 (tests); what HBM really carries is "
                                    "counter_traffic_* (FETCH_SIZE doubled as for 16-byte streaming reads: an upper bound where "
                                    "part of the reads are 8 bytes per lane)")


if __name__ == "__main__":
    main()
