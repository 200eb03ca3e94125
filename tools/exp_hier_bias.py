#!/usr/bin/env python3
"""The reference's multilevel scheme in reference order on the CPU (oracle MlmcRefO: HierarchicalSampler coarse samplers
sub-sampled ceil(2 tau_int) draws apart with the RUNNING tau_int, montecarlomultilevel.cc:170-190; TwoLevelMetropolisStep
fed with them) against single-level HMC chains of EVERY level: does the scheme itself carry the bias the device line shows
at a low hierarchical acceptance, and on which level?  Per level l the mean of the FINE part of Y_l (the two-level chain's
own QoI) is compared with single-level HMC on M_l, the coarse part (the sub-sampled hierarchical sampler's QoI) with HMC on
M_{l+1}.  Independent replicas (seed offsets) over the host cores.

    python tools/exp_hier_bias.py [--M 1024] [--T 256] [--levels 3] [--samples 60000] [--replicas 8] [--sub 0,240] [--only-level 0]
"""
import argparse, json, multiprocessing as mp, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

QUARTIC = 1


def mlmc(args):
    import oracle as O
    a, sub, rep = args
    L = O.lib()
    out, acc = np.zeros(8 * a.levels), np.zeros(a.levels)
    L.orc_mlmc_ref_run(QUARTIC, a.M, a.T, 1.0, 1.0, 1.0, 1.0, a.levels, a.nt, a.dt, a.window, sub, a.burnin, a.samples, a.only_level,
                       1000 + 17 * rep, out, acc)
    return out.reshape(a.levels, 8), acc


def single(args):
    import oracle as O
    a, level, rep = args
    L = O.lib()
    out = np.zeros(5)
    # dt scaled with the lattice spacing's effect on the acceptance: tuned by hand per level (p_accept in the output)
    dt = a.dt * (0.5 ** (a.levels - 1 - level)) ** 0.5
    L.orc_single_level_ref_run(QUARTIC, a.M >> level, a.T, 1.0, 1.0, 1.0, 1.0, 1, a.nt, dt, a.window, a.burnin, a.samples_single, 5000 + 13 * rep + 977 * level, out)
    return out


def stat(v):
    v = np.asarray(v)
    return float(v.mean()), float(v.std(ddof=1) / np.sqrt(len(v)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=1024)
    ap.add_argument("--T", type=float, default=256.0)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--nt", type=int, default=100)
    ap.add_argument("--dt", type=float, default=0.1, help="HMC step on the coarsest level")
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--burnin", type=int, default=2000)
    ap.add_argument("--samples", type=int, default=60000)
    ap.add_argument("--samples-single", type=int, default=100000)
    ap.add_argument("--replicas", type=int, default=8)
    ap.add_argument("--only-level", type=int, default=-1)
    ap.add_argument("--sub", default="0", help="comma list of sub-sampling modes: 0 = reference (running ceil(2 tau_int)), n = n draws apart")
    a = ap.parse_args()
    import oracle as O
    O.build()
    res = {"params": vars(a)}
    t0 = time.time()
    with mp.get_context("fork").Pool(min(a.replicas, os.cpu_count() or 1)) as pool:
        hmc = []
        for level in range(a.levels):
            r = np.array(pool.map(single, [(a, level, rep) for rep in range(a.replicas)]))
            m, e = stat(r[:, 0])
            hmc.append({"M": a.M >> level, "mean": m, "error": e, "tau_int": float(r[:, 2].mean()), "p_accept": float(r[:, 4].mean())})
        res["single_level_hmc"] = hmc
        print(json.dumps(hmc), flush=True)
        res["mlmc"] = []
        for sub in [int(x) for x in a.sub.split(",")]:
            out = pool.map(mlmc, [(a, sub, r) for r in range(a.replicas)])
            tab = np.array([o[0] for o in out])          # [replica, level, 8]
            acc = np.array([o[1] for o in out])
            rec = {"sub_mode": sub, "levels": []}
            for l in range(a.levels):
                if a.only_level >= 0 and l != a.only_level:
                    continue
                ym, ye = stat(tab[:, l, 0])
                fm, fe = stat(tab[:, l, 6])
                lv = {"level": l, "Y": ym, "Y_error": ye, "tau_int_Y": float(tab[:, l, 2].mean()),
                      "draws_between_coarse_samples": float(tab[:, l, 4].mean()), "twolevel_acceptance": float(tab[:, l, 5].mean()),
                      "feeding_sampler_acceptance": float(acc[:, l].mean()), "fine_part": fm, "fine_part_error": fe,
                      "z_fine_vs_hmc": (fm - hmc[l]["mean"]) / np.hypot(fe, hmc[l]["error"])}
                if l + 1 < a.levels:
                    cm, ce = stat(tab[:, l, 7])
                    lv.update({"coarse_part": cm, "coarse_part_error": ce, "z_coarse_vs_hmc": (cm - hmc[l + 1]["mean"]) / np.hypot(ce, hmc[l + 1]["error"]),
                               "Y_expected": hmc[l]["mean"] - hmc[l + 1]["mean"],
                               "z_Y": (ym - (hmc[l]["mean"] - hmc[l + 1]["mean"])) / np.sqrt(ye ** 2 + hmc[l]["error"] ** 2 + hmc[l + 1]["error"] ** 2)})
                rec["levels"].append(lv)
            if a.only_level < 0:
                est = tab[:, :, 0].sum(axis=1)
                rec["estimate"], rec["error"] = stat(est)
                rec["z_vs_hmc"] = (rec["estimate"] - hmc[0]["mean"]) / np.hypot(rec["error"], hmc[0]["error"])
            res["mlmc"].append(rec)
            print(json.dumps(rec), flush=True)
    res["seconds"] = time.time() - t0
    print(json.dumps(res))


if __name__ == "__main__":
    main()
