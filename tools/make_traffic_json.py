#!/usr/bin/env python3
"""profiles/traffic.json from the summary of tools/profile_all.sh (gpurun_out/prof_<tag>/summary.json): the per-launch PMC
figures bench.py quotes, each tagged with the kernel build it was measured on (bench.py quotes none from another build).
   python tools/make_traffic_json.py gpurun_out/prof_r02 r02"""
import json, os, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = json.load(open(os.path.join(src, "summary.json")))
# Issue cost of a wave64 VALU instruction by class, in shader cycles per SIMD.  Two sources, the smaller of the two per class:
# (a) tools/valu_issue_bench.hip (profiles/r02_valu_issue_cost.txt; 8 waves per SIMD, independent instructions, time x
#     2.4 GHz): v_add_f32 2.6, v_fma_f32 2.8, xor / add_u32 / shift 2.3, mad_u64_u32 / mul / bfe / and_or / cndmask / cmp /
#     cvt / rounding 4.1 ... 4.5, fp64 add / mul / fma 4.7 / 5.0 / 5.2, v_log_f32 8.5, v_rcp_f64 16.9;
# (b) the SQ's own accounting: SQ_ACTIVE_INST_VALU (quad-cycles) equals SQ_INSTS_VALU to 1 % on every kernel of this
#     library, fp64 streams included -- one quad-cycle = 4 cycles per instruction; and the densest fp64 stream measured
#     (rotor hmc_trajectory_kernel: 4.14e9 instructions in 7.6 ms) runs at 4.5 cycles per instruction ALL IN at 2.4 GHz,
#     i.e. the microbenchmark's fp64 figures (taken at the clock the chip holds under a pure fp64 load, which the
#     conversion to 2.4 GHz overstates) are too high.  So the non-transcendental classes are capped at 4.0.
# INT32 is a mix of 2.3-cycle (xor, add, shift) and 4-cycle operations: the Philox rounds that dominate it are 40 of the
# former to 20 of the latter = 2.9.  `other` = what the class counters do not see (moves, selects, compares, rounding).
ISSUE_CYCLES = {"ADD_F32": 2.6, "MUL_F32": 2.8, "FMA_F32": 2.8, "TRANS_F32": 8.5, "ADD_F64": 4.0, "MUL_F64": 4.0, "FMA_F64": 4.0,
                "TRANS_F64": 16.9, "CVT": 4.0, "INT32": 2.9, "INT64": 4.0, "other": 4.0}


def issue_model(k):
    """cost-weighted vector-issue time of one launch: sum over classes of count x cycles (wave-instructions x cycles per SIMD)"""
    if "SQ_INSTS_VALU_FMA_F64" not in k or "SQ_INSTS_VALU_INT32" not in k:
        return None
    mix = {c: k.get("SQ_INSTS_VALU_" + c, 0.0) for c in ISSUE_CYCLES if c != "other"}
    mix["other"] = max(0.0, k["SQ_INSTS_VALU"] - sum(mix.values()))
    cycles = sum(mix[c] * ISSUE_CYCLES[c] for c in mix)
    return {"mix": mix, "issue_cycles": cycles, "mean_cycles_per_inst": cycles / k["SQ_INSTS_VALU"]}


SIZES = {"schwinger": 1024, "gff": 512, "rotor_hmc": 65536, "quartic_hmc": 32768, "ho_hmc": 128, "quartic_mlmc": 32768, "quartic_mlmc_hier": 32768, "rotor_sweep": 65536}
CHAINS = {"schwinger": 32, "gff": 1024, "rotor_hmc": 1024, "quartic_hmc": 2048, "ho_hmc": 8192, "quartic_mlmc": 512, "quartic_mlmc_hier": 2048, "rotor_sweep": 1024}
out = {"_how": "tools/profile_all.sh: rocprofv3 --pmc <counters> --kernel-trace on `python3 bench.py --workload W --steps 5 --warmup 2 "
               "--no-cpu-baseline --no-extra-points`, one pass per counter group (SQ group; FETCH_SIZE; WRITE_SIZE), per-launch "
               "averages over every launch of the run.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 counts a "
               "128-byte request of a 16-byte-per-lane streaming read as 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE is "
               "exact.  `build` = hash of the kernel sources (bench.py build_id()).  Raw summary: profiles/%s_profile_summary.json." % tag,
       "entries": [], "valu": [], "kernels_valu_busy": [], "probes": []}
# The kernels bench.py's TIMED region spends its time in, by workload (name substrings).  The profiled run also holds
# what precedes the timed steps -- thermalisation with other kernels or other shapes (ho_hmc: 32 single-trajectory launches
# at a sixth of the chain kernel's utilisation; quartic_mlmc_hier: direct HMC on every level) -- and a utilisation averaged
# over all of it describes the warm-up, not the measurement (r03 first reported 0.34 for ho_hmc, whose chain kernel runs at
# 0.80, and 0.58 for quartic_mlmc_hier, whose coarsest-level kernel runs at 0.44).
TIMED = {"ho_hmc": ["hmc_chain_kernel"], "rotor_hmc": ["hmc_trajectory_kernel"], "quartic_hmc": ["hmc_trajectory_kernel"],
         "quartic_mlmc": ["hmc_trajectory_kernel"],
         # the coarsest-level draws: M_lat = 2048, 2048 chains, 16 sites per lane = 128 threads per chain (the direct-HMC
         # thermalisation of the finer levels runs the same kernel name on larger grids: profile_summarise.py keys by grid)
         "quartic_mlmc_hier": ["hmc_trajectory_kernel<1, 16> grid=262144"]}
def timed_match(t, name):
    """a TIMED pattern with a grid size names exactly one (kernel, grid) view; others are substrings of the kernel name"""
    return name.endswith(t) if "grid=" in t else t in name


for w, e in S.items():
    build = (e.get("bench_profiled") or {}).get("kernel_build")
    tot_insts = tot_ns = tot_cycles = 0.0
    all_insts = all_ns = 0.0
    timed = {}
    for name, k in e["kernels"].items():
        im = issue_model(k)
        if k.get("by_grid") and not (w in TIMED and any("grid=" in t and timed_match(t, name) for t in TIMED[w])):
            continue   # per-grid views of a kernel that is also listed under its plain name: used only when asked for by grid
        if "SQ_INSTS_VALU" in k and "avg_ns" in k:
            if not k.get("by_grid"):
                all_insts += k["SQ_INSTS_VALU"] * k["calls"]
                all_ns += k["avg_ns"] * k["calls"]
            if w in TIMED and not any(timed_match(t, name) for t in TIMED[w]):
                continue
            timed[name] = k["avg_ns"] * k["calls"]
            tot_insts += k["SQ_INSTS_VALU"] * k["calls"]
            tot_ns += k["avg_ns"] * k["calls"]
            tot_cycles += (im["issue_cycles"] if im else 4.0 * k["SQ_INSTS_VALU"]) * k["calls"]
        short = name.replace("mlmcpi::", "")
        kind = None
        if "or_heat_kernel" in short or "perm_heat_kernel" in short:
            kind = "or_heat"   # K overrelaxation sweeps + the heat-bath sweep in one launch: fuse = K + 1 sweeps
        elif "or_patch_kernel" in short or "or_block_kernel" in short or "or_kernel" in short or "sweep_kernel<false" in short or "schwinger_perm_kernel" in short:
            kind = "overrelax"
        elif "sweep_kernel<true" in short:
            kind = "heatbath"
        fuse = 1
        if "or_patch_kernel<" in short or "or_block_kernel<" in short:
            fuse = int(short.split("<")[1].split(">")[0].split(",")[0])   # <K> or <K, tile>
        if "or_heat_kernel<" in short:
            fuse = int(short.split("<")[1].split(">")[0].split(",")[0]) + 1   # <K> or <K, wide>
        if "perm_heat_kernel" in short:
            fuse = 11   # K is a launch argument: the profiled run is the reference's draw, 10 + 1 sweeps in one launch
        # bench.py's hbm_bound_probes (schwinger, --probes): single launches of the HBM-bound kernels of the path
        if w == "schwinger" and "hbm_bytes_per_launch" in k and any(
                t in short for t in ("schwinger_or_block_kernel<1>", "schwinger_perm_kernel", "schwinger_reduce_band_kernel", "schwinger_force_kernel")):
            out["probes"].append({"workload": w, "size": SIZES[w], "chains": CHAINS[w], "kernel": short,
                                  "hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "read_bytes": k["read_bytes_corrected"],
                                  "write_bytes": k["write_bytes"], "launches": k.get("FETCH_SIZE_launches"), "build": build,
                                  "source": f"profiles/{tag}_profile_summary.json"})
        if kind and w in ("schwinger", "gff", "rotor_sweep"):
            if "hbm_bytes_per_launch" in k:
                out["entries"].append({"workload": w, "size": SIZES[w], "chains": CHAINS[w], "fuse": fuse, "kind": kind, "kernel": short,
                                       "hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "read_bytes": k["read_bytes_corrected"],
                                       "write_bytes": k["write_bytes"], "launches": k.get("FETCH_SIZE_launches"), "build": build,
                                       "source": f"profiles/{tag}_profile_summary.json"})
            if "SQ_INSTS_VALU" in k:
                out["valu"].append({"workload": w, "size": SIZES[w], "chains": CHAINS[w], "fuse": fuse, "kind": kind, "kernel": short,
                                    "SQ_INSTS_VALU_per_launch": k["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": k.get("SQ_ACTIVE_INST_VALU"),
                                    "SQ_BUSY_CYCLES": k.get("SQ_BUSY_CYCLES"), "SQ_INSTS_SALU": k.get("SQ_INSTS_SALU"),
                                    "SQ_INSTS_LDS": k.get("SQ_INSTS_LDS"), "GRBM_GUI_ACTIVE": k.get("GRBM_GUI_ACTIVE"),
                                    "avg_ns_profiled": k.get("avg_ns"), "build": build, "source": f"profiles/{tag}_profile_summary.json",
                                    "issue": im})
    if tot_ns:
        # time-weighted vector-issue utilisation of the library's kernels over the profiled run:
        # wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time)
        out["kernels_valu_busy"].append({"workload": w, "size": SIZES[w], "chains": CHAINS[w], "wave_insts": tot_insts, "kernel_ns": tot_ns,
                                         "valu_frac": tot_insts / (tot_ns * 1e-9) / (256 * 4 * 2.4e9 / 4),
                                         # cost-weighted: sum of class counts x measured issue cycles / (1024 SIMDs x 2.4 GHz x time)
                                         "issue_frac": tot_cycles / (tot_ns * 1e-9) / (256 * 4 * 2.4e9), "build": build,
                                         "dominant_kernel": max(timed, key=timed.get).replace("mlmcpi::", ""),
                                         "kernels": "the timed region's: " + ", ".join(TIMED[w]) if w in TIMED else "every library kernel of the run",
                                         "all_kernels_valu_frac": all_insts / (all_ns * 1e-9) / (256 * 4 * 2.4e9 / 4),
                                         "source": f"profiles/{tag}_profile_summary.json"})
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
json.dump(S, open(os.path.join(ROOT, "profiles", f"{tag}_profile_summary.json"), "w"), indent=1)
for f in os.listdir(src):
    if f.endswith("_kernel_stats.csv"):
        open(os.path.join(ROOT, "profiles", f"{tag}_{f}"), "w").write(open(os.path.join(src, f)).read())
print(len(out["entries"]), "traffic entries,", len(out["valu"]), "valu entries,", len(out["kernels_valu_busy"]), "workload utilisations")
for e in out["kernels_valu_busy"]:
    print(f"  {e['workload']:18s} valu_frac {e['valu_frac']:.3f} issue_frac {e['issue_frac']:.3f}  ({e['dominant_kernel']})")
