"""The C++ host layer (include/mlmcpi/*.hh): mirror of the reference's SampleState / Lattice /
Action / Sampler / QoI / Statistics / MonteCarloSingleLevel classes over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "test_host")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mlmcpathintegral_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])


def run(*args, timeout=600):
    if not os.path.exists(EXE):
        build()
    return subprocess.run([EXE, *args], capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("case,message", [
    ("coarsen_odd", "cannot coarsen 1d lattice with M = 7 points."),       # lattice/lattice1d.hh:81-86
    ("gff_not_square", "Lattice has to be squared for GFF action"),        # action/qft/gffaction.hh:169-173
    ("heatbath_on_quartic", "heat bath update not implemented for this action"),  # action/action.hh:73-79
])
def test_error_convention_without_gpu(case, message):
    """Reference convention: "ERROR: ..." on stderr and exit(EXIT_FAILURE) (mpi/mpi_wrapper.cc:174-177)."""
    r = run("--fatal", case)
    assert r.returncode == 1
    assert r.stderr.startswith("ERROR: ") and message in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("case,message", [
    ("per_site_update", "heat bath update not implemented"),
    ("qoi_wrong_size", "Evaluating QoISusceptibility on path of wrong size."),    # qoi/qm/qoixsquared.cc:9-11 (sic)
])
def test_error_convention_on_gpu(case, message):
    r = run("--fatal", case)
    assert r.returncode == 1 and message in r.stderr


@pytest.mark.gpu
def test_host_layer_on_gpu():
    """Known answers through the C++ classes, lazy host/device mirroring, BASELINE config 1 (HO,
    M_lat=128, HMC with auto-tuning) and a Schwinger heat-bath run through MonteCarloSingleLevel."""
    r = run(timeout=900)
    print(r.stdout[-2500:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_driver_multilevel_on_gpu():
    """host/driver (the counterpart of driver_qm / driver_qft): multilevel method, hierarchical sampler, harmonic
    oscillator -- the estimate must agree with the closed form the driver prints (driver_qm.cc:411-425)."""
    import re
    if not os.path.exists(EXE):
        build()
    r = subprocess.run([os.path.join(ROOT, "host", "driver"), "--method", "multilevel", "--action", "harmonicoscillator",
                        "--M_lat", "64", "--T_final", "4", "--sampler", "hierarchical", "--n_level", "3", "--epsilon", "0.03",
                        "--nt", "20", "--dt", "0.15", "--n_meas", "50"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"\|analytic - numerical\| / error = ([0-9.eE+-]+)", r.stdout)
    assert m and float(m.group(1)) < 5.0


@pytest.mark.gpu
def test_driver_throughput_mode_on_gpu():
    """host/driver --method throughput: the C++ sampling loop (Sampler::draw without a copy, QoI and moments on the
    device) on a batch of Schwinger chains; prints one JSON line with link-updates/s and a sane plaquette."""
    import json
    if not os.path.exists(EXE):
        build()
    r = subprocess.run([os.path.join(ROOT, "host", "driver"), "--method", "throughput", "--action", "schwinger", "--Mt_lat", "256",
                        "--sampler", "heatbath", "--batch", "8", "--n_samples", "10", "--n_burnin", "20"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(line)
    assert line["batch"] == 8 and line["updates_per_s"] > 1e9 and 0.40 < line["qoi_mean"] < 0.49


@pytest.mark.gpu
def test_driver_throughput_at_the_headline_shape_matches_the_python_driven_loop():
    """the C++ path's rate is a record, not a claim.  host/driver at 1024^2 x 32 chains against the
    same loop driven through ctypes in this session (one ABI call for draw + QoI, one for record_sample): within 7 %
    (the two run minutes apart on a chip whose clock moves by a few per cent), and both see a thermalised plaquette."""
    import json
    import time
    import torch
    from mlmcpathintegral_amd import abi, ops
    if not os.path.exists(EXE):
        build()
    r = subprocess.run([os.path.join(ROOT, "host", "driver"), "--method", "throughput", "--action", "schwinger", "--Mt_lat", "1024",
                        "--sampler", "heatbath", "--batch", "32", "--n_samples", "40", "--n_burnin", "30"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
    x = ops.lattice_initialise(act, 32, 2481317, 0)
    w = torch.empty_like(x)
    acc = torch.zeros((32, 5), dtype=torch.float64, device="cuda")
    sweep = 0
    def step():
        nonlocal x, w, sweep
        x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, 10, 1, 2481317, 0, sweep, 1, 0)
        ops.stats_accumulate(acc, q)
        sweep += 11
    for _ in range(35):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 40
    print("C++ driver %.4f ms per sample, ctypes-driven %.4f ms" % (line["ms_per_sample"], ms))
    assert abs(line["ms_per_sample"] / ms - 1.0) < 0.07, (line, ms)
    assert 0.44 < line["qoi_mean"] < 0.452


@pytest.mark.gpu
def test_comm_single_rank_on_gpu():
    """libmlmcpi_rccl.so through ctypes with the RCCL runtime of this PyTorch process: communicator of one rank, device
    and host all-reduce (the N > 1 path of bench.py uses the same calls; one GPU per rank is all RCCL allows)."""
    import torch
    from mlmcpathintegral_amd import comm
    comm.open_runtime()
    c = comm.Comm(0, 1, comm.unique_id(), 0)
    t = torch.arange(7, dtype=torch.float64, device="cuda") * 0.5
    c.allreduce_sum_(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(7, dtype=torch.float64) * 0.5)
    assert c.allreduce_sum_host([1.0, 2.5]) == [1.0, 2.5]
    c.close()


@pytest.mark.gpu
def test_driver_two_ranks_over_rccl_when_two_gpus_are_visible():
    """the N > 1 RCCL path of the C++ host (RcclExchange over the rendezvous file, RcclExchange::verify)
    must RUN wherever two GPUs are visible: two driver processes, one per GPU, throughput mode; the line rank 0 prints
    names the rank count the communicator reports.  (One-GPU boxes: nothing to run -- RCCL allows one rank per GPU.)"""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL allows one rank per GPU")
    if not os.path.exists(EXE):
        build()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [os.path.join(ROOT, "host", "driver"), "--method", "throughput", "--action", "schwinger", "--Mt_lat", "256",
           "--sampler", "heatbath", "--batch", "4", "--n_samples", "10", "--n_burnin", "10"]
    procs = [subprocess.Popen(cmd, env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_PORT=str(port),
                                            HSA_ENABLE_IPC_MODE_LEGACY="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert line["ranks"] == 2 and 0.40 < line["qoi_mean"] < 0.49


def test_rccl_exchange_failure_is_fatal_for_the_driver(tmp_path):
    """A rank that cannot join (here: the only file at the rendezvous path was left by a dead run, and no rank 0 comes)
    ends with "ERROR: ..." and EXIT_FAILURE -- the reference's convention (mpi/mpi_wrapper.cc:174-177) -- instead of
    waiting in ncclCommInitRank.  CPU only: the failure comes before anything touches RCCL or a GPU."""
    import struct
    if not os.path.exists(os.path.join(ROOT, "host", "stats_check")):
        build()
    stale = tmp_path / "id"
    stale.write_bytes(b"MLMCPI1\0" + struct.pack("<qQ", 1 << 22, 1) + b"\x00" * 128)
    r = subprocess.run([os.path.join(ROOT, "host", "stats_check"), "--rccl-join", str(stale), "1", "2", "0.5"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "ERROR: mlmcpi_comm_init_file" in r.stderr and "stale" in r.stderr, r.stderr


def test_driver_rejects_unknown_options():
    r = subprocess.run([os.path.join(ROOT, "host", "driver"), "--no_such_option", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "unknown option" in r.stderr
