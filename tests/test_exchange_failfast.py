"""CPU, world_size 8 over gloo: the set-up of the N-rank statistics exchange (mlmcpathintegral_amd.comm.establish, the
procedure bench.py runs on the main thread before its timed region) ends the run on EVERY rank with a non-zero status
when any rank cannot build its exchange or when the check all-reduce does not return what N ranks must produce -- no
fallback, no rank left behind in a collective (replaces the reference's unchecked
mpi/mpi_wrapper.cc:44-120 calls)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 8


class _Exchange:
    """gloo stand-in with the two calls establish() uses; `corrupt_rank` adds 1 to that rank's contribution"""

    def __init__(self, torch, dist, rank, corrupt_rank=-1, claim=None):
        self.torch, self.dist, self.rank, self.corrupt_rank, self.claim = torch, dist, rank, corrupt_rank, claim

    def allreduce_sum_host(self, values):
        v = list(values)
        if self.rank == self.corrupt_rank:
            v[-1] += 1.0
        t = self.torch.tensor(v, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.tolist()

    def size(self):
        return self.claim if self.claim is not None else self.dist.get_world_size()


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mlmcpathintegral_amd import comm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def prepare():
        if mode == "prepare_fails" and rank == 0:
            raise comm.CommError("cannot open the RCCL runtime (injected)")

    def make():
        if mode == "construction_fails" and rank == 3:
            raise comm.CommError("ncclCommInitRank(rank 3 of 8): unhandled system error (injected)")
        return _Exchange(torch, dist, rank, corrupt_rank=5 if mode == "wrong_sum" else -1,
                         claim=7 if mode == "wrong_count" and rank == 6 else None)

    ex, rec = comm.establish(rank, world, dist, torch, make, prepare=prepare, watchdog_s=120.0)
    # only a proven exchange gets here
    total = ex.allreduce_sum_host([float(rank + 1)])[0]
    q.put((rank, rec, total))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(mode):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, mode, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=170)
    codes = [p.exitcode for p in procs]
    got = []
    while not q.empty():
        got.append(q.get())
    return codes, sorted(got)


@pytest.mark.parametrize("mode", ["prepare_fails", "construction_fails", "wrong_sum", "wrong_count"])
def test_a_failing_exchange_ends_every_rank_with_status_3(mode):
    codes, got = _run(mode)
    assert codes == [3] * WORLD, f"{mode}: exit codes {codes}"
    assert got == [], "no rank may go on with an exchange that was not proven"


def test_a_healthy_exchange_reports_its_ranks():
    codes, got = _run("healthy")
    assert codes == [0] * WORLD
    assert len(got) == WORLD
    for rank, rec, total in got:
        assert rec == {"ranks": WORLD, "allreduce_check": WORLD * (WORLD - 1) / 2, "expected": WORLD * (WORLD - 1) / 2}
        assert total == WORLD * (WORLD + 1) / 2


def test_bench_has_no_fallback_or_silent_exit():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for banned in ("os._exit", "daemon=True", "threading", "fallback;"):
        assert banned not in src, banned
    assert "comm.establish(" in src and '"rccl": rccl' in src


def test_mapped_rccl_files_names_torchs_runtime():
    """bench.py's rccl record says whether torch.distributed and the library communicator run on ONE librccl file
    (comm.mapped_rccl_files, /proc/self/maps): a process that has imported torch.distributed maps torch's own librccl,
    which is also the file comm.open_runtime() hands to the library by default."""
    import os
    import torch.distributed  # noqa: F401  (maps torch's librccl)
    from mlmcpathintegral_amd import comm
    mapped = comm.mapped_rccl_files()
    want = comm.torch_rccl_path()
    assert want is not None and os.path.realpath(want) in mapped, (want, mapped)
    assert all("librccl" in os.path.basename(p) for p in mapped)
