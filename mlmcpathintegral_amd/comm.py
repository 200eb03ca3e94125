"""ctypes binding of include/mlmcpi_comm.h (libmlmcpi_rccl.so): the packed statistics all-reduce on RCCL over xGMI.

Used by bench.py at N > 1 and by tests.  In a PyTorch process the RCCL runtime torch already carries is handed to the
library (one copy of RCCL per process); the rendezvous id travels by whatever channel the caller has -- bench.py
broadcasts it with torch.distributed, C++ hosts use a file (mlmcpi_comm_init_file)."""
import ctypes as C
import glob
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ID_BYTES = 128
_lib = None

SIGNATURES = {
    "mlmcpi_comm_last_error": (C.c_char_p, []),
    "mlmcpi_comm_load": (C.c_int, [C.c_char_p]),
    "mlmcpi_comm_runtime": (C.c_char_p, []),
    "mlmcpi_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mlmcpi_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mlmcpi_comm_init_file": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_void_p]),
    "mlmcpi_comm_rank": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mlmcpi_comm_size": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mlmcpi_comm_allreduce_sum_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mlmcpi_comm_allreduce_sum_host_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mlmcpi_comm_destroy": (C.c_int, [C.c_void_p]),
}


class CommError(RuntimeError):
    pass


def load():
    """dlopen libmlmcpi_rccl.so (no CPU fallback: raises when it is missing) and declare the signatures."""
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libmlmcpi_rccl.so")
        if not os.path.exists(path):
            raise CommError(f"{path} is missing: run `python __graft_entry__.py build`")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise CommError(f"{what} failed ({rc}): {load().mlmcpi_comm_last_error().decode()}")


def torch_rccl_path():
    """the RCCL runtime a PyTorch-ROCm process already has mapped (torch/lib/librccl.so*), if any"""
    try:
        import torch
    except ImportError:
        return None
    hits = sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")))
    return hits[0] if hits else None


def mapped_rccl_files():
    """every librccl file this process has mapped (/proc/self/maps): torch.distributed's RCCL and the one the library
    opened are the same file when the list has one entry"""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split(None, 5)[-1].strip() if len(line.split(None, 5)) == 6 else ""
                if "librccl" in os.path.basename(path) and os.path.realpath(path) not in seen:
                    seen.append(os.path.realpath(path))
    except OSError:
        pass
    return seen


def open_runtime(path=None):
    p = path or os.environ.get("MLMCPI_RCCL_LIB") or torch_rccl_path()
    _check(load().mlmcpi_comm_load(p.encode() if p else None), "mlmcpi_comm_load")


def unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    _check(load().mlmcpi_comm_unique_id(buf), "mlmcpi_comm_unique_id")
    return buf.raw


class Comm:
    def __init__(self, rank, world, id128, device):
        self.h = C.c_void_p()
        _check(load().mlmcpi_comm_init(rank, world, id128, device, C.byref(self.h)), "mlmcpi_comm_init")
        self.rank, self.world = rank, world

    def allreduce_sum_(self, t, stream=None):
        """in place on a contiguous float64 CUDA tensor, enqueued on `stream` (default: torch's current stream)"""
        import torch
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _check(load().mlmcpi_comm_allreduce_sum_f64(self.h, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(s)),
               "mlmcpi_comm_allreduce_sum_f64")
        return t

    def allreduce_sum_host(self, values):
        arr = (C.c_double * len(values))(*values)
        _check(load().mlmcpi_comm_allreduce_sum_host_f64(self.h, arr, len(values)), "mlmcpi_comm_allreduce_sum_host_f64")
        return list(arr)

    def size(self):
        """number of ranks as the communicator itself reports it (ncclCommCount)"""
        n = C.c_int(0)
        _check(load().mlmcpi_comm_size(self.h, C.byref(n)), "mlmcpi_comm_size")
        return n.value

    def comm_rank(self):
        r = C.c_int(-1)
        _check(load().mlmcpi_comm_rank(self.h, C.byref(r)), "mlmcpi_comm_rank")
        return r.value

    def close(self):
        if self.h:
            load().mlmcpi_comm_destroy(self.h)
            self.h = C.c_void_p()


def library_path():
    return os.path.join(_HERE, "libmlmcpi_rccl.so")


def runtime_path():
    """file the RCCL runtime in use was mapped from ('' before open_runtime)"""
    return load().mlmcpi_comm_runtime().decode()


def make_rccl_exchange(rank, world, device, dist, torch):
    """The library communicator of this rank: rendezvous id broadcast with torch.distributed, ncclCommInitRank through
    mlmcpi_comm_init (the RCCL runtime was opened by open_runtime(), the local step).  Collective; raises CommError."""
    idt = torch.zeros(ID_BYTES, dtype=torch.uint8, device="cuda")
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, src=0)
    return Comm(rank, world, bytes(idt.cpu().tolist()), device)


def establish(rank, world, dist, torch, make_exchange, prepare=None, agree_device="cpu", watchdog_s=180.0):
    """Set up the statistics exchange of an N-rank run on the MAIN thread, prove what it is, or end the run.

    prepare()        this rank's local, non-collective steps (dlopen of the RCCL runtime); may raise
    make_exchange()  collective construction; returns an object with allreduce_sum_host(list) -> list and size()
                     (Comm, or a stand-in in tests); may raise
    After each of the two, and after the check below, the ranks agree through a torch.distributed all-reduce(MIN) on
    whether EVERY rank succeeded; if not, every rank prints the error it has and exits with status 3 -- no fallback,
    no retry, no rank left waiting in a collective its peer never enters.
    The check: one all-reduce of (1, rank) through the exchange must return (N, N (N - 1) / 2) on every rank and the
    exchange must report N ranks.
    A rank that hangs inside a collective is ended by the watchdog (faulthandler: traceback of every thread to stderr,
    then _exit(1)), so a wedged communicator is a red run with a trace, never a green one.
    Returns (exchange, record) with record = {"ranks", "allreduce_check", "expected"}."""
    import faulthandler
    import sys

    def agree(stage, err):
        ok = torch.tensor([0.0 if err else 1.0], device=agree_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) != 1.0:
            print(f"[rank {rank}] statistics exchange, {stage}: {err or 'failed on another rank'}; exiting with status 3",
                  file=sys.stderr, flush=True)
            # Status 3 on EVERY rank, whatever the process group's worker threads do while the interpreter is taken down: a
            # normal exit from here lets gloo / RCCL threads race the teardown, and now and then one of them aborts the
            # process ("terminate called without an active exception", status 134) -- still red, but not the documented code.
            sys.stdout.flush()
            os._exit(3)

    def attempt(f):
        try:
            return f(), None
        except Exception as e:  # noqa: BLE001 -- reported and fatal in agree()
            return None, repr(e)[:300]

    faulthandler.dump_traceback_later(watchdog_s, exit=True, file=sys.stderr)
    try:
        if prepare is not None:
            agree("local set-up", attempt(prepare)[1])
        ex, err = attempt(make_exchange)
        agree("communicator set-up", err)
        expected = 0.5 * world * (world - 1)
        got, err = attempt(lambda: (ex.allreduce_sum_host([1.0, float(rank)]), ex.size()))
        if err is None:
            (ones, ranks), n = got
            if not (n == world and ones == float(world) and ranks == expected):
                err = (f"asked for {world} ranks, the exchange reports {n}, sum of ones {ones}, sum of ranks {ranks} "
                       f"(expected {expected})")
        agree("all-reduce check", err)
        return ex, {"ranks": n, "allreduce_check": ranks, "expected": expected}
    finally:
        faulthandler.cancel_dump_traceback_later()
