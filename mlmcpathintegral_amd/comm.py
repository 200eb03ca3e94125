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

    def close(self):
        if self.h:
            load().mlmcpi_comm_destroy(self.h)
            self.h = C.c_void_p()
