"""Probe: communicator of one rank on the system RCCL (no torch in the process).  python tools/rccl_probe.py [lib]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmcpathintegral_amd import comm
lib = sys.argv[1] if len(sys.argv) > 1 else "/opt/rocm/lib/librccl.so.1"
comm.open_runtime(lib)
c = comm.Comm(0, 1, comm.unique_id(), 0)
print("init ok", c.allreduce_sum_host([1.0, 2.0]))
c.close()
