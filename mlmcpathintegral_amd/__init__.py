"""mlmcpathintegral_amd -- MI355X-native inner MCMC sweep of eikehmueller/mlmcpathintegral.

Layout
  csrc/      hand-written HIP kernels for gfx950 and the C ABI of include/mlmcpi_hip.h
  abi.py     ctypes binding of that ABI (raises when libmlmcpi_hip.so is missing: no CPU fallback)
  ops.py     torch-tensor convenience layer over the ABI (tests, bench.py)
  chains.py  sharding of independent chains over ranks and the packed statistics all-reduce

The C++ host layer that mirrors the reference's Action / Sampler / QoI classes lives in
include/mlmcpi/ and binds the same ABI.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
