"""spmv_amd -- MI355X (gfx950) native backend for the LIBSPMV hot path:
fp64 CSR / symmetric-CSR SpMV, the CG loop and the L2GMap halo exchange.

Layers:
  csrc/hip/   hand-written HIP kernels + C ABI   -> lib/libspmv_hip.so
  csrc/host/  C++17 mirror of the reference's DeviceExecutor / CSRMatrix /
              L2GMap / Matrix / cg interface      -> lib/libspmv_host.so
  hip.py, host.py  ctypes handles used by tests/ and bench.py

Importing this package loads the native libraries and raises if they are
missing: there is no CPU or torch fallback.
"""
from . import _lib  # noqa: F401  (loads libspmv_hip.so, fails loudly)
from . import hip, poisson  # noqa: F401
