"""CPU oracle for the LIBSPMV hot path -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (spmv_amd) must never import it.
"""
from ._c import (build, cg, cpu_baseline, csr_spmv, csr_spmv_sym, ddot, gather_ghosts,  # noqa: F401
                 max_threads, omp_row_split, omp_spmv, poisson3d,
                 poisson3d_lower, time_cg,
                 time_spmv)
from .host_logic import *  # noqa: F401,F403
from . import host_logic, petsc_io  # noqa: F401
