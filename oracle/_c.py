"""ctypes binding of oracle/libspmv_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; the product (spmv_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libspmv_oracle.so")


def build(force=False):
    """Compile the C restatement (gcc, seconds)."""
    src = os.path.join(_HERE, "spmv_oracle.c")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libspmv_oracle.so"])
    return _SO


_lib = None


class BaselineResult(C.Structure):
    _fields_ = [("spmv_s_per_apply", C.c_double), ("cg_loop_s", C.c_double),
                ("setup_s", C.c_double), ("rel_residual", C.c_double),
                ("rel_residual_k10", C.c_double),
                ("cg_iters", C.c_int), ("threads", C.c_int)]

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    L.oracle_csr_spmv.argtypes = [C.c_int32, _i32p, _i32p, _f64p, C.c_double,
                                  _f64p, C.c_double, _f64p]
    L.oracle_csr_spmv.restype = None
    L.oracle_csr_spmv_sym.argtypes = [C.c_int32, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, _f64p,
                                      C.c_double, _f64p, C.c_double, _f64p]
    L.oracle_csr_spmv_sym.restype = None
    L.oracle_csr_spmv_f32.argtypes = [C.c_int32, _i32p, _i32p, _f32p,
                                      C.c_float, _f32p, C.c_float, _f32p]
    L.oracle_csr_spmv_f32.restype = None
    L.oracle_csr_spmv_sym_f32.argtypes = [C.c_int32, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, _f32p,
                                          C.c_float, _f32p, C.c_float, _f32p]
    L.oracle_csr_spmv_sym_f32.restype = None
    L.oracle_gather_ghosts.argtypes = [C.c_int, _i32p, _f64p, _f64p]
    L.oracle_gather_ghosts.restype = None
    L.oracle_omp_row_split.argtypes = [C.c_int32, C.c_int64, _i32p, C.c_int,
                                       _i32p]
    L.oracle_omp_row_split.restype = None
    L.oracle_omp_init.argtypes = [C.c_int32, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int]
    L.oracle_omp_init.restype = C.c_void_p
    L.oracle_omp_free.argtypes = [C.c_void_p]
    L.oracle_omp_free.restype = None
    L.oracle_omp_spmv.argtypes = [C.c_void_p, C.c_int32, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_double, _f64p, C.c_double,
                                  _f64p]
    L.oracle_omp_spmv.restype = None
    L.oracle_ddot.argtypes = [C.c_int64, _f64p, _f64p]
    L.oracle_ddot.restype = C.c_double
    L.oracle_cg.argtypes = [C.c_int32, C.c_int64, _i32p, _i32p, _f64p,
                            C.c_void_p, _f64p, _f64p, C.c_int, C.c_double,
                            C.c_void_p, C.c_int]
    L.oracle_cg.restype = C.c_int
    L.oracle_time_spmv.argtypes = [C.c_int32, C.c_int64, _i32p, _i32p, _f64p,
                                   C.c_void_p, _f64p, _f64p, C.c_int, C.c_int]
    L.oracle_time_spmv.restype = C.c_double
    L.oracle_poisson3d.argtypes = [C.c_int32, _i32p, _i32p, _f64p]
    L.oracle_poisson3d.restype = None
    L.oracle_poisson3d_lower.argtypes = [C.c_int32, _i32p, _i32p, _f64p, _f64p]
    L.oracle_poisson3d_lower.restype = None
    L.oracle_time_cg.argtypes = [C.c_int32, C.c_int64, _i32p, _i32p, _f64p,
                                 _f64p, _f64p, C.c_int, C.c_int,
                                 C.POINTER(C.c_int)]
    L.oracle_time_cg.restype = C.c_double
    L.oracle_cpu_baseline.argtypes = [C.c_int32, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(BaselineResult)]
    L.oracle_cpu_baseline.restype = C.c_int
    L.oracle_max_threads.argtypes = []
    L.oracle_max_threads.restype = C.c_int
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def csr_spmv(rowptr, colind, values, x, alpha=1.0, beta=0.0, y=None):
    """General CSR SpMV, reference order (csr_kernels.cpp:41-51)."""
    n = len(rowptr) - 1
    f32 = np.asarray(values).dtype == np.float32
    dt = np.float32 if f32 else np.float64
    out = np.zeros(n, dt) if y is None else _c(y, dt).copy()
    fn = lib().oracle_csr_spmv_f32 if f32 else lib().oracle_csr_spmv
    fn(n, _c(rowptr, np.int32), _c(colind, np.int32), _c(values, dt),
       alpha, _c(x, dt), beta, out)
    return out


def csr_spmv_sym(rowptr, colind, values, diagonal, x, alpha=1.0, beta=0.0,
                 y=None):
    """Symmetric (strictly-lower + diagonal) SpMV (csr_kernels.cpp:26-40)."""
    n = len(diagonal)
    f32 = np.asarray(diagonal).dtype == np.float32
    dt = np.float32 if f32 else np.float64
    out = np.zeros(n, dt) if y is None else _c(y, dt).copy()
    nnz = 0 if values is None else len(values)
    rp = None if nnz == 0 else _c(rowptr, np.int32)
    ci = None if nnz == 0 else _c(colind, np.int32)
    va = None if nnz == 0 else _c(values, dt)
    fn = lib().oracle_csr_spmv_sym_f32 if f32 else lib().oracle_csr_spmv_sym
    fn(n, nnz, _ptr(rp), _ptr(ci), _ptr(va), _c(diagonal, dt), alpha,
       _c(x, dt), beta, out)
    return out


def gather_ghosts(indices, x):
    idx = _c(indices, np.int32)
    out = np.empty(len(idx), np.float64)
    lib().oracle_gather_ghosts(len(idx), idx, _c(x, np.float64), out)
    return out


def omp_row_split(rowptr, num_threads):
    rp = _c(rowptr, np.int32)
    n = len(rp) - 1
    out = np.zeros(num_threads + 1, np.int32)
    lib().oracle_omp_row_split(n, int(rp[-1]), rp, num_threads, out)
    return out


def omp_spmv(rowptr, colind, values, x, diagonal=None, alpha=1.0, beta=0.0,
             y=None, num_threads=2):
    """OpenMP restatement (csr_kernels.openmp.cpp:172-244)."""
    sym = diagonal is not None
    n = len(diagonal) if sym else len(rowptr) - 1
    nnz = 0 if values is None else len(values)
    rp = None if nnz == 0 else _c(rowptr, np.int32)
    ci = None if nnz == 0 else _c(colind, np.int32)
    va = None if nnz == 0 else _c(values, np.float64)
    dg = None if not sym else _c(diagonal, np.float64)
    out = np.zeros(n) if y is None else _c(y, np.float64).copy()
    plan = lib().oracle_omp_init(n, nnz, _ptr(rp), _ptr(ci), int(sym),
                                 num_threads)
    try:
        lib().oracle_omp_spmv(plan, n, nnz, _ptr(rp), _ptr(ci), _ptr(va),
                              _ptr(dg), alpha, _c(x, np.float64), beta, out)
    finally:
        lib().oracle_omp_free(plan)
    return out


def ddot(x, y):
    return lib().oracle_ddot(len(x), _c(x, np.float64), _c(y, np.float64))


def cg(rowptr, colind, values, b, kmax, rtol, diagonal=None, num_threads=1):
    """Single-rank CG (cg.cpp:21-98). Returns (x, k, rnorm_history)."""
    n = len(b)
    rp = _c(rowptr, np.int32)
    ci = _c(colind, np.int32)
    va = _c(values, np.float64)
    dg = None if diagonal is None else _c(diagonal, np.float64)
    x = np.zeros(n)
    hist = np.zeros(kmax + 1)
    k = lib().oracle_cg(n, len(va), rp, ci, va, _ptr(dg), _c(b, np.float64),
                        x, kmax, rtol, _ptr(hist), num_threads)
    return x, k, hist[:k + 1]


def time_spmv(rowptr, colind, values, x, diagonal=None, reps=10,
              num_threads=1):
    """Seconds per apply of the (OpenMP) CPU path -- cpu_baseline leg."""
    rp = _c(rowptr, np.int32)
    ci = _c(colind, np.int32)
    va = _c(values, np.float64)
    dg = None if diagonal is None else _c(diagonal, np.float64)
    n = len(rp) - 1
    out = np.zeros(n)
    return lib().oracle_time_spmv(n, len(va), rp, ci, va, _ptr(dg),
                                  _c(x, np.float64), out, reps, num_threads)


def poisson3d(n):
    """C generator of the benchmark matrix (fast path for the cpu_baseline)."""
    N, nnz = n ** 3, 7 * n ** 3 - 6 * n ** 2
    rp = np.zeros(N + 1, np.int32)
    ci = np.zeros(nnz, np.int32)
    va = np.zeros(nnz, np.float64)
    lib().oracle_poisson3d(n, rp, ci, va)
    return rp, ci, va


def poisson3d_lower(n):
    """Strictly-lower CSR + diagonal of the same matrix (symmetric storage)."""
    N, nnz = n ** 3, 3 * n ** 3 - 3 * n ** 2
    rp = np.zeros(N + 1, np.int32)
    ci = np.zeros(nnz, np.int32)
    va = np.zeros(nnz, np.float64)
    dg = np.zeros(N, np.float64)
    lib().oracle_poisson3d_lower(n, rp, ci, va, dg)
    return rp, ci, va, dg


def time_cg(rowptr, colind, values, b, kmax, num_threads):
    """(seconds, iterations) of one fixed-length oracle CG solve."""
    n = len(b)
    x = np.zeros(n)
    k = C.c_int()
    t = lib().oracle_time_cg(n, len(values), rowptr, colind, values,
                             _c(b, np.float64), x, kmax, num_threads,
                             C.byref(k))
    return t, k.value


def cpu_baseline(n, threads, spmv_reps, cg_iters):
    """OpenMP SpMV + CG on the n^3 Poisson matrix, owner first touch, loops
    timed alone.  Returns a dict."""
    res = BaselineResult()
    rc = lib().oracle_cpu_baseline(n, threads, spmv_reps, cg_iters, C.byref(res))
    if rc != 0:
        raise RuntimeError(f"oracle_cpu_baseline failed ({rc})")
    return {f: getattr(res, f) for f, _ in BaselineResult._fields_}


def max_threads():
    return lib().oracle_max_threads()
