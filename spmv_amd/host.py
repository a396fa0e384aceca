"""ctypes handles over the C facade (include/spmv_host_c.h) of the C++17 host
mirror: spmv::HipExecutor, L2GMap, Matrix<double>, cg.

Harness plumbing for tests/ and bench.py.  Method names follow the C++
classes so the parity tests read like the reference's tests/test_spmv.cpp.
Nothing here computes.
"""
import ctypes as C

import numpy as np

from . import _lib

P2P_BLOCKING, P2P_NONBLOCKING, COLLECTIVE_BLOCKING, COLLECTIVE_NONBLOCKING = 0, 1, 2, 3
ONESIDED_PUT_ACTIVE, ONESIDED_PUT_PASSIVE, SHMEM, SHMEM_NODUP = 4, 5, 6, 7

lib = _lib._load("libspmv_host.so")
lib.spmvh_last_error.restype = C.c_char_p

vp, i32, i64, f64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
PTR = C.POINTER

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp, sz)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, vp, sz, C.c_int, PTR(C.c_int), vp,
                          PTR(i32), PTR(i32), vp, PTR(i32), PTR(i32), vp)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, sz, vp)

_PROTOS = {
    "spmvh_exec_create": [C.c_int, PTR(vp)],
    "spmvh_exec_destroy": [vp],
    "spmvh_exec_alloc": [vp, sz, PTR(vp)],
    "spmvh_exec_free": [vp, vp],
    "spmvh_exec_memset": [vp, vp, C.c_int, sz],
    "spmvh_exec_copy": [vp, vp, vp, sz],
    "spmvh_exec_copy_from_host": [vp, vp, vp, sz],
    "spmvh_exec_copy_to_host": [vp, vp, vp, sz],
    "spmvh_exec_synchronize": [vp],
    "spmvh_exec_num_cus": [vp, PTR(C.c_int)],
    "spmvh_exec_device_type": [vp, PTR(C.c_int)],
    "spmvh_exec_context": [vp, PTR(vp)],
    "spmvh_host_executor_rejects_compute": [],
    "spmvh_comm_self": [PTR(vp)],
    "spmvh_rccl_unique_id": [vp],
    "spmvh_comm_rccl": [vp, C.c_int, C.c_int, vp, PTR(vp)],
    "spmvh_comm_rccl_info": [vp, PTR(C.c_int), C.c_char_p, C.c_int],
    "spmvh_comm_callback": [C.c_int, C.c_int, ALLGATHER_FN, EXCHANGE_FN,
                            ALLREDUCE_FN, vp, PTR(vp)],
    "spmvh_comm_destroy": [vp],
    "spmvh_comm_enable_peer_reduce": [vp, vp, PTR(C.c_int)],
    "spmvh_comm_reduce_sum": [vp, vp, C.c_int, vp],
    "spmvh_comm_ranks_share_a_process": [vp, PTR(C.c_int)],
    "spmvh_matrix_create": [vp, vp, vp, vp, vp, i64, i64, vp, i64, vp, i64,
                            C.c_int, C.c_int, PTR(vp)],
    "spmvh_matrix_f32_create": [vp, vp, vp, vp, vp, i64, i64, vp, i64, vp, i64,
                                C.c_int, C.c_int, PTR(vp)],
    "spmvh_matrix_f32_destroy": [vp],
    "spmvh_matrix_f32_info": [vp, PTR(C.c_int), PTR(i64), PTR(i32), PTR(i32)],
    "spmvh_matrix_f32_update": [vp, vp],
    "spmvh_matrix_f32_mult": [vp, vp, vp],
    "spmvh_split_create_dist": [vp, vp, vp, vp, i64, i64, vp, i64, vp, i64,
                                C.c_int, C.c_int, PTR(vp), PTR(i64)],
    "spmvh_matrix_release_csr": [vp, PTR(i64)],
    "spmvh_matrix_create_poisson3d": [vp, vp, i32, C.c_int, C.c_int, PTR(vp)],
    "spmvh_matrix_create_unstructured": [vp, vp, i64, C.c_int, i64, C.c_int,
                                         C.c_uint64, PTR(vp)],
    "spmvh_matrix_create_fem_like": [vp, vp, vp, PTR(vp)],
    "spmvh_matrix_create_fem_like_sym": [vp, vp, vp, PTR(vp)],
    "spmvh_matrix_create_poisson3d_boxes": [vp, vp, i32, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, PTR(vp)],
    "spmvh_poisson3d_box_rows": [i32, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                 vp, vp, vp, vp],
    "spmvh_matrix_destroy": [vp],
    "spmvh_matrix_rows": [vp, PTR(C.c_int)],
    "spmvh_matrix_cols": [vp, PTR(C.c_int)],
    "spmvh_matrix_non_zeros": [vp, PTR(i64)],
    "spmvh_matrix_format_size": [vp, PTR(sz)],
    "spmvh_matrix_symmetric": [vp, PTR(C.c_int)],
    "spmvh_matrix_blocks": [vp, PTR(i64)],
    "spmvh_matrix_plan_get": [vp, C.c_int, C.c_char_p, PTR(C.c_int)],
    "spmvh_matrix_plan_set": [vp, C.c_int, C.c_char_p, C.c_int],
    "spmvh_matrix_enable_mixed": [vp, PTR(C.c_int)],
    "spmvh_matrix_use_mixed": [vp, C.c_int],
    "spmvh_matrix_update": [vp, vp],
    "spmvh_matrix_update_finalise": [vp, vp],
    "spmvh_matrix_mult": [vp, vp, vp],
    "spmvh_split_create": [vp, vp, vp, i64, i64, i64, i64, vp, i64, C.c_int,
                           C.c_int, PTR(vp), PTR(i64)],
    "spmvh_split_get": [vp, C.c_int, vp, vp, vp],
    "spmvh_split_extra": [vp, vp, vp],
    "spmvh_split_destroy": [vp],
    "spmvh_l2g_sizes": [vp, PTR(i32), PTR(i32), PTR(i64), PTR(i64),
                        PTR(C.c_int), PTR(C.c_int), PTR(C.c_int), PTR(C.c_int)],
    "spmvh_l2g_ghosts": [vp, vp],
    "spmvh_l2g_onesided": [vp, vp],
    "spmvh_l2g_plan": [vp, vp, vp, vp, vp, vp, vp],
    "spmvh_l2g_global_to_local": [vp, i64, PTR(i32)],
    "spmvh_l2g_create": [vp, vp, C.c_int, i64, vp, i64, C.c_int, PTR(vp)],
    "spmvh_l2g_destroy": [vp],
    "spmvh_l2g_map_sizes": [vp, PTR(C.c_int), PTR(C.c_int), PTR(C.c_int)],
    "spmvh_l2g_map_plan": [vp, vp, vp, vp, vp, vp, vp],
    "spmvh_l2g_map_update": [vp, vp],
    "spmvh_l2g_map_reverse_update": [vp, vp],
    "spmvh_l2g_map_reverse_update_f32": [vp, vp],
    "spmvh_cg": [vp, vp, vp, vp, vp, C.c_int, f64, PTR(C.c_int), vp],
    "spmvh_read_petsc_matrix": [vp, vp, C.c_char_p, C.c_int, C.c_int, PTR(vp)],
    "spmvh_read_petsc_vector": [vp, vp, C.c_char_p, PTR(vp), PTR(i64)],
    "spmvh_petsc_rows_read": [C.c_char_p, C.c_int, C.c_int, PTR(vp), PTR(i64)],
    "spmvh_petsc_rows_get": [vp, vp, vp, vp, vp],
    "spmvh_petsc_rows_destroy": [vp],
    "spmvh_cg_workspace_create": [vp, PTR(vp)],
    "spmvh_cg_workspace_destroy": [vp],
    "spmvh_cg_mixed": [vp, vp, vp, vp, vp, C.c_int, C.c_double, C.c_int,
                       PTR(C.c_int), vp, C.c_int, vp, C.c_int,
                       PTR(C.c_double)],
    "spmvh_cg_workspace_reserve_timing": [vp, C.c_int],
    "spmvh_cg_ex": [vp, vp, vp, vp, vp, C.c_int, f64, PTR(C.c_int), vp, vp,
                    C.c_int, PTR(f64), PTR(C.c_int)],
}
for _n, _a in _PROTOS.items():
    _f = getattr(lib, _n)
    _f.argtypes = _a
    _f.restype = C.c_int
HOST_SYMBOLS = tuple(_PROTOS) + ("spmvh_last_error",)


class FemParams(C.Structure):
    """spmv_hip_fem_params (include/spmv_hip.h)"""
    _fields_ = [("num_rows", C.c_int64), ("min_len", C.c_int32),
                ("max_len", C.c_int32), ("layer", C.c_int32),
                ("jitter", C.c_int32), ("tail_permille", C.c_int32),
                ("tail_min", C.c_int32), ("tail_max", C.c_int32),
                ("tail_stride", C.c_int32), ("seed", C.c_uint64)]


class SpmvHostError(RuntimeError):
    pass


def call(name, *args):
    if getattr(lib, name)(*args) != 0:
        raise SpmvHostError(f"{name}: {lib.spmvh_last_error().decode()}")


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipExecutor:
    """spmv::HipExecutor::create(device_id, HostExecutor::create())"""

    def __init__(self, device_id=0):
        h = vp()
        call("spmvh_exec_create", device_id, C.byref(h))
        self.h = h

    def close(self):
        if self.h:
            call("spmvh_exec_destroy", self.h)
            self.h = None

    def alloc(self, count, dtype=np.float64):
        p = vp()
        call("spmvh_exec_alloc", self.h, int(count) * np.dtype(dtype).itemsize,
             C.byref(p))
        return p.value or 0

    def free(self, ptr):
        call("spmvh_exec_free", self.h, ptr)

    def memset(self, ptr, value, nbytes):
        call("spmvh_exec_memset", self.h, ptr, value, nbytes)

    def copy(self, dst, src, nbytes):
        call("spmvh_exec_copy", self.h, dst, src, nbytes)

    def copy_from_host(self, dst, arr):
        arr = np.ascontiguousarray(arr)
        call("spmvh_exec_copy_from_host", self.h, dst, _np_ptr(arr), arr.nbytes)

    def copy_to_host(self, src, count, dtype=np.float64):
        out = np.empty(count, dtype)
        call("spmvh_exec_copy_to_host", self.h, _np_ptr(out), src, out.nbytes)
        return out

    def synchronize(self):
        call("spmvh_exec_synchronize", self.h)

    @property
    def num_cus(self):
        n = C.c_int()
        call("spmvh_exec_num_cus", self.h, C.byref(n))
        return n.value

    @property
    def device_type(self):
        t = C.c_int()
        call("spmvh_exec_device_type", self.h, C.byref(t))
        return t.value

    @property
    def context(self):
        """spmv_hip_ctx* behind the executor (timing events in bench.py)."""
        c = vp()
        call("spmvh_exec_context", self.h, C.byref(c))
        return c


class Comm:
    def __init__(self, h, keep=()):
        self.h, self._keep = h, keep

    @classmethod
    def self_comm(cls):
        h = vp()
        call("spmvh_comm_self", C.byref(h))
        return cls(h)

    @classmethod
    def rccl(cls, exec_, nranks, rank, unique_id):
        h = vp()
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        call("spmvh_comm_rccl", exec_.h, nranks, rank, buf, C.byref(h))
        return cls(h)

    def rccl_info(self):
        """{nranks, rank, version, separate_reduction_comm, lib_path} as RCCL
        itself reports them"""
        out = (C.c_int * 4)()
        path = C.create_string_buffer(512)
        call("spmvh_comm_rccl_info", self.h, out, path, 512)
        v = out[2]
        return dict(nranks=out[0], rank=out[1], version_code=v,
                    version=f"{v // 10000}.{v // 100 % 100}.{v % 100}",
                    separate_reduction_comm=bool(out[3]),
                    lib_path=path.value.decode())

    @classmethod
    def callback(cls, rank, nranks, allgather, exchange=None, allreduce=None):
        """Transport supplied as Python callables (see spmv_host_c.h)."""
        ag = ALLGATHER_FN(allgather)
        ex = EXCHANGE_FN(exchange) if exchange else EXCHANGE_FN()
        ar = ALLREDUCE_FN(allreduce) if allreduce else ALLREDUCE_FN()
        h = vp()
        call("spmvh_comm_callback", rank, nranks, ag, ex, ar, None, C.byref(h))
        return cls(h, keep=(ag, ex, ar))

    def enable_peer_reduce(self, exec_):
        """Comm::enable_peer_reduce (collective): cg() then reduces its scalars
        through peer windows, added in rank order; False = not available."""
        ok = C.c_int()
        call("spmvh_comm_enable_peer_reduce", self.h, exec_.h, C.byref(ok))
        return bool(ok.value)

    def reduce_sum(self, device_ptr, count=1, stream=None):
        call("spmvh_comm_reduce_sum", self.h, device_ptr, int(count), stream)

    def ranks_share_a_process(self):
        """Comm::ranks_share_a_process (collective on its first call)"""
        v = C.c_int()
        call("spmvh_comm_ranks_share_a_process", self.h, C.byref(v))
        return bool(v.value)

    def close(self):
        if self.h:
            call("spmvh_comm_destroy", self.h)
            self.h = None


def rccl_unique_id():
    buf = (C.c_ubyte * 128)()
    call("spmvh_rccl_unique_id", buf)
    return bytes(buf)


class L2GPlanView:
    """Plan arrays of an L2GMap (reference member names, no underscore)."""

    def __init__(self, nn, ni, getter):
        self.neighbours = np.zeros(nn, np.int32)
        self.send_count = np.zeros(max(nn, 1), np.int32)
        self.recv_count = np.zeros(max(nn, 1), np.int32)
        self.send_offset = np.zeros(max(nn, 1) + 1, np.int32)
        self.recv_offset = np.zeros(max(nn, 1) + 1, np.int32)
        self.indexbuf = np.zeros(ni, np.int32)
        getter(_np_ptr(self.neighbours), _np_ptr(self.send_count),
               _np_ptr(self.recv_count), _np_ptr(self.send_offset),
               _np_ptr(self.recv_offset), _np_ptr(self.indexbuf))
        self.send_count = self.send_count[:nn]
        self.recv_count = self.recv_count[:nn]
        self.send_offset = self.send_offset[:nn + 1]
        self.recv_offset = self.recv_offset[:nn + 1]


class L2GMap:
    """Stand-alone spmv::L2GMap (plan tests)."""

    def __init__(self, comm, local_size, ghosts, exec_=None,
                 cm=COLLECTIVE_BLOCKING):
        g = np.ascontiguousarray(ghosts, dtype=np.int64)
        h = vp()
        call("spmvh_l2g_create", comm.h, exec_.h if exec_ else None,
             0 if exec_ else 1, int(local_size), _np_ptr(g), len(g), cm,
             C.byref(h))
        self.h = h

    def plan(self):
        nn, ni, packs = C.c_int(), C.c_int(), C.c_int()
        call("spmvh_l2g_map_sizes", self.h, C.byref(nn), C.byref(ni),
             C.byref(packs))
        v = L2GPlanView(nn.value, ni.value,
                        lambda *a: call("spmvh_l2g_map_plan", self.h, *a))
        v.packs = bool(packs.value)
        return v

    def update(self, x_ptr):
        call("spmvh_l2g_map_update", self.h, x_ptr)

    def reverse_update(self, x_ptr, f32=False):
        call("spmvh_l2g_map_reverse_update_f32" if f32
             else "spmvh_l2g_map_reverse_update", self.h, x_ptr)

    def close(self):
        if self.h:
            call("spmvh_l2g_destroy", self.h)
            self.h = None


class ColMapView:
    """A.col_map() of a Matrix."""

    def __init__(self, A):
        self.A = A
        ls, ng, gs, go = i32(), i32(), i64(), i64()
        ov, nn, ni, pk = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        call("spmvh_l2g_sizes", A.h, C.byref(ls), C.byref(ng), C.byref(gs),
             C.byref(go), C.byref(ov), C.byref(nn), C.byref(ni), C.byref(pk))
        self._local_size, self._num_ghosts = ls.value, ng.value
        self._global_size, self._global_offset = gs.value, go.value
        self._overlapping, self._nn, self._ni = bool(ov.value), nn.value, ni.value
        self.packs = bool(pk.value)

    def local_size(self):
        return self._local_size

    def num_ghosts(self):
        return self._num_ghosts

    def global_size(self):
        return self._global_size

    def global_offset(self):
        return self._global_offset

    def overlapping(self):
        return self._overlapping

    def onesided(self):
        """the halo moves by peer stores (onesided_put_* models, > 1 rank)"""
        v = C.c_int()
        call("spmvh_l2g_onesided", self.A.h, C.byref(v))
        return bool(v.value)

    def ghosts(self):
        g = np.zeros(self._num_ghosts, np.int64)
        call("spmvh_l2g_ghosts", self.A.h, _np_ptr(g))
        return g

    def global_to_local(self, i):
        out = i32()
        call("spmvh_l2g_global_to_local", self.A.h, int(i), C.byref(out))
        return out.value

    def plan(self):
        return L2GPlanView(self._nn, self._ni,
                           lambda *a: call("spmvh_l2g_plan", self.A.h, *a))

    def update(self, x_ptr):
        call("spmvh_matrix_update", self.A.h, x_ptr)

    def update_finalise(self, x_ptr):
        call("spmvh_matrix_update_finalise", self.A.h, x_ptr)


class Matrix:
    """spmv::Matrix<double>"""

    def __init__(self, h):
        self.h = h

    @classmethod
    def create_matrix(cls, comm, exec_, rowptr, colind, values, nrows_local,
                      ncols_local, row_ghosts, col_ghosts, symmetric=False,
                      cm=COLLECTIVE_BLOCKING):
        rp = np.ascontiguousarray(rowptr, np.int32)
        ci = np.ascontiguousarray(colind, np.int32)
        va = np.ascontiguousarray(values, np.float64)
        rg = np.ascontiguousarray(row_ghosts, np.int64)
        cg = np.ascontiguousarray(col_ghosts, np.int64)
        h = vp()
        call("spmvh_matrix_create", comm.h, exec_.h, _np_ptr(rp), _np_ptr(ci),
             _np_ptr(va), int(nrows_local), int(ncols_local), _np_ptr(rg),
             len(rg), _np_ptr(cg), len(cg), int(symmetric), cm, C.byref(h))
        return cls(h)

    @classmethod
    def create_poisson3d(cls, comm, exec_, n, symmetric=False,
                         cm=COLLECTIVE_BLOCKING):
        h = vp()
        call("spmvh_matrix_create_poisson3d", comm.h, exec_.h, n,
             int(symmetric), cm, C.byref(h))
        return cls(h)

    @classmethod
    def create_unstructured(cls, comm, exec_, nrows, per_row=7, band=2048,
                            far_permille=100, seed=0x5EED0003):
        """Seeded unstructured test matrix, generated on the device (one rank;
        numpy twin: spmv_amd.poisson.unstructured_csr)."""
        h = vp()
        call("spmvh_matrix_create_unstructured", comm.h, exec_.h, int(nrows),
             int(per_row), int(band), int(far_permille), int(seed), C.byref(h))
        return cls(h)

    @classmethod
    def create_fem_like(cls, comm, exec_, nrows, symmetric=False, **params):
        """Seeded FEM-like test matrix (ragged rows, optional tail of very long
        rows, bandwidth-reducing order), generated on the device (one rank;
        numpy twin and parameters: spmv_amd.poisson.fem_like_csr).  symmetric:
        its strictly lower part + diagonal in symmetric storage."""
        from .poisson import fem_params
        h = vp()
        p = FemParams(**fem_params(nrows, **params))
        call("spmvh_matrix_create_fem_like_sym" if symmetric
             else "spmvh_matrix_create_fem_like", comm.h, exec_.h, C.byref(p),
             C.byref(h))
        return cls(h)

    @classmethod
    def create_poisson3d_boxes(cls, comm, exec_, n, parts, symmetric=False,
                               cm=COLLECTIVE_BLOCKING):
        """The Poisson matrix on a 3-D block partition, parts = (px, py, pz)."""
        h = vp()
        call("spmvh_matrix_create_poisson3d_boxes", comm.h, exec_.h, n,
             int(parts[0]), int(parts[1]), int(parts[2]), int(symmetric), cm,
             C.byref(h))
        return cls(h)

    def close(self):
        if self.h:
            call("spmvh_matrix_destroy", self.h)
            self.h = None

    def release_csr(self):
        """CSRMatrix::release_csr on both blocks: frees the device copies of
        colind / values where the plan holds the matrix in its own format;
        returns the bytes given back (0: the plan still reads them)."""
        v = i64()
        call("spmvh_matrix_release_csr", self.h, C.byref(v))
        return v.value

    def rows(self):
        v = C.c_int()
        call("spmvh_matrix_rows", self.h, C.byref(v))
        return v.value

    def cols(self):
        v = C.c_int()
        call("spmvh_matrix_cols", self.h, C.byref(v))
        return v.value

    def non_zeros(self):
        v = i64()
        call("spmvh_matrix_non_zeros", self.h, C.byref(v))
        return v.value

    def format_size(self):
        v = sz()
        call("spmvh_matrix_format_size", self.h, C.byref(v))
        return v.value

    def symmetric(self):
        v = C.c_int()
        call("spmvh_matrix_symmetric", self.h, C.byref(v))
        return bool(v.value)

    def blocks(self):
        out = (i64 * 6)()
        call("spmvh_matrix_blocks", self.h, out)
        return dict(local=tuple(out[0:3]), remote=tuple(out[3:6]))

    def col_map(self):
        return ColMapView(self)

    def plan_get(self, key, remote=False):
        v = C.c_int()
        call("spmvh_matrix_plan_get", self.h, int(remote), key.encode(),
             C.byref(v))
        return v.value

    def enable_mixed(self):
        ok = C.c_int()
        call("spmvh_matrix_enable_mixed", self.h, C.byref(ok))
        return bool(ok.value)

    def use_mixed(self, on):
        call("spmvh_matrix_use_mixed", self.h, int(bool(on)))

    def plan_set(self, key, value, remote=False):
        call("spmvh_matrix_plan_set", self.h, int(remote), key.encode(),
             int(value))

    def mult(self, x_ptr, y_ptr):
        call("spmvh_matrix_mult", self.h, x_ptr, y_ptr)


class MatrixF32:
    """spmv::Matrix<float> (fp32 instantiation)"""

    def __init__(self, comm, exec_, rowptr, colind, values, nrows_local,
                 ncols_local, row_ghosts, col_ghosts, symmetric=False,
                 cm=COLLECTIVE_BLOCKING):
        rp = np.ascontiguousarray(rowptr, np.int32)
        ci = np.ascontiguousarray(colind, np.int32)
        va = np.ascontiguousarray(values, np.float32)
        rg = np.ascontiguousarray(row_ghosts, np.int64)
        cg = np.ascontiguousarray(col_ghosts, np.int64)
        h = vp()
        call("spmvh_matrix_f32_create", comm.h, exec_.h, _np_ptr(rp),
             _np_ptr(ci), _np_ptr(va), int(nrows_local), int(ncols_local),
             _np_ptr(rg), len(rg), _np_ptr(cg), len(cg), int(symmetric), cm,
             C.byref(h))
        self.h = h

    def info(self):
        rows, nnz, ls, ng = C.c_int(), i64(), i32(), i32()
        call("spmvh_matrix_f32_info", self.h, C.byref(rows), C.byref(nnz),
             C.byref(ls), C.byref(ng))
        return dict(rows=rows.value, nnz=nnz.value, local_size=ls.value,
                    num_ghosts=ng.value)

    def update(self, x_ptr):
        call("spmvh_matrix_f32_update", self.h, x_ptr)

    def mult(self, x_ptr, y_ptr):
        call("spmvh_matrix_f32_mult", self.h, x_ptr, y_ptr)

    def close(self):
        if self.h:
            call("spmvh_matrix_f32_destroy", self.h)
            self.h = None


def _split_result(h, sizes, nrows_local, symmetric):
    out = dict(nnz=sizes[7])
    for which, name in ((0, "local"), (1, "remote")):
        rows, cols, nnz = sizes[3 * which:3 * which + 3]
        brp = np.zeros(rows + 1, np.int32)
        bci = np.zeros(nnz, np.int32)
        bva = np.zeros(nnz, np.float64)
        call("spmvh_split_get", h, which, _np_ptr(brp), _np_ptr(bci),
             _np_ptr(bva))
        out[name] = (brp, bci, bva)
        out[name + "_cols"] = cols
    diag = np.zeros(nrows_local if symmetric else 0, np.float64)
    ghosts = np.zeros(sizes[6], np.int64)
    call("spmvh_split_extra", h, _np_ptr(diag) if symmetric else None,
         _np_ptr(ghosts))
    out["diagonal"] = diag if symmetric else None
    out["ghosts"] = ghosts
    call("spmvh_split_destroy", h)
    return out


def split_rows(rowptr, colind, values, nrows_local, ncols_local, row_offset,
               col_offset, col_ghosts, symmetric, cm):
    """Matrix<double>::split_rows: the host half of create_matrix (no GPU)."""
    rp = np.ascontiguousarray(rowptr, np.int32)
    ci = np.ascontiguousarray(colind, np.int32)
    va = np.ascontiguousarray(values, np.float64)
    cg = np.ascontiguousarray(col_ghosts, np.int64)
    h, sizes = vp(), (i64 * 8)()
    call("spmvh_split_create", _np_ptr(rp), _np_ptr(ci), _np_ptr(va),
         int(nrows_local), int(ncols_local), int(row_offset), int(col_offset),
         _np_ptr(cg), len(cg), int(symmetric), cm, C.byref(h), sizes)
    return _split_result(h, sizes, nrows_local, symmetric)


def split_rows_distributed(comm, rowptr, colind, values, nrows_local,
                           ncols_local, row_ghosts, col_ghosts, symmetric, cm):
    """Matrix<double>::split_rows_distributed: collective over `comm`, ships
    ghost rows to their owners (Matrix.cpp:188-292); no GPU."""
    rp = np.ascontiguousarray(rowptr, np.int32)
    ci = np.ascontiguousarray(colind, np.int32)
    va = np.ascontiguousarray(values, np.float64)
    rg = np.ascontiguousarray(row_ghosts, np.int64)
    cg = np.ascontiguousarray(col_ghosts, np.int64)
    h, sizes = vp(), (i64 * 8)()
    call("spmvh_split_create_dist", comm.h, _np_ptr(rp), _np_ptr(ci),
         _np_ptr(va), int(nrows_local), int(ncols_local), _np_ptr(rg), len(rg),
         _np_ptr(cg), len(cg), int(symmetric), cm, C.byref(h), sizes)
    return _split_result(h, sizes, nrows_local, symmetric)


def poisson3d_box_rows(n, parts, rank):
    """Host half of Matrix.create_poisson3d_boxes: (rowptr, colind, values,
    col_ghosts, global_row_offset, box extents) of one rank; no device."""
    sizes = (i64 * 7)()
    px, py, pz = (int(p) for p in parts)
    call("spmvh_poisson3d_box_rows", n, px, py, pz, rank, sizes, None, None,
         None, None)
    rowptr = np.empty(sizes[0] + 1, np.int32)
    colind = np.empty(sizes[1], np.int32)
    values = np.empty(sizes[1], np.float64)
    ghosts = np.empty(sizes[2], np.int64)
    call("spmvh_poisson3d_box_rows", n, px, py, pz, rank, sizes,
         _np_ptr(rowptr), _np_ptr(colind), _np_ptr(values), _np_ptr(ghosts))
    return rowptr, colind, values, ghosts, int(sizes[3]), tuple(sizes[4:7])


def cg(comm, exec_, A, b_ptr, x_ptr, kmax, rtol, history=True):
    """spmv::cg(comm, exec, A, b, x, kmax, rtol) -> (k, rnorm_history)"""
    k = C.c_int()
    hist = np.zeros(kmax + 1) if history else None
    call("spmvh_cg", comm.h, exec_.h, A.h, b_ptr, x_ptr, kmax, float(rtol),
         C.byref(k), _np_ptr(hist))
    return k.value, (hist[:k.value + 1] if history else None)


def cg_mixed(comm, exec_, A, b_ptr, x_ptr, kmax, rtol, replace_every=50,
             workspace=None, time_spmv=False, consumer_reductions=True):
    """spmv::cg with CgOptions::mixed -> (k, rnorm_history, stats)"""
    k = C.c_int()
    hist = np.zeros(kmax + 2)
    st = (C.c_double * 6)()
    call("spmvh_cg_mixed", comm.h, exec_.h, A.h, b_ptr, x_ptr, kmax,
         float(rtol), int(replace_every), C.byref(k), _np_ptr(hist), len(hist),
         workspace.h if workspace else None,
         int(time_spmv) | (0 if consumer_reductions else 4), st)
    stats = dict(spmv_ms_total=st[0], spmv_launches=int(st[1]),
                 replacements=int(st[2]), true_rel_residual=st[3],
                 continuation_iterations=int(st[4]),
                 final_true_rel_residual=st[5])
    return k.value, hist[:k.value + 1], stats


def read_petsc_binary_matrix(filename, comm, exec_, symmetric=False,
                             cm=COLLECTIVE_BLOCKING):
    """spmv::read_petsc_binary_matrix (spmv/read_petsc.cpp:40-228)"""
    h = vp()
    call("spmvh_read_petsc_matrix", comm.h, exec_.h, str(filename).encode(),
         int(symmetric), cm, C.byref(h))
    return Matrix(h)


def read_petsc_binary_vector(comm, exec_, filename):
    """-> (device pointer, local length); free with exec_.free()"""
    p, n = vp(), i64()
    call("spmvh_read_petsc_vector", comm.h, exec_.h, str(filename).encode(),
         C.byref(p), C.byref(n))
    return p.value or 0, n.value


def read_petsc_binary_rows(filename, rank, size):
    """Host-only parse of one rank's slice (no device)."""
    h, sizes = vp(), (i64 * 7)()
    call("spmvh_petsc_rows_read", str(filename).encode(), rank, size,
         C.byref(h), sizes)
    nloc = sizes[4] - sizes[3]
    rp = np.zeros(nloc + 1, np.int32)
    ci = np.zeros(sizes[5], np.int32)
    va = np.zeros(sizes[5], np.float64)
    gh = np.zeros(sizes[6], np.int64)
    call("spmvh_petsc_rows_get", h, _np_ptr(rp), _np_ptr(ci), _np_ptr(va),
         _np_ptr(gh))
    call("spmvh_petsc_rows_destroy", h)
    return dict(nrows=sizes[0], ncols=sizes[1], nnz=sizes[2],
                row_begin=sizes[3], row_end=sizes[4], rowptr=rp, colind=ci,
                values=va, col_ghosts=gh)


class CgWorkspace:
    """spmv::CgWorkspace: work vectors kept across cg() calls."""

    def __init__(self, exec_):
        h = vp()
        call("spmvh_cg_workspace_create", exec_.h, C.byref(h))
        self.h = h

    def reserve_timing(self, iterations):
        call("spmvh_cg_workspace_reserve_timing", self.h, int(iterations))

    def close(self):
        if self.h:
            call("spmvh_cg_workspace_destroy", self.h)
            self.h = None


def cg_ex(comm, exec_, A, b_ptr, x_ptr, kmax, rtol, workspace=None,
          time_spmv=False, history=False, consumer_reductions=True,
          poll_every=0):
    """cg with the optional arguments: returns (k, history, spmv_ms_total,
    spmv_launches).  poll_every: CgOptions::poll_every (how many iterations the
    host may run ahead of the device's `done` flag; 0 = the default, 16)."""
    k, n = C.c_int(), C.c_int()
    ms = f64()
    hist = np.zeros(kmax + 1) if history else None
    call("spmvh_cg_ex", comm.h, exec_.h, A.h, b_ptr, x_ptr, kmax, float(rtol),
         C.byref(k), _np_ptr(hist), workspace.h if workspace else None,
         int(time_spmv) | (0 if consumer_reductions else 4)
         | ((int(poll_every) & 0xff) << 8), C.byref(ms),
         C.byref(n))
    return (k.value, hist[:k.value + 1] if history else None, ms.value,
            n.value)


def host_executor_rejects_compute():
    return lib.spmvh_host_executor_rejects_compute() == 0
