"""Thin Python handles over the C ABI of libspmv_hip.so (include/spmv_hip.h).

Harness plumbing for tests/ and bench.py: owns contexts, device buffers,
streams and plans, and forwards every compute call 1:1 to the C ABI.  No
arithmetic happens in Python and nothing here falls back to numpy/torch.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import call

ALGO_AUTO, ALGO_ROWBLOCK, ALGO_VECTOR, ALGO_SCALAR, ALGO_ROWLIST = 0, 1, 2, 3, 4
PART_ALL, PART_LOCAL, PART_REMOTE, PART_LOCAL_LOWER = 0, 1, 2, 3


def device_count():
    n = C.c_int()
    call("spmv_hip_device_count", C.byref(n))
    return n.value


class Buffer:
    """A device allocation made through spmv_hip_alloc."""

    def __init__(self, ctx, nbytes, dtype=None, count=None):
        self.ctx, self.nbytes, self.dtype, self.count = ctx, int(nbytes), dtype, count
        p = C.c_void_p()
        call("spmv_hip_alloc", ctx.h, self.nbytes, C.byref(p))
        self.ptr = p.value  # None for a zero-byte allocation

    def free(self):
        # (freed = True: given back already, the address kept as a token --
        # CsrBlock.release_matrix)
        if self.ptr and not getattr(self, "freed", False):
            call("spmv_hip_free", self.ctx.h, self.ptr)
        self.ptr = None

    def numpy(self, count=None, offset=0):
        """Blocking device->host copy of `count` elements from `offset`."""
        count = self.count - offset if count is None else count
        out = np.empty(count, self.dtype)
        if count:
            isz = np.dtype(self.dtype).itemsize
            call("spmv_hip_copy_d2h_async", self.ctx.h,
                 out.ctypes.data_as(C.c_void_p), self.ptr + offset * isz,
                 count * isz, None)
            self.ctx.stream_sync()
        return out

    def at(self, offset):
        return self.ptr + offset * np.dtype(self.dtype).itemsize

    def write(self, arr):
        """Blocking host->device copy INTO this allocation (same length)."""
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.size == self.count
        if self.nbytes:
            call("spmv_hip_copy_h2d_async", self.ctx.h, self.ptr,
                 arr.ctypes.data_as(C.c_void_p), self.nbytes, None)
            self.ctx.stream_sync()


class Context:
    """spmv_hip_ctx: one GPU."""

    def __init__(self, device_id=0):
        h = C.c_void_p()
        call("spmv_hip_ctx_create", device_id, C.byref(h))
        self.h = h
        self.device_id = device_id

    def close(self):
        if self.h:
            call("spmv_hip_ctx_destroy", self.h)
            self.h = None

    @property
    def num_cus(self):
        n = C.c_int()
        call("spmv_hip_num_cus", self.h, C.byref(n))
        return n.value

    @property
    def dot_partials_len(self):
        n = C.c_int()
        call("spmv_hip_dot_partials_len", self.h, C.byref(n))
        return n.value

    # -- memory ---------------------------------------------------------
    def empty(self, count, dtype):
        return Buffer(self, int(count) * np.dtype(dtype).itemsize, dtype,
                      int(count))

    def zeros(self, count, dtype):
        b = self.empty(count, dtype)
        if b.nbytes:
            call("spmv_hip_memset_async", self.h, b.ptr, 0, b.nbytes, None)
        return b

    def upload(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        b = self.empty(arr.size, arr.dtype)
        if b.nbytes:
            call("spmv_hip_copy_h2d_async", self.h, b.ptr,
                 arr.ctypes.data_as(C.c_void_p), b.nbytes, None)
            self.stream_sync()  # arr may be a temporary
        return b

    def copy_h2d(self, dst_ptr, arr):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            call("spmv_hip_copy_h2d_async", self.h, dst_ptr,
                 arr.ctypes.data_as(C.c_void_p), arr.nbytes, None)
            self.stream_sync()

    def copy(self, dst, src, nbytes, stream=None):
        call("spmv_hip_copy_d2d_async", self.h, dst, src, nbytes, stream)

    def memset(self, ptr, value, nbytes, stream=None):
        call("spmv_hip_memset_async", self.h, ptr, value, nbytes, stream)

    # -- streams / events -------------------------------------------------
    def set_option(self, key, value):
        call("spmv_hip_ctx_set_option", self.h, key.encode(), int(value))

    def synchronize(self):
        call("spmv_hip_synchronize", self.h)

    def stream_sync(self, stream=None):
        call("spmv_hip_stream_synchronize", self.h, stream)

    def stream_create(self):
        s = C.c_void_p()
        call("spmv_hip_stream_create", self.h, C.byref(s))
        return s

    def stream_destroy(self, s):
        call("spmv_hip_stream_destroy", self.h, s)

    def set_stream(self, s):
        call("spmv_hip_set_stream", self.h, s)

    def event_create(self, timing=True):
        e = C.c_void_p()
        call("spmv_hip_event_create", self.h, int(timing), C.byref(e))
        return e

    def event_destroy(self, e):
        call("spmv_hip_event_destroy", self.h, e)

    def event_record(self, e, stream=None):
        call("spmv_hip_event_record", self.h, e, stream)

    def event_sync(self, e):
        call("spmv_hip_event_synchronize", self.h, e)

    def elapsed_ms(self, e0, e1):
        ms = C.c_float()
        call("spmv_hip_event_elapsed_ms", self.h, e0, e1, C.byref(ms))
        return ms.value

    # -- kernels ------------------------------------------------------------
    def gather(self, indices, x, out, n, stream=None):
        call("spmv_hip_gather_f64", self.h, n, indices.ptr, x.ptr, out.ptr,
             stream)

    def scatter_add(self, indices, x, out, n, stream=None, dtype=np.float64):
        fn = ("spmv_hip_scatter_add_f64" if dtype == np.float64
              else "spmv_hip_scatter_add_f32")
        call(fn, self.h, n, indices.ptr, x.ptr, out.ptr, stream)

    def fill_gaussian(self, N, i_begin, count, x_ptr, stream=None):
        call("spmv_hip_fill_gaussian_f64", self.h, N, i_begin, count, x_ptr,
             stream)

    def fill_const(self, count, value, x_ptr, stream=None):
        call("spmv_hip_fill_const_f64", self.h, count, float(value), x_ptr,
             stream)

    def dot(self, n, x_ptr, y_ptr):
        """Deterministic two-stage dot product; blocking, returns a float."""
        part = self.empty(self.dot_partials_len, np.float64)
        res = self.empty(1, np.float64)
        call("spmv_hip_dot_partial_f64", self.h, n, x_ptr, y_ptr, part.ptr,
             None)
        call("spmv_hip_reduce_partials_f64", self.h, part.ptr, res.ptr, None)
        v = float(res.numpy()[0])
        part.free()
        res.free()
        return v


class CsrBlock:
    """Device-resident CSR block + its SpMV plan (what CSRMatrix owns,
    spmv/csr_matrix.cpp:22-70)."""

    def __init__(self, ctx, nrows, ncols, rowptr, colind, values,
                 diagonal=None, symmetric=False, algo=ALGO_AUTO,
                 dtype=np.float64):
        self.ctx, self.nrows, self.ncols = ctx, int(nrows), int(ncols)
        self.symmetric, self.dtype = bool(symmetric), np.dtype(dtype)
        self.owned = []

        def dev(a, dt):
            if a is None or isinstance(a, Buffer):
                return a
            b = ctx.upload(a, dt)
            self.owned.append(b)
            return b
        self.nnz = (values.count if isinstance(values, Buffer)
                    else (0 if values is None else len(values)))
        if self.nnz == 0:  # csr_matrix.cpp:34: arrays are never allocated
            rowptr = colind = values = None
        self.rowptr, self.colind = dev(rowptr, np.int32), dev(colind, np.int32)
        self.values, self.diagonal = dev(values, dtype), dev(diagonal, dtype)
        plan = C.c_void_p()
        call("spmv_hip_csr_plan_create", ctx.h, self.nrows, self.ncols,
             self.nnz, _p(self.rowptr), _p(self.colind), int(self.symmetric),
             algo, C.byref(plan))
        self.plan = plan

    def bake(self, drop=False):
        """spmv_hip_csr_plan_bake_values_*: the plan's own copy of the values
        by offset (symmetric diagonal form)."""
        name = ("spmv_hip_csr_plan_bake_values_f64" if self.dtype == np.float64
                else "spmv_hip_csr_plan_bake_values_f32")
        call(name, self.ctx.h, self.plan, None if drop else _p(self.values),
             None if drop else _p(self.diagonal), None)

    def values_changed(self):
        """spmv_hip_csr_plan_values_changed: the baked arrays were rewritten in
        place; refresh the plan's copies."""
        call("spmv_hip_csr_plan_values_changed", self.ctx.h, self.plan, None)

    def owns_matrix(self):
        """spmv_hip_csr_plan_owns_matrix: bit 0 = colind, bit 1 = values are no
        longer read by the plan's kernel"""
        m = C.c_int()
        call("spmv_hip_csr_plan_owns_matrix", self.plan, C.byref(m))
        return m.value

    def release_matrix(self):
        """spmv_hip_csr_plan_release_matrix + free this block's device copies of
        the arrays the plan owns (the buffers' addresses stay as the tokens the
        launches compare).  Returns the mask released."""
        m = self.owns_matrix()
        if m == 3:
            call("spmv_hip_csr_plan_release_matrix", self.plan, m)
            self.ctx.synchronize()
            for buf in (self.colind, self.values):
                call("spmv_hip_free", self.ctx.h, buf.ptr)
                buf.freed = True
            return m
        return 0

    def set(self, key, value):
        call("spmv_hip_csr_plan_set", self.plan, key.encode(), int(value))

    def get(self, key):
        v = C.c_int()
        call("spmv_hip_csr_plan_get", self.plan, key.encode(), C.byref(v))
        return v.value

    @property
    def algo(self):
        a = C.c_int()
        call("spmv_hip_csr_plan_algo", self.plan, C.byref(a))
        return a.value

    def mult(self, alpha, x_ptr, beta, y_ptr, dot_partials=None, stream=None):
        if self.dtype == np.float64:
            call("spmv_hip_csr_spmv_f64", self.ctx.h, self.plan, self.nrows,
                 self.ncols, self.nnz, _p(self.rowptr), _p(self.colind),
                 _p(self.values), _p(self.diagonal), float(alpha), x_ptr,
                 float(beta), y_ptr, dot_partials, stream)
        else:
            assert dot_partials is None
            call("spmv_hip_csr_spmv_f32", self.ctx.h, self.plan, self.nrows,
                 self.ncols, self.nnz, _p(self.rowptr), _p(self.colind),
                 _p(self.values), _p(self.diagonal), float(alpha), x_ptr,
                 float(beta), y_ptr, stream)

    def free(self):
        if self.plan:
            call("spmv_hip_csr_plan_destroy", self.plan)
            self.plan = None
        for b in self.owned:
            b.free()
        self.owned = []


def _p(buf):
    return None if buf is None else buf.ptr


def poisson3d_block(ctx, n, row_begin, row_end, part, with_diagonal=False,
                    algo=ALGO_AUTO):
    """Generate a Poisson CSR block directly on the device
    (spmv_hip_poisson3d_count / _fill_f64)."""
    nrows = row_end - row_begin
    gb, ga = C.c_int64(), C.c_int64()
    call("spmv_hip_poisson3d_ghosts", n, row_begin, row_end, C.byref(gb),
         C.byref(ga))
    ncols = nrows + gb.value + ga.value
    if part in (PART_LOCAL, PART_LOCAL_LOWER):
        ncols_blk = nrows if part == PART_LOCAL else ncols
    else:
        ncols_blk = ncols
    rowptr = ctx.empty(nrows + 1, np.int32)
    nnz = C.c_int64()
    call("spmv_hip_poisson3d_count", ctx.h, n, row_begin, row_end, part,
         rowptr.ptr, C.byref(nnz), None)
    colind = ctx.empty(nnz.value, np.int32)
    values = ctx.empty(nnz.value, np.float64)
    diag = ctx.empty(nrows, np.float64) if with_diagonal else None
    call("spmv_hip_poisson3d_fill_f64", ctx.h, n, row_begin, row_end, part,
         rowptr.ptr, colind.ptr, values.ptr, _p(diag), None)
    blk = CsrBlock(ctx, nrows, ncols_blk, rowptr, colind, values, diag,
                   symmetric=(part == PART_LOCAL_LOWER), algo=algo)
    blk.owned += [b for b in (rowptr, colind, values, diag) if b is not None]
    blk.ghosts_below, blk.ghosts_above = gb.value, ga.value
    return blk
