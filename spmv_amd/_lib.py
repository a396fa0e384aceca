"""ctypes loader for the in-tree native libraries.

libspmv_hip.so  -- HIP kernels + C ABI (include/spmv_hip.h)
libspmv_host.so -- C++17 host mirror of the reference interface + C facade
                   (include/spmv_host_c.h)

There is no fallback: if a library is missing or fails to load, importing the
product raises.  torch is imported first so that this process holds exactly
one HIP runtime (torch bundles libamdhip64.so.7 / librccl.so.1 with the same
SONAMEs as /opt/rocm; whichever is loaded first serves both).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede loading libspmv_hip.so, see above)

# SPMV_AMD_LIBDIR: developer override (the AddressSanitizer build of the host
# mirror lives in gpurun_out/asan (scratch, not shipped), see `make -C spmv_amd/csrc asan`)
_LIBDIR = os.environ.get("SPMV_AMD_LIBDIR") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "lib")


class SpmvHipError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what} failed: [{code}] {error_string(code)}")
        self.code = code


def _load(name):
    path = os.path.join(_LIBDIR, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import "
            f"__graft_entry__ as g; g.build()'` (or make -C spmv_amd/csrc). "
            "There is no CPU fallback.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip = _load("libspmv_hip.so")

vp = C.c_void_p
i32, i64, f64, f32, sz = C.c_int32, C.c_int64, C.c_double, C.c_float, C.c_size_t
P = C.POINTER

_PROTOS = {
    "spmv_hip_abi_version": ([], C.c_int),
    "spmv_hip_error_string": ([C.c_int], C.c_char_p),
    "spmv_hip_device_count": ([P(C.c_int)], C.c_int),
    "spmv_hip_ctx_create": ([C.c_int, P(vp)], C.c_int),
    "spmv_hip_ctx_destroy": ([vp], C.c_int),
    "spmv_hip_ctx_device": ([vp, P(C.c_int)], C.c_int),
    "spmv_hip_num_cus": ([vp, P(C.c_int)], C.c_int),
    "spmv_hip_synchronize": ([vp], C.c_int),
    "spmv_hip_ctx_set_option": ([vp, C.c_char_p, i64], C.c_int),
    "spmv_hip_ctx_get_option": ([vp, C.c_char_p, C.POINTER(i64)], C.c_int),
    "spmv_hip_stream_create": ([vp, P(vp)], C.c_int),
    "spmv_hip_stream_create_priority": ([vp, C.c_int, P(vp)], C.c_int),
    "spmv_hip_stream_destroy": ([vp, vp], C.c_int),
    "spmv_hip_stream_synchronize": ([vp, vp], C.c_int),
    "spmv_hip_set_stream": ([vp, vp], C.c_int),
    "spmv_hip_get_stream": ([vp, P(vp)], C.c_int),
    "spmv_hip_event_create": ([vp, C.c_int, P(vp)], C.c_int),
    "spmv_hip_event_destroy": ([vp, vp], C.c_int),
    "spmv_hip_event_record": ([vp, vp, vp], C.c_int),
    "spmv_hip_event_synchronize": ([vp, vp], C.c_int),
    "spmv_hip_stream_wait_event": ([vp, vp, vp], C.c_int),
    "spmv_hip_event_elapsed_ms": ([vp, vp, vp, P(C.c_float)], C.c_int),
    "spmv_hip_alloc": ([vp, sz, P(vp)], C.c_int),
    "spmv_hip_free": ([vp, vp], C.c_int),
    "spmv_hip_host_alloc": ([vp, sz, P(vp)], C.c_int),
    "spmv_hip_host_free": ([vp, vp], C.c_int),
    "spmv_hip_memset_async": ([vp, vp, C.c_int, sz, vp], C.c_int),
    "spmv_hip_copy_d2d_async": ([vp, vp, vp, sz, vp], C.c_int),
    "spmv_hip_copy_h2d_async": ([vp, vp, vp, sz, vp], C.c_int),
    "spmv_hip_copy_d2h_async": ([vp, vp, vp, sz, vp], C.c_int),
    "spmv_hip_copy_peer_async": ([vp, vp, vp, vp, sz, vp], C.c_int),
    "spmv_hip_csr_plan_create": ([vp, i32, i32, i64, vp, vp, C.c_int, C.c_int,
                                  P(vp)], C.c_int),
    "spmv_hip_csr_plan_destroy": ([vp], C.c_int),
    "spmv_hip_zwalk_table": ([i32, i64, C.c_int, C.c_int, vp, i64, P(i64),
                              P(C.c_int)], C.c_int),
    "spmv_hip_csr_plan_bake_values_f64": ([vp, vp, vp, vp, vp], C.c_int),
    "spmv_hip_csr_plan_bake_values_f32": ([vp, vp, vp, vp, vp], C.c_int),
    "spmv_hip_csr_plan_bake_values_f32f64": ([vp, vp, vp, vp], C.c_int),
    "spmv_hip_csr_plan_values_changed": ([vp, vp, vp], C.c_int),
    "spmv_hip_csr_plan_owns_matrix": ([vp, C.POINTER(C.c_int)], C.c_int),
    "spmv_hip_csr_plan_release_matrix": ([vp, C.c_int], C.c_int),
    "spmv_hip_csr_plan_algo": ([vp, P(C.c_int)], C.c_int),
    "spmv_hip_csr_plan_set": ([vp, C.c_char_p, C.c_int], C.c_int),
    "spmv_hip_csr_plan_get": ([vp, C.c_char_p, P(C.c_int)], C.c_int),
    "spmv_hip_csr_spmv_f64": ([vp, vp, i32, i32, i64, vp, vp, vp, vp, f64, vp,
                               f64, vp, vp, vp], C.c_int),
    "spmv_hip_csr_spmv_f32f64": ([vp, vp, i32, i32, i64, vp, vp, vp, f64, vp,
                                  f64, vp, vp, vp], C.c_int),
    "spmv_hip_convert_f64_f32": ([vp, i64, vp, vp, vp], C.c_int),
    "spmv_hip_cg_init_f64": ([vp, vp, i64, vp, vp, vp, vp, vp], C.c_int),
    "spmv_hip_axpy_f64": ([vp, i64, f64, vp, vp, vp], C.c_int),
    "spmv_hip_cg_residual_f64": ([vp, vp, C.c_int, i64, vp, vp, vp, vp],
                                 C.c_int),
    "spmv_hip_csr_spmv_f32": ([vp, vp, i32, i32, i64, vp, vp, vp, vp, f32, vp,
                               f32, vp, vp], C.c_int),
    "spmv_hip_gather_f64": ([vp, C.c_int, vp, vp, vp, vp], C.c_int),
    "spmv_hip_gather_f32": ([vp, C.c_int, vp, vp, vp, vp], C.c_int),
    "spmv_hip_scatter_add_f64": ([vp, C.c_int, vp, vp, vp, vp], C.c_int),
    "spmv_hip_scatter_add_f32": ([vp, C.c_int, vp, vp, vp, vp], C.c_int),
    "spmv_hip_dot_partials_len": ([vp, P(C.c_int)], C.c_int),
    "spmv_hip_dot_partial_f64": ([vp, i64, vp, vp, vp, vp], C.c_int),
    "spmv_hip_reduce_partials_f64": ([vp, vp, vp, vp], C.c_int),
    "spmv_hip_cg_update_r_cs_f64": ([vp, vp, C.c_int, i64, vp, vp, vp, vp], C.c_int),
    "spmv_hip_cg_update_xp_cs_f64": ([vp, vp, C.c_int, i64, vp, vp, vp, vp], C.c_int),
    "spmv_hip_cg_ws_create": ([vp, C.c_int, P(vp)], C.c_int),
    "spmv_hip_cg_ws_destroy": ([vp], C.c_int),
    "spmv_hip_cg_ws_reset": ([vp, f64, vp], C.c_int),
    "spmv_hip_cg_ws_rr": ([vp, C.c_int, P(vp)], C.c_int),
    "spmv_hip_cg_ws_pAp": ([vp, C.c_int, P(vp)], C.c_int),
    "spmv_hip_cg_ws_partials": ([vp, P(vp)], C.c_int),
    "spmv_hip_cg_ws_done_flag": ([vp, P(vp)], C.c_int),
    "spmv_hip_cg_ws_read_async": ([vp, vp, vp, C.c_size_t, vp], C.c_int),
    "spmv_hip_cg_ws_capacity": ([vp, vp], C.c_int),
    "spmv_hip_cg_update_xr_f64": ([vp, vp, C.c_int, i64, vp, vp, vp, vp, vp],
                                  C.c_int),
    "spmv_hip_cg_update_p_f64": ([vp, vp, C.c_int, i64, vp, vp, vp], C.c_int),
    "spmv_hip_cg_update_r_f64": ([vp, vp, C.c_int, i64, vp, vp, vp], C.c_int),
    "spmv_hip_cg_update_xp_f64": ([vp, vp, C.c_int, i64, vp, vp, vp, vp],
                                  C.c_int),
    "spmv_hip_cg_reduce_rr": ([vp, vp, C.c_int, vp], C.c_int),
    "spmv_hip_cg_reduce_pAp": ([vp, vp, C.c_int, vp], C.c_int),
    "spmv_hip_cg_reduce_pAp2": ([vp, vp, C.c_int, vp, vp], C.c_int),
    "spmv_hip_cg_dot_rr_f64": ([vp, vp, i64, vp, vp], C.c_int),
    "spmv_hip_poisson3d_count": ([vp, i32, i64, i64, C.c_int, vp, P(i64), vp],
                                 C.c_int),
    "spmv_hip_poisson3d_fill_f64": ([vp, i32, i64, i64, C.c_int, vp, vp, vp,
                                     vp, vp], C.c_int),
    "spmv_hip_poisson3d_ghosts": ([i32, i64, i64, P(i64), P(i64)], C.c_int),
    "spmv_hip_poisson3d_box_count": ([vp, i32, vp, vp, C.c_int, vp, P(i64), P(i64),
                                      vp], C.c_int),
    "spmv_hip_poisson3d_box_fill_f64": ([vp, i32, vp, vp, C.c_int, vp, vp, vp, vp,
                                         vp], C.c_int),
    "spmv_hip_put_create": ([vp, sz, P(vp), vp, P(C.c_uint64), P(i64)], C.c_int),
    "spmv_hip_put_connect": ([vp, C.c_int, vp, C.c_uint64, i64, sz, i32, i32, i32,
                              i32, i32, i32, C.c_int], C.c_int),
    "spmv_hip_put_fine_grained": ([vp, P(C.c_int)], C.c_int),
    "spmv_hip_put_label": ([vp, C.c_int, C.c_int, C.c_int], C.c_int),
    "spmv_hip_peer_error_detail": ([vp, C.c_char_p, sz], C.c_int),
    "spmv_hip_put_finish": ([vp], C.c_int),
    "spmv_hip_put_exchange": ([vp, vp, sz, vp, vp, vp], C.c_int),
    "spmv_hip_put_status": ([vp, P(C.c_int)], C.c_int),
    "spmv_hip_put_destroy": ([vp], C.c_int),
    "spmv_hip_reduce_create": ([vp, C.c_int, C.c_int, P(vp), vp, P(C.c_uint64), P(i64),
                                P(C.c_int)], C.c_int),
    "spmv_hip_reduce_connect": ([vp, C.c_int, vp, C.c_uint64, i64, C.c_int], C.c_int),
    "spmv_hip_reduce_sum_f64": ([vp, vp, vp, C.c_int, vp], C.c_int),
    "spmv_hip_reduce_destroy": ([vp], C.c_int),
    "spmv_hip_unstructured_fill_f64": ([vp, i64, C.c_int, i64, C.c_int,
                                        C.c_uint64, vp, vp, vp, vp], C.c_int),
    "spmv_hip_fem_count": ([vp, vp, vp, P(i64), vp], C.c_int),
    "spmv_hip_fem_fill_f64": ([vp, vp, i64, vp, vp, vp, vp], C.c_int),
    "spmv_hip_csr_lower_split_count": ([vp, i32, vp, vp, vp, P(i64), vp], C.c_int),
    "spmv_hip_csr_lower_split_fill_f64": ([vp, i32, vp, vp, vp, vp, vp, vp, vp, vp],
                                          C.c_int),
    "spmv_hip_fill_gaussian_f64": ([vp, i64, i64, i64, vp, vp], C.c_int),
    "spmv_hip_fill_const_f64": ([vp, i64, f64, vp, vp], C.c_int),
    "spmv_hip_comm_unique_id": ([vp], C.c_int),
    "spmv_hip_comm_create": ([vp, C.c_int, C.c_int, vp, P(vp)], C.c_int),
    "spmv_hip_comm_destroy": ([vp], C.c_int),
    "spmv_hip_comm_neighbor_exchange_f64": ([vp, C.c_int, vp, vp, vp, vp, vp,
                                             vp, vp, vp], C.c_int),
    "spmv_hip_comm_neighbor_exchange_f32": ([vp, C.c_int, vp, vp, vp, vp, vp,
                                             vp, vp, vp], C.c_int),
    "spmv_hip_comm_rank": ([vp, P(C.c_int), P(C.c_int)], C.c_int),
    "spmv_hip_comm_info": ([vp, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int),
                            C.c_char_p, C.c_int], C.c_int),
    "spmv_hip_comm_allreduce_sum_f64": ([vp, vp, sz, vp], C.c_int),
    "spmv_hip_comm_allgather_host": ([vp, vp, vp, sz], C.c_int),
}

for _name, (_args, _res) in _PROTOS.items():
    _fn = getattr(hip, _name)  # AttributeError = ABI symbol missing
    _fn.argtypes = _args
    _fn.restype = _res

HIP_SYMBOLS = tuple(_PROTOS)


def error_string(code):
    return hip.spmv_hip_error_string(int(code)).decode()


def check(code, what="spmv_hip call"):
    if code != 0:
        raise SpmvHipError(code, what)


def call(name, *args):
    """Call a C-ABI entry point and raise SpmvHipError on a non-zero code."""
    check(getattr(hip, name)(*args), name)
