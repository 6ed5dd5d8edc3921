"""ctypes loader for libimplicit_hip.so (the C-ABI in include/implicit_hip.h).

No CPU fallback exists: if the library is missing or a call fails the error surfaces
(ImportError / the mapped Python exception)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree library: no environment variable redirects it.  Measurement tooling that wants a compile-time
# variant of the same library sets implicit_amd._libpath.OVERRIDE before importing this package (bench.py does, from IMP_LIB_PATH).
from .. import _libpath  # noqa: E402

LIB_PATH = _libpath.OVERRIDE or os.path.join(os.path.dirname(_HERE), "libimplicit_hip.so")

c_void_pp = ctypes.POINTER(ctypes.c_void_p)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_f32_p = ctypes.POINTER(ctypes.c_float)

IMP_OK, IMP_INVALID_ARGUMENT, IMP_OUT_OF_RANGE, IMP_RUNTIME_ERROR = 0, 1, 2, 3

# name -> argtypes ; every function returns int (status) unless listed in _RESTYPES
_SIGNATURES = {
    "imp_get_device_count": [ctypes.POINTER(ctypes.c_int)],
    "imp_set_device": [ctypes.c_int],
    "imp_get_device": [ctypes.POINTER(ctypes.c_int)],
    "imp_set_oversubscribe": [ctypes.c_int],
    "imp_get_oversubscribe": [ctypes.POINTER(ctypes.c_int)],
    "imp_set_deferred_sync": [ctypes.c_int],
    "imp_debug_occupy": [ctypes.c_int, ctypes.c_int],
    "imp_debug_core_clock": [ctypes.c_int, ctypes.POINTER(ctypes.c_double)],
    "imp_device_synchronize": [],
    "imp_solver_fixup_rows": [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int],
    "imp_mem_get_info": [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)],
    "imp_release_workspaces": [],
    "imp_host_csr_transpose": [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int],
    "imp_matrix_create": [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, c_void_pp],
    "imp_matrix_wrap_device": [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, c_void_pp],
    "imp_matrix_row": [ctypes.c_void_p, ctypes.c_size_t, c_void_pp],
    "imp_matrix_slice": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, c_void_pp],
    "imp_matrix_gather": [ctypes.c_void_p, ctypes.c_void_p, c_void_pp],
    "imp_matrix_resize": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t],
    "imp_matrix_assign_rows": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    "imp_matrix_astype": [ctypes.c_void_p, ctypes.c_size_t, c_void_pp],
    "imp_matrix_calculate_norms": [ctypes.c_void_p, c_void_pp],
    "imp_matrix_to_host": [ctypes.c_void_p, ctypes.c_void_p],
    "imp_matrix_from_host": [ctypes.c_void_p, ctypes.c_void_p],
    "imp_matrix_copy_rows": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t],
    "imp_matrix_shape": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t),
                         ctypes.POINTER(ctypes.c_size_t)],
    "imp_matrix_device_ptr": [ctypes.c_void_p, c_void_pp],
    "imp_matrix_destroy": [ctypes.c_void_p],
    "imp_intvector_create": [ctypes.c_void_p, ctypes.c_size_t, c_void_pp],
    "imp_intvector_destroy": [ctypes.c_void_p],
    "imp_csr_create": [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_void_p, c_void_pp],
    "imp_csr_create64": [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                         ctypes.c_void_p, c_void_pp],
    "imp_csr_shape": [ctypes.c_void_p, c_i32_p, c_i32_p, ctypes.POINTER(ctypes.c_int64)],
    "imp_csr_destroy": [ctypes.c_void_p],
    "imp_coo_create": [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_void_p, c_void_pp],
    "imp_coo_create_from_csr_pattern": [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.POINTER(ctypes.c_void_p)],
    "imp_coo_destroy": [ctypes.c_void_p],
    "imp_solver_create": [c_void_pp],
    "imp_solver_destroy": [ctypes.c_void_p],
    "imp_solver_calculate_yty": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float],
    "imp_solver_least_squares": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_int],
    "imp_solver_least_squares_cholesky": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_int64)],
    "imp_solver_calculate_loss": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_float, c_f32_p],
    "imp_knn_create": [ctypes.c_size_t, c_void_pp],
    "imp_knn_destroy": [ctypes.c_void_p],
    "imp_knn_topk": [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    "imp_random_create": [ctypes.c_int64, c_void_pp],
    "imp_random_destroy": [ctypes.c_void_p],
    "imp_random_uniform": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_float, ctypes.c_float,
                           c_void_pp],
    "imp_random_randn": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_float, ctypes.c_float,
                         c_void_pp],
    "imp_comm_unique_id": [ctypes.c_void_p],
    "imp_comm_init_rank": [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_void_pp],
    "imp_comm_destroy": [ctypes.c_void_p],
    "imp_comm_allreduce_sum": [ctypes.c_void_p, ctypes.c_void_p],
    "imp_comm_allgather_rows": [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)],
    "imp_comm_allgather_rows_begin": [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                      ctypes.POINTER(ctypes.c_int64)],
    "imp_comm_allgather_rows_end": [ctypes.c_void_p],
    "imp_comm_alltoall_rows": [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                               ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)],
    "imp_comm_barrier": [ctypes.c_void_p],
    "imp_comm_ranks_seen": [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)],
    "imp_prof_enable": [ctypes.c_int],
    "imp_prof_filter": [ctypes.c_char_p],
    "imp_prof_reset": [],
    "imp_prof_get": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)],
    "imp_prof_names": [ctypes.c_char_p, ctypes.c_size_t],
}
_RESTYPES = {"imp_last_error": ctypes.c_char_p, "imp_version": ctypes.c_char_p}

EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_RESTYPES))

_lib = None


def lib():
    """The loaded library; raises ImportError (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m implicit_amd._build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        for name, restype in _RESTYPES.items():
            fn = getattr(L, name)
            fn.argtypes = []
            fn.restype = restype
        _lib = L
    return _lib


def check(status):
    """Maps a C-ABI status onto the exception the reference's Cython binding would raise."""
    if status == IMP_OK:
        return
    msg = (lib().imp_last_error() or b"").decode("utf-8", "replace")
    if status == IMP_INVALID_ARGUMENT:
        raise ValueError(msg)
    if status == IMP_OUT_OF_RANGE:
        raise IndexError(msg)
    raise RuntimeError(msg)
