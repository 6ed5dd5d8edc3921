"""Python surface of the `implicit.gpu` plug-in, re-hosted on libimplicit_hip.so.

Mirrors, name for name and argument for argument, the classes the reference exposes from its
Cython module implicit/gpu/_cuda.pyx (file:line cited per class): RandomState, KnnQuery, Matrix,
IntVector, CSRMatrix, COOMatrix, LeastSquaresSolver, calculate_norms, get_device_count,
bpr_update.  Host code stays Python; every heavy call goes through the C-ABI with the GIL
released (ctypes.CDLL does that), as the reference does with `with nogil` (_cuda.pyx:79,257,264,271).

NEW relative to the reference: LeastSquaresSolver.least_squares_cholesky (the reference GPU path
has no Cholesky solver) and the Comm class (RCCL exchange for the multi-GPU fit).
"""
import ctypes

import numpy as np

from ..utils import check_csr
from ._hip import check, lib


def _vp(a):
    return ctypes.c_void_p(a.ctypes.data)


class RandomState:
    """_cuda.pyx:25-42"""

    def __init__(self, seed=42):
        self._h = ctypes.c_void_p()
        check(lib().imp_random_create(int(seed), ctypes.byref(self._h)))

    def uniform(self, rows, cols, low=0.0, high=1.0):
        ret = Matrix(None)
        check(lib().imp_random_uniform(self._h, rows, cols, float(low), float(high), ctypes.byref(ret._h)))
        return ret

    def randn(self, rows, cols, mean=0.0, stddev=1.0):
        ret = Matrix(None)
        check(lib().imp_random_randn(self._h, rows, cols, float(mean), float(stddev), ctypes.byref(ret._h)))
        return ret

    def __del__(self):
        if getattr(self, "_h", None):
            lib().imp_random_destroy(self._h)
            self._h = None


class KnnQuery:
    """_cuda.pyx:45-83"""

    def __init__(self, max_temp_memory=0):
        self._h = ctypes.c_void_p()
        check(lib().imp_knn_create(int(max_temp_memory), ctypes.byref(self._h)))

    def topk(self, items, m, k, item_norms=None, query_filter=None, item_filter=None):
        if not isinstance(items, Matrix) or not isinstance(m, Matrix):
            raise TypeError("KnnQuery.topk expects implicit.gpu.Matrix arguments")
        k = int(k)
        rows = m.shape[0]
        indices = np.zeros((rows, k), dtype="int32")
        distances = np.zeros((rows, k), dtype="float32")
        check(lib().imp_knn_topk(
            self._h, items._h, m._h, k, _vp(indices), _vp(distances),
            item_norms._h if item_norms is not None else None,
            query_filter._h if query_filter is not None else None,
            item_filter._h if item_filter is not None else None))
        return indices, distances

    def topk_device(self, items, m, k, item_norms=None, query_filter=None, item_filter=None):
        """NEW: the same call with DEVICE output buffers (knn.cu:40-54,147-164 detects where its output pointers live): returns
        (ids, distances) as two rows x k device Matrix objects; the ids are int32 bit patterns in a 4-byte-per-element
        matrix (`ids.to_numpy().view(np.int32)`).  For callers that keep post-processing on the device."""
        if not isinstance(items, Matrix) or not isinstance(m, Matrix):
            raise TypeError("KnnQuery.topk expects implicit.gpu.Matrix arguments")
        k = int(k)
        rows = m.shape[0]
        ids, dist = Matrix.zeros(rows, k), Matrix.zeros(rows, k)
        check(lib().imp_knn_topk(
            self._h, items._h, m._h, k, ctypes.c_void_p(ids.device_ptr), ctypes.c_void_p(dist.device_ptr),
            item_norms._h if item_norms is not None else None,
            query_filter._h if query_filter is not None else None,
            item_filter._h if item_filter is not None else None))
        return ids, dist

    def __del__(self):
        if getattr(self, "_h", None):
            lib().imp_knn_destroy(self._h)
            self._h = None


class Matrix:
    """_cuda.pyx:85-207.  Always truthy when constructed (the reference's cdef class defines no
    __len__/__bool__, and the model layer relies on `if self.item_factors`)."""

    def __init__(self, X):
        self._h = ctypes.c_void_p()
        self._keepalive = None
        if X is None:
            return
        cai = getattr(X, "__cuda_array_interface__", None)
        if cai:
            shape = cai["shape"]
            data = cai["data"][0]
            typestr = cai["typestr"]
            if typestr not in ("<f4", "<f2", "=f4", "=f2", "|f4", "|f2"):
                raise ValueError(f"unhandled dtype for GPU Matrix {typestr} (float32 / float16 only)")
            if len(shape) != 2:
                raise ValueError("Matrix expects a 2 dimensional array")
            itemsize = int(typestr[2])
            strides = cai.get("strides")
            if strides is not None and tuple(strides) != (shape[1] * itemsize, itemsize):
                raise ValueError("Matrix can only wrap C-contiguous device arrays")
            # the library works on its own non-blocking stream: order it after whatever the producer still has in
            # flight (the reference runs on the legacy default stream, which is implicitly ordered)
            synchronize()
            check(lib().imp_matrix_wrap_device(shape[0], shape[1], ctypes.c_void_p(data), itemsize,
                                               ctypes.byref(self._h)))
            self._keepalive = X  # no ownership of the memory: keep the exporter alive
            return
        if not hasattr(X, "dtype") or X.dtype.char not in ("f", "e"):
            raise ValueError(f"unhandled dtype for GPU Matrix {getattr(X, 'dtype', type(X))}")
        if X.ndim != 2:
            raise ValueError("Matrix expects a 2 dimensional array")
        X = np.ascontiguousarray(X)
        check(lib().imp_matrix_create(X.shape[0], X.shape[1], _vp(X), X.dtype.itemsize, ctypes.byref(self._h)))

    @classmethod
    def zeros(cls, rows, cols):
        ret = cls(None)
        check(lib().imp_matrix_create(rows, cols, None, 4, ctypes.byref(ret._h)))
        return ret

    def _dims(self):
        r, c, i = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().imp_matrix_shape(self._h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(i)))
        return r.value, c.value, i.value

    @property
    def shape(self):
        r, c, _ = self._dims()
        return r, c

    @property
    def itemsize(self):
        return self._dims()[2]

    @property
    def dtype(self):
        return np.dtype(np.float32) if self.itemsize == 4 else np.dtype(np.float16)

    @property
    def device_ptr(self):
        p = ctypes.c_void_p()
        check(lib().imp_matrix_device_ptr(self._h, ctypes.byref(p)))
        return p.value

    def __getitem__(self, idx):
        ret = Matrix(None)
        ret._keepalive = self._keepalive
        rows = self.shape[0]
        if isinstance(idx, slice):
            if idx.step and idx.step != 1:
                raise ValueError(f"Can't slice matrix with step {idx.step} yet")
            start = idx.start if idx.start is not None else 0
            stop = idx.stop if idx.stop is not None else rows
            if start < 0 or stop < 0:
                start, stop, _ = idx.indices(rows)
            check(lib().imp_matrix_slice(self._h, start, stop, ctypes.byref(ret._h)))
        elif isinstance(idx, int):
            if idx < 0:
                raise ValueError("row index out of bounds for matrix")
            check(lib().imp_matrix_row(self._h, idx, ctypes.byref(ret._h)))
        else:
            try:
                idx = np.array(idx).astype("int32")
            except Exception:
                raise IndexError(f"don't know how to handle __getitem__ on {idx}") from None
            if len(idx.shape) == 0:
                idx = idx.reshape([1])
            if len(idx.shape) != 1:
                raise IndexError(f"don't know how to handle __getitem__ on {idx} - shape={idx.shape}")
            if ((idx < 0) | (idx >= rows)).any():
                raise IndexError("row id out of range for selecting items from matrix")
            ids = IntVector(idx)
            check(lib().imp_matrix_gather(self._h, ids._h, ctypes.byref(ret._h)))
        return ret

    def assign_rows(self, rowids, other):
        rows = IntVector(np.array(rowids).astype("int32"))
        check(lib().imp_matrix_assign_rows(self._h, rows._h, other._h))

    def astype(self, dtype):
        dtype = np.dtype(dtype)
        allowed = (np.float16, np.float32)
        if dtype not in allowed:
            raise ValueError(f"Invalid dtype '{dtype}' for GPU model. Allowed dtypes are: {allowed}")
        ret = Matrix(None)
        check(lib().imp_matrix_astype(self._h, dtype.itemsize, ctypes.byref(ret._h)))
        return ret

    def resize(self, rows, cols):
        check(lib().imp_matrix_resize(self._h, int(rows), int(cols)))

    def to_numpy(self):
        r, c, itemsize = self._dims()
        if itemsize == 4:
            ret = np.zeros((r, c), dtype="float32")
        elif itemsize == 2:
            ret = np.zeros((r, c), dtype="float16")
        else:
            raise ValueError(f"Invalid itemsize {itemsize}")
        check(lib().imp_matrix_to_host(self._h, _vp(ret)))
        return ret

    def copy_from_numpy(self, X):
        """NEW: overwrite in place from a host array of identical shape/dtype (parity harness)."""
        X = np.ascontiguousarray(X)
        if X.shape != self.shape or X.dtype != self.dtype:
            raise ValueError("shape/dtype mismatch in Matrix.copy_from_numpy")
        check(lib().imp_matrix_from_host(self._h, _vp(X)))

    def copy_rows_from(self, dst_row, other, src_row, rows):
        """NEW: device-to-device copy of whole rows other[src_row : src_row + rows] -> self[dst_row : ...] on the library
        stream (the exchange between logical ranks sharing one device, local_comm.py)."""
        check(lib().imp_matrix_copy_rows(self._h, int(dst_row), other._h, int(src_row), int(rows)))

    def __repr__(self):
        return f"Matrix({str(self.to_numpy())})"

    def __str__(self):
        return str(self.to_numpy())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().imp_matrix_destroy(self._h)
            self._h = None


class IntVector:
    """_cuda.pyx:211-218"""

    def __init__(self, data):
        data = np.ascontiguousarray(data)
        if data.dtype != np.int32 or data.ndim != 1:
            raise ValueError("IntVector expects a 1-d int32 buffer")
        self._h = ctypes.c_void_p()
        self.size = len(data)
        check(lib().imp_intvector_create(_vp(data), len(data), ctypes.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().imp_intvector_destroy(self._h)
            self._h = None


def _int32_buffer(a, what):
    a = np.asarray(a)
    if a.dtype != np.int32:
        # the reference binds `cdef int[:] indptr = X.indptr`: a non-int32 buffer is a ValueError
        raise ValueError(f"Buffer dtype mismatch, expected 'int' but got '{a.dtype}' for {what}")
    return np.ascontiguousarray(a)


class CSRMatrix:
    """_cuda.pyx:221-234"""

    def __init__(self, X):
        X = check_csr(X)
        data = np.ascontiguousarray(X.data.astype(np.float32, copy=False))
        self._h = ctypes.c_void_p()
        self.shape = X.shape
        self.nnz = len(data)
        if np.asarray(X.indptr).dtype == np.int64:
            # NEW: the reference binds int32 buffers only (a ValueError for int64); scipy switches both index arrays to
            # int64 once nnz >= 2^31, and the CPU solver accepts that (_als.pyx:76).  Column ids always fit int32.
            indptr = np.ascontiguousarray(X.indptr)
            indices = np.asarray(X.indices)
            if indices.dtype != np.int32:
                if len(indices) and (indices.max() > np.iinfo(np.int32).max or indices.min() < 0):
                    raise ValueError("column index out of range for CSRMatrix")
                indices = indices.astype(np.int32)
            indices = np.ascontiguousarray(indices)
            check(lib().imp_csr_create64(X.shape[0], X.shape[1], len(data), _vp(indptr), _vp(indices), _vp(data),
                                         ctypes.byref(self._h)))
            return
        indptr = _int32_buffer(X.indptr, "indptr")
        indices = _int32_buffer(X.indices, "indices")
        check(lib().imp_csr_create(X.shape[0], X.shape[1], len(data), _vp(indptr), _vp(indices), _vp(data),
                                   ctypes.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().imp_csr_destroy(self._h)
            self._h = None


class COOMatrix:
    """_cuda.pyx:236-247"""

    def __init__(self, X):
        row = _int32_buffer(X.row, "row")
        col = _int32_buffer(X.col, "col")
        data = np.ascontiguousarray(X.data.astype(np.float32))
        self._h = ctypes.c_void_p()
        self.shape = X.shape
        check(lib().imp_coo_create(X.shape[0], X.shape[1], len(data), _vp(row), _vp(col), _vp(data),
                                   ctypes.byref(self._h)))

    @classmethod
    def from_csr_pattern(cls, X):
        """NEW: the (row, col) pattern of a scipy CSR matrix as a device COO without its values -- all the top-k filters read
        (knn.cu:197-214 reads row / col only).  Skips scipy's tocoo() and the value upload: recommend() builds one per batch."""
        self = cls.__new__(cls)
        indptr = np.ascontiguousarray(X.indptr)
        if indptr.dtype not in (np.int32, np.int64):
            indptr = indptr.astype(np.int64)
        col = np.ascontiguousarray(X.indices, dtype=np.int32)
        self._h = ctypes.c_void_p()
        self.shape = X.shape
        # indptr + indices go to page-locked staging and a kernel expands the row ids: no np.repeat, no blocking uploads
        check(lib().imp_coo_create_from_csr_pattern(X.shape[0], X.shape[1], _vp(indptr), int(indptr.dtype == np.int64), _vp(col),
                                                    ctypes.byref(self._h)))
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            lib().imp_coo_destroy(self._h)
            self._h = None


class LeastSquaresSolver:
    """_cuda.pyx:250-275"""

    def __init__(self):
        self._h = ctypes.c_void_p()
        check(lib().imp_solver_create(ctypes.byref(self._h)))

    def least_squares(self, cui, X, YtY, Y, cg_steps):
        check(lib().imp_solver_least_squares(self._h, cui._h, X._h, YtY._h, Y._h, int(cg_steps)))

    def least_squares_cholesky(self, cui, X, YtY, Y, regularization):
        """NEW: mirrors the CPU implicit.cpu._als._least_squares(YtY, ..., regularization): YtY is
        the UNregularised gramian; raises ValueError on a non-positive-definite row."""
        failed = ctypes.c_int64(-1)
        try:
            check(lib().imp_solver_least_squares_cholesky(self._h, cui._h, X._h, YtY._h, Y._h,
                                                          float(regularization), ctypes.byref(failed)))
        except ValueError as e:
            e.failed_row = failed.value  # -1 unless the factorisation itself failed (then: the smallest failing row)
            raise

    def calculate_loss(self, cui, X, Y, regularization):
        loss = ctypes.c_float(0)
        check(lib().imp_solver_calculate_loss(self._h, cui._h, X._h, Y._h, float(regularization),
                                              ctypes.byref(loss)))
        return loss.value

    def calculate_yty(self, Y, YtY, regularization):
        check(lib().imp_solver_calculate_yty(self._h, Y._h, YtY._h, float(regularization)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().imp_solver_destroy(self._h)
            self._h = None


def calculate_norms(items):
    """_cuda.pyx:278-281: returns a (1, rows) Matrix."""
    ret = Matrix(None)
    check(lib().imp_matrix_calculate_norms(items._h, ctypes.byref(ret._h)))
    return ret


def get_device_count():
    """_cuda.pyx:284-285: raises RuntimeError when no device is usable."""
    n = ctypes.c_int(0)
    check(lib().imp_get_device_count(ctypes.byref(n)))
    return n.value


def bpr_update(*args, **kwargs):
    """_cuda.pyx:288-297.  BPR is outside this build's hot path."""
    raise NotImplementedError("bpr_update is not part of the MI355X ALS hot path")


class Comm:
    """NEW: RCCL communicator, one process per GPU (include/implicit_hip.h, imp_comm_*)."""

    UNIQUE_ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(Comm.UNIQUE_ID_BYTES)
        check(lib().imp_comm_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id, nranks, rank):
        self._h = ctypes.c_void_p()
        self.nranks, self.rank = int(nranks), int(rank)
        buf = ctypes.create_string_buffer(bytes(unique_id), Comm.UNIQUE_ID_BYTES)
        check(lib().imp_comm_init_rank(buf, self.nranks, self.rank, ctypes.byref(self._h)))

    def allreduce_sum(self, m):
        check(lib().imp_comm_allreduce_sum(self._h, m._h))

    def allgather_rows(self, full, row_offsets):
        offs = (ctypes.c_int64 * (self.nranks + 1))(*[int(o) for o in row_offsets])
        check(lib().imp_comm_allgather_rows(self._h, full._h, offs))

    def allgather_rows_begin(self, full, row_lo, row_hi):
        """Queue the exchange of rows [row_lo[r], row_hi[r]) (owner: rank r) behind the work queued so far; returns
        immediately.  Pair with allgather_rows_end()."""
        lo = (ctypes.c_int64 * self.nranks)(*[int(o) for o in row_lo])
        hi = (ctypes.c_int64 * self.nranks)(*[int(o) for o in row_hi])
        check(lib().imp_comm_allgather_rows_begin(self._h, full._h, lo, hi))

    def allgather_rows_end(self):
        check(lib().imp_comm_allgather_rows_end(self._h))

    def alltoall_rows(self, send, send_lo, send_hi, recv, recv_lo, recv_hi):
        """Rows [send_lo[p], send_hi[p]) of `send` go to rank p and land in rows [recv_lo[q], recv_hi[q]) of its `recv`
        (q = the sender).  Blocking; bytes travel untouched."""
        arr = lambda v: (ctypes.c_int64 * self.nranks)(*[int(o) for o in v])  # noqa: E731
        check(lib().imp_comm_alltoall_rows(self._h, send._h, arr(send_lo), arr(send_hi), recv._h, arr(recv_lo), arr(recv_hi)))

    def barrier(self):
        check(lib().imp_comm_barrier(self._h))

    def ranks_seen(self):
        """(ranks counted on the library stream's communicator, ranks counted on the exchange stream's): both must equal
        `nranks`, else the rendezvous wired fewer processes together than were launched."""
        out = (ctypes.c_int * 2)(0, 0)
        check(lib().imp_comm_ranks_seen(self._h, out))
        return int(out[0]), int(out[1])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().imp_comm_destroy(self._h)
            self._h = None


def set_device(device):
    check(lib().imp_set_device(int(device)))


def get_device():
    d = ctypes.c_int(0)
    check(lib().imp_get_device(ctypes.byref(d)))
    return d.value


def set_oversubscribe(factor):
    """Workgroups launched per resident slot by the persistent row kernels (1 = exactly what the device holds; the
    multi-GPU driver uses 4 so that slots held by RCCL's kernels only delay small shares)."""
    check(lib().imp_set_oversubscribe(int(factor)))


def get_oversubscribe():
    n = ctypes.c_int(0)
    check(lib().imp_get_oversubscribe(ctypes.byref(n)))
    return n.value


def set_deferred_sync(on):
    """Deferred mode of the current device: calculate_yty / least_squares / Comm.allreduce_sum only queue their work; one
    synchronize() orders (and checks) everything queued.  set_deferred_sync(False) waits and restores the default."""
    check(lib().imp_set_deferred_sync(1 if on else 0))


def release_workspaces():
    """Free the library's per-device scratch buffers (they are re-allocated on demand); fit() calls it when it returns."""
    check(lib().imp_release_workspaces())


def debug_occupy(workgroups, microseconds):
    """Measurement aid: park `workgroups` spinning workgroups on the device for about `microseconds`."""
    check(lib().imp_debug_occupy(int(workgroups), int(microseconds)))


def core_clock_mhz(microseconds=50):
    """Measurement aid: the shader core clock (MHz) seen by a probe kernel queued behind the work already on the stream;
    microseconds < 0: by a probe running BESIDE the work queued in deferred mode (the clock those kernels run at)."""
    mhz = ctypes.c_double(0.0)
    check(lib().imp_debug_core_clock(int(microseconds), ctypes.byref(mhz)))
    return mhz.value


def synchronize():
    check(lib().imp_device_synchronize())


def fixup_rows(reset=True):
    """Rows of CG sweeps on the current device that a fast kernel handed to the fp32 fix-up kernel since the last reset (a lost
    cluster exchange, or long rows whose matrix-core operands left the fp16 range): results are correct either way, a non-zero
    count is a performance signal."""
    n = ctypes.c_ulonglong(0)
    check(lib().imp_solver_fixup_rows(ctypes.byref(n), 1 if reset else 0))
    return int(n.value)


class Profiler:
    """Per-kernel HIP-event timing on the library stream (imp_prof_*), for bench.py's roofline leg."""

    @staticmethod
    def enable(on=True, only=None):
        """`only`: time just the kernels whose name contains this substring (an event pair costs stream time)."""
        check(lib().imp_prof_filter((only or "").encode()))
        check(lib().imp_prof_enable(1 if on else 0))

    @staticmethod
    def reset():
        check(lib().imp_prof_reset())

    @staticmethod
    def get(kernel):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        check(lib().imp_prof_get(kernel.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    @staticmethod
    def names():
        buf = ctypes.create_string_buffer(4096)
        check(lib().imp_prof_names(buf, 4096))
        return [s for s in buf.value.decode().split("\n") if s]
