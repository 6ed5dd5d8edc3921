"""numpy-facing wrapper over oracle/liboracle.so (plain-C restatement of the reference's
Cython CPU path) -- TEST INFRASTRUCTURE ONLY, see oracle/als_oracle.c for the file:line map.

Function names and argument order mirror implicit/cpu/_als.pyx and implicit/cpu/topk.pyx so
parity tests read like the reference's own tests.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "als_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.oracle_calculate_loss.restype = ctypes.c_double
        L.oracle_least_squares_chol.restype = ctypes.c_int64
        L.oracle_num_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(_f32p)


def _i(a):
    return a.ctypes.data_as(_i32p)


def _csr(Cui):
    indptr = np.ascontiguousarray(Cui.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(Cui.indices, dtype=np.int32)
    data = np.ascontiguousarray(Cui.data, dtype=np.float32)
    return indptr, indices, data


def _check(a, name):
    if a.dtype != np.float32 or not a.flags.c_contiguous:
        raise ValueError(f"{name} must be C-contiguous float32")


def num_threads():
    return lib().oracle_num_threads()


def gramian(Y):
    """np.dot(Y.T, Y)  (_als.pyx:70)"""
    _check(Y, "Y")
    out = np.zeros((Y.shape[1], Y.shape[1]), dtype=np.float32)
    lib().oracle_gramian(_f(Y), ctypes.c_int64(Y.shape[0]), Y.shape[1], _f(out))
    return out


def least_squares_cg(Cui, X, Y, regularization, num_threads=0, cg_steps=3, YtY=None):
    """_als.least_squares_cg(Cui, X, Y, regularization, num_threads, cg_steps); X in place.

    `YtY` optionally overrides the *regularised* gramian A0 (what the GPU boundary is handed)."""
    _check(X, "X"), _check(Y, "Y")
    f = X.shape[1]
    if YtY is None:
        # regularization is a C float in the reference (_als.pyx:153)
        YtY = gramian(Y) + np.float32(regularization) * np.eye(f, dtype=np.float32)
    YtY = np.ascontiguousarray(YtY, dtype=np.float32)
    indptr, indices, data = _csr(Cui)
    lib().oracle_least_squares_cg(_i(indptr), _i(indices), _f(data), ctypes.c_int64(X.shape[0]),
                                  _f(X), _f(Y), _f(YtY), f, int(cg_steps), int(num_threads))


def least_squares_cg_f64(Cui, X, Y, regularization, num_threads=0, cg_steps=3):
    """fp64 evaluation of the same CG sweep from fp32 inputs; returns a float64 array (X untouched).
    Measures the fp32 oracle's own rounding noise."""
    _check(X, "X"), _check(Y, "Y")
    X64 = np.ascontiguousarray(X, dtype=np.float64)
    indptr, indices, data = _csr(Cui)
    lib().oracle_least_squares_cg_f64(_i(indptr), _i(indices), _f(data), ctypes.c_int64(X.shape[0]),
                                      X64.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _f(Y),
                                      ctypes.c_int64(Y.shape[0]), X.shape[1],
                                      ctypes.c_double(np.float32(regularization)), int(cg_steps),
                                      int(num_threads))
    return X64


def least_squares(Cui, X, Y, regularization, num_threads=0, YtY=None):
    """_als.least_squares (Cholesky); `YtY` optionally supplies the UNregularised gramian
    (as _als._least_squares's first argument, _als.pyx:75)."""
    _check(X, "X"), _check(Y, "Y")
    f = X.shape[1]
    if YtY is None:
        YtY = gramian(Y)
    YtY = np.ascontiguousarray(YtY, dtype=np.float32)
    indptr, indices, data = _csr(Cui)
    err = ctypes.c_int(0)
    failed = lib().oracle_least_squares_chol(_f(YtY), _i(indptr), _i(indices), _f(data),
                                             ctypes.c_int64(X.shape[0]), _f(X), _f(Y), f,
                                             ctypes.c_double(regularization), int(num_threads),
                                             ctypes.byref(err))
    if failed:
        raise ValueError("cython_lapack.posv failed (err=%i) on row %i. Try "
                         "increasing the regularization parameter." % (err.value, failed - 1))


def calculate_loss(Cui, X, Y, regularization, num_threads=0):
    _check(X, "X"), _check(Y, "Y")
    indptr, indices, data = _csr(Cui)
    YtY = gramian(Y)
    return lib().oracle_calculate_loss(_i(indptr), _i(indices), _f(data), ctypes.c_int64(X.shape[0]),
                                       ctypes.c_int64(Y.shape[0]), ctypes.c_int64(Cui.nnz), _f(X), _f(Y),
                                       _f(YtY), X.shape[1], ctypes.c_float(regularization),
                                       int(num_threads))


def select(batch, k):
    """implicit::select (select.h:12-40) on a dense score matrix."""
    batch = np.ascontiguousarray(batch, dtype=np.float32)
    rows, cols = batch.shape
    ids = np.zeros((rows, k), dtype=np.int32)
    dist = np.zeros((rows, k), dtype=np.float32)
    lib().oracle_select(_f(batch), rows, cols, int(k), _i(ids), _f(dist))
    return ids, dist


def topk(items, query, k, item_norms=None, filter_query_items=None, filter_items=None, num_threads=0):
    """implicit.cpu.topk.topk (topk.pyx:15-67)."""
    if query.ndim == 1:
        query = query.reshape(1, -1)
    items = np.ascontiguousarray(items, dtype=np.float32)
    query = np.ascontiguousarray(query, dtype=np.float32)
    nq = query.shape[0]
    ids = np.zeros((nq, k), dtype=np.int32)
    dist = np.zeros((nq, k), dtype=np.float32)
    norms = None if item_norms is None else np.ascontiguousarray(item_norms, dtype=np.float32)
    fp = fi = None
    if filter_query_items is not None:
        fp = np.ascontiguousarray(filter_query_items.indptr, dtype=np.int32)
        fi = np.ascontiguousarray(filter_query_items.indices, dtype=np.int32)
    fit = None if filter_items is None else np.ascontiguousarray(filter_items, dtype=np.int32)
    lib().oracle_topk(_f(items), items.shape[0], _f(query), nq, items.shape[1], int(k),
                      None if norms is None else _f(norms),
                      None if fp is None else _i(fp), None if fi is None else _i(fi),
                      None if fit is None else _i(fit), 0 if fit is None else len(fit),
                      _i(ids), _f(dist), int(num_threads))
    return ids, dist


def norms(Y):
    _check(Y, "Y")
    out = np.zeros(Y.shape[0], dtype=np.float32)
    lib().oracle_norms(_f(Y), ctypes.c_int64(Y.shape[0]), Y.shape[1], _f(out))
    return out


def fit(user_items, factors, regularization=0.01, alpha=1.0, iterations=15, use_cg=True, cg_steps=3,
        random_state=None, user_factors=None, item_factors=None, num_threads=0, callback=None):
    """The fit() glue of implicit/cpu/als.py:98-202 over the oracle solvers: float32 cast, alpha
    scaling of the matrix (:133-134), transpose (:137), init order users-then-items with
    rng.random(shape, float32) * 0.01 (:144-147), user sweep then item sweep (:164-177)."""
    import time

    rng = np.random.default_rng(random_state)
    Cui = user_items.tocsr().astype(np.float32)
    if alpha != 1.0:
        Cui = alpha * Cui
    Ciu = Cui.T.tocsr()
    items, users = Ciu.shape
    X = user_factors if user_factors is not None else rng.random((users, factors), dtype=np.float32) * 0.01
    Y = item_factors if item_factors is not None else rng.random((items, factors), dtype=np.float32) * 0.01
    X = np.ascontiguousarray(X, dtype=np.float32)
    Y = np.ascontiguousarray(Y, dtype=np.float32)
    for it in range(iterations):
        s = time.time()
        if use_cg:
            least_squares_cg(Cui, X, Y, regularization, num_threads, cg_steps)
            least_squares_cg(Ciu, Y, X, regularization, num_threads, cg_steps)
        else:
            least_squares(Cui, X, Y, regularization, num_threads)
            least_squares(Ciu, Y, X, regularization, num_threads)
        if callback:
            callback(it, time.time() - s, None)
    return X, Y
