"""Small host-side helpers shared by the model layer (semantics of implicit/utils.py)."""
import time
import warnings

import numpy as np
import scipy.sparse


class ParameterWarning(Warning):
    """Raised (as a warning) when an input had to be converted (implicit/utils.py:155-156)."""


def check_csr(user_items):
    """Accepts only CSR; anything else is converted with a ParameterWarning (implicit/utils.py:159-169)."""
    if isinstance(user_items, scipy.sparse.csr_matrix):
        return user_items
    kind = type(user_items).__name__
    t0 = time.time()
    converted = user_items.tocsr()
    warnings.warn(
        f"Method expects CSR input, and was passed {kind} instead. "
        f"Converting to CSR took {time.time() - t0} seconds",
        ParameterWarning,
    )
    return converted


def check_random_state(random_state):
    """numpy Generator from None / int / RandomState / Generator (implicit/utils.py:65-83; the
    reference's RandomState branch calls a non-existent `rand_int` -- here it works)."""
    if isinstance(random_state, np.random.RandomState):
        return np.random.default_rng(random_state.randint(2**31))
    return np.random.default_rng(random_state)


def nonzeros(m, row):
    """(index, value) pairs of one CSR row (implicit/utils.py:9-12)."""
    for k in range(m.indptr[row], m.indptr[row + 1]):
        yield m.indices[k], m.data[k]


def random_factors(rng, rows, cols, scale=0.01, workers=None):
    """`rng.random((rows, cols), dtype=float32) * scale` -- the CPU path's initial factors (implicit/cpu/als.py:144-147) --
    with the SAME bits and the same generator state afterwards, drawn by a pool of threads: PCG64 can jump ahead, one 64-bit
    step feeds two float32 draws, so every thread fills its slice from a copy of the generator advanced to the slice's start
    (83 M draws at configs[2]: 0.13 s on one core, the largest item of fit()'s set-up).  Any other bit generator, an odd
    element count or a half-consumed 64-bit word takes the plain call."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    n = rows * cols
    bg = rng.bit_generator
    state = bg.state if type(bg).__name__ == "PCG64" else None
    if state is None or n % 2 or n < (1 << 20) or state.get("has_uint32", 0):
        return rng.random((rows, cols), dtype=np.float32) * scale
    if workers is None:
        workers = max(1, min(16, (os.cpu_count() or 2) // 2))
    out = np.empty(n, dtype=np.float32)
    chunk = ((n // workers + 1) // 2) * 2
    scale32 = np.float32(scale)

    def fill(start):
        stop = min(n, start + chunk)
        g = np.random.PCG64()
        g.state = state
        g.advance(start // 2)
        np.random.Generator(g).random(stop - start, dtype=np.float32, out=out[start:stop])
        out[start:stop] *= scale32

    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(fill, range(0, n, chunk)))
    bg.advance(n // 2)
    return out.reshape(rows, cols)


def transpose_csr(m, threads=0):
    """`m.T.tocsr()` for a canonical float32 CSR with 32-bit offsets, by the library's threaded counting transpose
    (imp_host_csr_transpose: host code, no device involved) -- scipy's single-threaded conversion is a third of fit()'s
    set-up at configs[2].  Anything else (64-bit offsets, other dtypes, unsorted or duplicated entries, library not
    built) goes through scipy."""
    ok = (isinstance(m, scipy.sparse.csr_matrix) and m.dtype == np.float32 and m.indptr.dtype == np.int32
          and m.indices.dtype == np.int32 and m.nnz < 2**31 - 1 and m.has_canonical_format)
    if ok:
        try:
            from .gpu._hip import check, lib

            fn = lib().imp_host_csr_transpose
        except (ImportError, OSError, AttributeError):
            ok = False
    if not ok:
        return m.T.tocsr()
    rows, cols = m.shape
    indptr, indices, data = (np.ascontiguousarray(a) for a in (m.indptr, m.indices, m.data))
    t_indptr = np.empty(cols + 1, dtype=np.int32)
    t_indices = np.empty(m.nnz, dtype=np.int32)
    t_data = np.empty(m.nnz, dtype=np.float32)
    check(fn(rows, cols, m.nnz, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, t_indptr.ctypes.data,
             t_indices.ctypes.data, t_data.ctypes.data, int(threads)))
    out = scipy.sparse.csr_matrix((t_data, t_indices, t_indptr), shape=(cols, rows), copy=False)
    out.has_sorted_indices = True
    out.has_canonical_format = True
    return out
