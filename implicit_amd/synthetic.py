"""Deterministic synthetic implicit-feedback matrices.

The reference's datasets (implicit/datasets/*.py) download HDF5 files; there is no network
here, so benchmarks and parity tests use shape-matched synthetic CSR: log-normal row degrees,
power-law item popularity, de-duplicated sorted int32 indices, confidences 1 + 4*U(0,1).
"""
import numpy as np
import scipy.sparse as sp

# name -> (users, items, nnz target, popularity exponent gamma)
SHAPES = {
    "ml100k": (943, 1682, 100_000, 2.0),            # BASELINE config 1
    "c2": (1_000_000, 100_000, 50_000_000, 2.0),    # BASELINE config 2
    "lastfm360k": (358_868, 292_385, 17_500_000, 3.0),  # BASELINE config 3
    "c4": (10_000_000, 1_000_000, 500_000_000, 2.0),  # BASELINE config 4 (8 GPUs)
    "ml20m": (138_493, 26_744, 20_000_000, 2.0),    # BASELINE config 5
}


def synthetic_csr(users, items, nnz, gamma=2.0, seed=42, neg_frac=0.0, empty_frac=0.0,
                  sigma=1.0, col_offset=0, total_items=None):
    """Returns a canonical scipy CSR (users x items) float32 / int32.

    Row degrees ~ log-normal(sigma) scaled so they sum to ~nnz, clipped to [1, items];
    columns floor(items * r**gamma); duplicates dropped, so the actual nnz is a little lower
    than requested (callers report the actual value).  `neg_frac` of the entries get a negative
    confidence (the reference's "disliked" branch, _als.pyx:117-118) and `empty_frac` of the
    rows are emptied (the zero-row branch, _als.pyx:98-100).
    """
    rng = np.random.default_rng(seed)
    deg = rng.lognormal(mean=0.0, sigma=sigma, size=users)
    deg = np.clip(np.rint(deg * (nnz / deg.sum())), 1, items).astype(np.int64)
    if empty_frac > 0:
        deg[rng.random(users) < empty_frac] = 0
    total = int(deg.sum())
    rows = np.repeat(np.arange(users, dtype=np.int64), deg)
    cols = np.minimum((items * rng.random(total) ** gamma).astype(np.int64), items - 1)
    key = np.unique(rows * items + cols)
    rows = key // items
    cols = (key % items).astype(np.int32)
    data = (1.0 + 4.0 * rng.random(len(key), dtype=np.float32)).astype(np.float32)
    if neg_frac > 0:
        data[rng.random(len(key)) < neg_frac] *= -1
    indptr = np.zeros(users + 1, dtype=np.int64)
    indptr[1:] = np.bincount(rows, minlength=users)
    indptr = np.cumsum(indptr).astype(np.int32)
    shape_items = total_items if total_items is not None else items
    m = sp.csr_matrix((data, cols + np.int32(col_offset), indptr), shape=(users, shape_items))
    m.has_sorted_indices = True
    return m


def named(name, seed=42, scale=1.0, **kw):
    users, items, nnz, gamma = SHAPES[name]
    if scale != 1.0:
        users, items, nnz = max(int(users * scale), 8), max(int(items * scale), 8), max(int(nnz * scale), 8)
    return synthetic_csr(users, items, nnz, gamma=gamma, seed=seed, **kw)
