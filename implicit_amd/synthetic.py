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


# ---- block-composed matrices for the multi-GPU driver ---------------------------------------------------------------
#
# A global (users x items) matrix is DEFINED as a grid x grid composition of independent blocks: block (s, t) covers user
# range s and item range t (equal splits), holds ~nnz / grid^2 entries, and is a deterministic function of
# (seed, s, t) alone -- so a rank can produce exactly the blocks it owns (its user rows: one block row; its item rows:
# one block column) without generating, or communicating, anything else.  Every block has the same power-law item
# popularity, i.e. popular items are spread evenly over the item ranges (what relabelling item ids round-robin by
# popularity does to a real catalogue), so contiguous item shards carry equal work.  The matrix depends on `grid`, not
# on the number of ranks: with grid = 8, runs on 1 / 2 / 4 / 8 ranks factorise the SAME matrix.


def grid_bounds(total, grid):
    return (np.arange(grid + 1, dtype=np.int64) * total) // grid


def grid_block(users_total, items_total, nnz_total, grid, s, t, gamma=2.0, seed=42, sigma=1.0):
    """Block (s, t) as scipy CSR (block users x block items), float32 / int32, sorted indices, de-duplicated."""
    ub, ib = grid_bounds(users_total, grid), grid_bounds(items_total, grid)
    nu, ni = int(ub[s + 1] - ub[s]), int(ib[t + 1] - ib[t])
    # per-user degree: log-normal with the global mean nnz / users (the same for every item block of this user range)
    deg = np.random.default_rng([seed, 1, s]).lognormal(mean=0.0, sigma=sigma, size=nu)
    deg *= (nnz_total / users_total) / np.exp(sigma * sigma / 2.0) / grid
    rng = np.random.default_rng([seed, 2, s, t])
    cnt = np.minimum(np.floor(deg + rng.random(nu)).astype(np.int64), ni)  # randomised rounding of the block's share
    total = int(cnt.sum())
    rows = np.repeat(np.arange(nu, dtype=np.int64), cnt)
    cols = np.minimum((ni * rng.random(total) ** gamma).astype(np.int64), ni - 1)
    key = np.unique(rows * ni + cols)
    data = (1.0 + 4.0 * rng.random(len(key), dtype=np.float32)).astype(np.float32)
    indptr = np.zeros(nu + 1, dtype=np.int64)
    indptr[1:] = np.bincount(key // ni, minlength=nu)
    m = sp.csr_matrix((data, (key % ni).astype(np.int32), np.cumsum(indptr).astype(np.int32)), shape=(nu, ni))
    m.has_sorted_indices = True
    return m


def _as_int32_csr(m):
    m = m.tocsr()
    m.sort_indices()
    if m.indices.dtype != np.int32:
        m.indices = m.indices.astype(np.int32)
    if m.indptr.dtype != np.int32 and m.nnz < 2**31:
        m.indptr = m.indptr.astype(np.int32)
    return m


def grid_shards(rank, nranks, users_total, items_total, nnz_total, grid, gamma=2.0, seed=42, workers=None):
    """Rank `rank`'s pieces of the block-composed matrix: (Cui_shard, Ciu_shard, u_offsets, i_offsets).

    Cui_shard: the rank's user rows x ALL items (global item ids); Ciu_shard: the rank's item rows x ALL users (global
    user ids); *_offsets: the nranks + 1 row offsets of the shards (every rank computes the same ones).  Blocks are
    independent, so they are generated (and transposed) by a pool of `workers` threads (default: up to 16; numpy's sort
    and scipy's conversions release the GIL) -- one rank of one holds the whole matrix: 64 blocks at grid = 8."""
    import concurrent.futures
    import os

    if grid % nranks:
        raise ValueError(f"the block grid ({grid}) must be a multiple of the number of ranks ({nranks})")
    per = grid // nranks
    ub, ib = grid_bounds(users_total, grid), grid_bounds(items_total, grid)
    mine = range(rank * per, (rank + 1) * per)
    row_blocks = [(s, t) for s in mine for t in range(grid)]
    col_blocks = [(s, t) for t in mine for s in range(grid)]
    wanted = sorted(set(row_blocks) | set(col_blocks))
    if workers is None:
        workers = max(1, min(16, (os.cpu_count() or 2) // (2 * max(1, nranks)), len(wanted)))

    def make(st):
        s, t = st
        b = grid_block(users_total, items_total, nnz_total, grid, s, t, gamma, seed)
        return st, b, (b.T.tocsr() if st in col_set else None)

    col_set = set(col_blocks)
    with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as ex:
        made = {st: (b, bt) for st, b, bt in ex.map(make, wanted)}

        def stack_rows(s):
            return sp.hstack([made[(s, t)][0] for t in range(grid)], format="csr")

        def stack_cols(t):
            return sp.hstack([made[(s, t)][1] for s in range(grid)], format="csr")

        cui_rows = list(ex.map(stack_rows, mine))
        ciu_rows = list(ex.map(stack_cols, mine))
    del made
    cui = sp.vstack(cui_rows, format="csr") if len(cui_rows) > 1 else cui_rows[0]
    ciu = sp.vstack(ciu_rows, format="csr") if len(ciu_rows) > 1 else ciu_rows[0]
    u_off = ub[::per].copy()
    i_off = ib[::per].copy()
    return _as_int32_csr(cui), _as_int32_csr(ciu), u_off, i_off
