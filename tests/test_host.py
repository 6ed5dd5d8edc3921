"""Host-side logic that needs no GPU: synthetic generator, CSR checks, filter remapping, sharding plan."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp
from numpy.testing import assert_array_equal


def test_synthetic_csr_is_canonical_and_deterministic():
    from implicit_amd.synthetic import synthetic_csr

    a = synthetic_csr(500, 300, 6000, seed=3, neg_frac=0.1, empty_frac=0.05)
    b = synthetic_csr(500, 300, 6000, seed=3, neg_frac=0.1, empty_frac=0.05)
    assert a.shape == (500, 300) and a.indices.dtype == np.int32 and a.indptr.dtype == np.int32 and a.dtype == np.float32
    assert (a != b).nnz == 0
    assert a.has_canonical_format
    assert 0.8 * 6000 < a.nnz <= 6000 * 1.05
    lens = np.diff(a.indptr)
    assert (lens == 0).any() and (a.data < 0).any()
    assert np.abs(a.data).min() >= 1.0 and np.abs(a.data).max() <= 5.0


def test_check_csr_warns_and_converts():
    from implicit_amd.utils import ParameterWarning, check_csr

    m = sp.random(5, 4, density=0.5, format="csr", dtype=np.float32, random_state=0)
    assert check_csr(m) is m
    with pytest.warns(ParameterWarning):
        out = check_csr(m.tocoo())
    assert isinstance(out, sp.csr_matrix)


def test_filter_items_remap_matches_reference_semantics():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from implicit_amd.gpu.matrix_factorization_base import _filter_items_from_sparse_matrix

    liked = sp.csr_matrix(np.array([[1, 0, 1, 0, 1, 0], [0, 1, 0, 0, 0, 1]], dtype=np.float32))
    items = np.array([1, 2, 4])
    out = _filter_items_from_sparse_matrix(items, liked).toarray()
    # row 0 liked {0,2,4} -> positions of 2 and 4 in items = {1,2}; row 1 liked {1,5} -> position of 1 = {0}
    assert_array_equal(out[:, :3] != 0, [[False, True, True], [True, False, False]])


def test_check_random_state():
    from implicit_amd.utils import check_random_state

    a = check_random_state(5).random(3)
    b = check_random_state(5).random(3)
    assert_array_equal(a, b)
    assert isinstance(check_random_state(np.random.RandomState(1)), np.random.Generator)


def test_grid_shards_compose_one_matrix_whatever_the_rank_count():
    """The block-composed generator of the multi-GPU bench: a rank's user rows and item rows are consistent pieces of
    ONE global matrix, and the matrix depends on the grid, not on how many ranks share it."""
    from implicit_amd.synthetic import grid_shards

    users, items, nnz, grid = 3000, 800, 40_000, 4
    whole = {}
    for n in (1, 2, 4):
        parts = [grid_shards(r, n, users, items, nnz, grid, gamma=2.0, seed=3) for r in range(n)]
        cui = sp.vstack([p[0] for p in parts]).tocsr()
        ciu = sp.vstack([p[1] for p in parts]).tocsr()
        assert cui.shape == (users, items) and ciu.shape == (items, users)
        assert abs(cui - ciu.T).nnz == 0                      # item rows = transpose of the user rows
        for p in parts:
            assert p[0].indices.dtype == np.int32 and p[0].indptr.dtype == np.int32 and p[0].has_sorted_indices
            assert_array_equal(p[2], parts[0][2])             # every rank computes the same offsets
            assert p[2][0] == 0 and p[2][-1] == users and p[3][-1] == items
        whole[n] = cui
    assert abs(whole[1] - whole[2]).nnz == 0 and abs(whole[1] - whole[4]).nnz == 0
    assert 0.8 * nnz < whole[1].nnz <= 1.05 * nnz
    # popular items are spread over the item ranges: the shards carry comparable work
    per_shard = np.diff(whole[1].tocsc().indptr).reshape(4, -1).sum(axis=1)
    assert per_shard.max() < 1.3 * per_shard.min()
    with pytest.raises(ValueError):
        grid_shards(0, 3, users, items, nnz, grid)


def test_bench_keeps_stdout_for_the_json_line():
    """bench.py's contract is ONE JSON line on stdout; whatever a library prints at C level (librccl's version banner
    surfaces at process exit) must land on stderr: fd 1 is re-pointed at stderr and the line goes to a private duplicate
    of the original stdout (bench.protect_stdout / emit)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import bench, os, sys\n"
            "saved = bench.protect_stdout()\n"
            "print('python-level noise')\n"
            "os.write(1, b'C-level noise\\n')\n"
            "bench.emit(saved, {'metric': 'x', 'value': 1})\n"
            "os.write(1, b'late C-level noise\\n')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout == '{"metric": "x", "value": 1}\n'
    assert "python-level noise" in out.stderr and "late C-level noise" in out.stderr


def test_threaded_initial_factors_are_the_sequential_draw():
    """utils.random_factors: the CPU path's `rng.random((n, f), float32) * 0.01` drawn by several threads -- same bits, and the
    generator ends in the same state (the item factors are drawn right after the user factors)."""
    from implicit_amd.utils import random_factors

    a, b = np.random.default_rng(5), np.random.default_rng(5)
    for rows, cols in ((9000, 128), (8191, 64), (33, 7), (20_001, 64)):   # the last two take the plain path (odd / small counts)
        want = a.random((rows, cols), dtype=np.float32) * 0.01
        got = random_factors(b, rows, cols, workers=4)
        assert got.dtype == np.float32 and got.shape == (rows, cols)
        np.testing.assert_array_equal(got, want)
    assert a.random() == b.random()
    legacy = np.random.Generator(np.random.MT19937(3))                    # not PCG64: plain path
    np.testing.assert_array_equal(random_factors(legacy, 8192, 128), np.random.Generator(np.random.MT19937(3)).random((8192, 128), dtype=np.float32) * 0.01)


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_threaded_transpose_equals_scipy(threads):
    """utils.transpose_csr (imp_host_csr_transpose, host code): the same CSR as `m.T.tocsr()`, rows sorted inside every
    output row, empty rows / columns kept; non-canonical or 64-bit input takes scipy."""
    from implicit_amd.synthetic import synthetic_csr
    from implicit_amd.utils import transpose_csr

    for C in (synthetic_csr(3000, 1700, 90_000, seed=2, empty_frac=0.1, neg_frac=0.1), synthetic_csr(40, 5000, 70_000, seed=3),
              sp.csr_matrix((7, 9), dtype=np.float32)):
        want = C.T.tocsr()
        got = transpose_csr(C, threads=threads)
        assert got.shape == want.shape and got.indptr.dtype == np.int32
        assert_array_equal(got.indptr, want.indptr)
        assert_array_equal(got.indices, want.indices)
        assert_array_equal(got.data, want.data)
    C64 = synthetic_csr(300, 200, 4000, seed=4)
    C64.indptr = C64.indptr.astype(np.int64)
    assert (transpose_csr(C64) != C64.T.tocsr()).nnz == 0          # scipy path
    bad = synthetic_csr(300, 200, 4000, seed=4)
    bad.indices[0] = 5000                                           # out of range: refused, not scattered
    bad.has_canonical_format = True
    with pytest.raises(ValueError):
        transpose_csr(bad, threads=threads)
    bent = synthetic_csr(300, 200, 4000, seed=4)                    # a decreasing indptr: refused before anything is scattered
    bent.indptr[10], bent.indptr[11] = bent.indptr[11], bent.indptr[10]
    bent.has_canonical_format = True
    with pytest.raises(ValueError):
        transpose_csr(bent, threads=threads)
