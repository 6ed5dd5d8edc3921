"""GPU parity tests for KnnQuery.topk / calculate_norms: read like the reference's tests/gpu_test.py
(ids exact vs numpy argsort, distances rtol 1e-6) plus oracle comparisons with norms and filters."""
import numpy as np
import pytest
import scipy.sparse as sp
from numpy.testing import assert_allclose, assert_array_equal

pytestmark = pytest.mark.gpu


def _check_knn_queries(gpu, items, queries, k=5, max_temp_memory=500_000_000):
    knn = gpu.KnnQuery(max_temp_memory=max_temp_memory)
    ids, distances = knn.topk(gpu.Matrix(items), gpu.Matrix(queries), k)
    batch = queries.dot(items.T)
    exact_ids = np.flip(np.argsort(batch)[:, -k:], axis=1)
    exact_distances = np.zeros(exact_ids.shape)
    for r in range(batch.shape[0]):
        exact_distances[r] = batch[r][exact_ids[r]]
    assert_allclose(distances, exact_distances, rtol=1e-06)
    assert_array_equal(ids, exact_ids)


@pytest.mark.parametrize("k", [4, 16, 64, 128, 1000])
@pytest.mark.parametrize("batch", [1, 10, 100])
@pytest.mark.parametrize("temp_memory", [500_000_000, 5_000_000])
def test_topk_ascending(gpu, k, batch, temp_memory):
    """tests/gpu_test.py:9-19"""
    num_items, factors = 10000, 10
    items = np.arange(num_items * factors).reshape((num_items, factors)).astype("float32")
    queries = np.arange(batch * factors).reshape((batch, factors)).astype("float32")
    _check_knn_queries(gpu, items, queries, k, max_temp_memory=temp_memory)


@pytest.mark.parametrize("k", [4, 64])
@pytest.mark.parametrize("batch", [1, 10, 100])
@pytest.mark.parametrize("temp_memory", [500_000_000, 500_000])
def test_topk_random(gpu, k, batch, temp_memory):
    """tests/gpu_test.py:21-33"""
    rs = np.random.default_rng(0)
    items = rs.random(size=(1000, 10), dtype="float32")
    queries = rs.random(size=(batch, 10), dtype="float32")
    _check_knn_queries(gpu, items, queries, k, max_temp_memory=temp_memory)


def test_calculate_norms(gpu):
    """tests/gpu_test.py:54-65"""
    items = np.arange(100 * 8).reshape((100, 8)).astype("float32")
    norms = gpu.calculate_norms(gpu.Matrix(items)).to_numpy().reshape(100)
    assert_allclose(norms, np.linalg.norm(items, axis=1))
    z = np.zeros((3, 5), dtype=np.float32)
    assert_allclose(gpu.calculate_norms(gpu.Matrix(z)).to_numpy().reshape(3), 1e-10)


def _near_tie_rows(scores_sorted, f):
    """rows where two adjacent retained scores are closer than the fp32 accumulation noise"""
    gaps = np.abs(np.diff(scores_sorted.astype(np.float64), axis=1))
    scale = np.abs(scores_sorted[:, :-1]).astype(np.float64) + 1e-30
    return (gaps < 4 * np.finfo(np.float32).eps * f * scale).any(axis=1)


@pytest.mark.parametrize("f,k", [(64, 10), (128, 10), (128, 100)])
def test_topk_vs_oracle_with_filters(gpu, oracle, f, k):
    rng = np.random.default_rng(11)
    n_items, n_q = 20_000, 300
    items = (rng.standard_normal((n_items, f)) * 0.1).astype(np.float32)
    query = (rng.standard_normal((n_q, f)) * 0.1).astype(np.float32)
    liked = sp.random(n_q, n_items, density=0.002, format="csr", dtype=np.float32, random_state=3)
    filter_items = np.array([0, 5, 17, n_items - 1], dtype=np.int32)
    norms = oracle.norms(items)
    for use_norms in (False, True):
        want_ids, want_d = oracle.topk(items, query, k + 1, item_norms=norms if use_norms else None,
                                       filter_query_items=liked, filter_items=filter_items)
        knn = gpu.KnnQuery()
        ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(query), k,
                          item_norms=gpu.Matrix(norms.reshape(1, -1)) if use_norms else None,
                          query_filter=gpu.COOMatrix(liked.tocoo()),
                          item_filter=gpu.IntVector(filter_items))
        audit = _near_tie_rows(want_d, f)  # k+1 columns: includes the boundary gap
        ok = ~audit
        print(f"f={f} k={k} norms={use_norms}: {audit.sum()} near-tie rows of {n_q} audited separately")
        assert_array_equal(ids[ok], want_ids[ok, :k])
        assert_allclose(d[ok], want_d[ok, :k], rtol=2e-5, atol=1e-7)
        # near-tie rows must still hold the same id SET up to the swapped neighbours
        for r in np.nonzero(audit)[0]:
            assert len(set(ids[r]) ^ set(want_ids[r, :k])) <= 2


def test_topk_ties_and_all_filtered_tail(gpu):
    """Documented tie rule: (score desc, column desc); k > items.rows writes items.rows entries."""
    items = np.zeros((6, 4), dtype=np.float32)
    items[:, 0] = [5, 5, 9, 5, 1, 9]
    q = np.array([[1, 0, 0, 0]], dtype=np.float32)
    knn = gpu.KnnQuery()
    ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(q), 4)
    assert_array_equal(ids[0], [5, 2, 3, 1])
    assert_allclose(d[0], [9, 9, 5, 5])
    ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(q), 8)
    assert_array_equal(ids[0][:6], [5, 2, 3, 1, 0, 4])
    assert_array_equal(ids[0][6:], [0, 0])
    assert_allclose(d[0][6:], [0, 0])
    # filtered entries score -FLT_MAX and come last
    ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(q), 6, item_filter=gpu.IntVector(np.array([2, 5], dtype=np.int32)))
    assert set(ids[0][:4]) == {0, 1, 3, 4} and set(ids[0][4:]) == {2, 5}
    assert (d[0][4:] == -np.finfo(np.float32).max).all()


def test_topk_argument_errors(gpu):
    knn = gpu.KnnQuery()
    a = gpu.Matrix(np.zeros((4, 3), dtype=np.float32))
    b = gpu.Matrix(np.zeros((2, 5), dtype=np.float32))
    with pytest.raises(ValueError):
        knn.topk(a, b, 2)
    with pytest.raises(ValueError):
        knn.topk(a, gpu.Matrix(np.zeros((2, 3), dtype=np.float16)), 2)


def test_topk_fp16_factors(gpu):
    rng = np.random.default_rng(2)
    items = rng.random((500, 32), dtype=np.float32).astype(np.float16)
    q = rng.random((7, 32), dtype=np.float32).astype(np.float16)
    ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), 5)
    scores = q.astype(np.float32) @ items.astype(np.float32).T
    want = np.flip(np.argsort(scores)[:, -5:], axis=1)
    assert d.dtype == np.float32
    assert_array_equal(ids, want)
