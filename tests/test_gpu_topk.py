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


def test_topk_ties_match_select_h_exactly(gpu, oracle):
    """Exact ties: the retained set follows the reference heap's arrival-order rule (select.h:12-40,
    SURVEY App. A.4) bit for bit -- including the golden tie rows produced by the compiled reference."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "als_golden.npz"))
    knn = gpu.KnnQuery()
    one = gpu.Matrix(np.ones((1, 1), dtype=np.float32))
    for i in range(int(g["n_ties"])):
        row = g[f"tie{i}_row"]
        for k in (2, 3, 5):
            ids, dist = knn.topk(gpu.Matrix(row.reshape(-1, 1).copy()), one, k)
            assert_array_equal(ids, g[f"tie{i}_k{k}_ids"], err_msg=f"row {row} k={k}")
            assert_array_equal(dist, g[f"tie{i}_k{k}_dist"])
    # heavy ties: small-integer scores, every k, against the oracle's heap
    rng = np.random.default_rng(4)
    items = rng.integers(0, 4, size=(700, 1)).astype(np.float32)
    items[::37] = -0.0
    queries = np.array([[1.0], [0.0], [-1.0], [2.0]], dtype=np.float32)
    for k in (1, 7, 64, 300, 700):
        want_ids, want_d = oracle.topk(items, queries, k)
        ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(queries), k)
        assert_array_equal(ids, want_ids, err_msg=f"k={k}")
        assert_array_equal(d, want_d)


@pytest.mark.parametrize("k", [37, 99, 100])
def test_pruned_select_resolves_ties_at_the_kth_score(gpu, oracle, k):
    """Catalogues too small for the emit path (items < 512 k) take the materialising path, whose pruned select now
    applies the heap's arrival-order rule (select.h:12-40) inside its candidate list instead of sending tie rows to
    the whole-row select.  Small-integer factors: every arithmetic form gives the same exact scores, so ids AND
    distances must equal the oracle's bit for bit -- with tie groups of every size at the k-th score, and filters."""
    rng = np.random.default_rng(31)
    ni, f, nq = 20_000, 32, 96
    items = rng.integers(-7, 8, size=(ni, f)).astype(np.float32)
    items[ni // 2:] = items[: ni // 2]             # every score appears at least twice: odd k splits a pair
    queries = rng.integers(-7, 8, size=(nq, f)).astype(np.float32)
    queries[3] = 0.0                               # all scores tie
    want_ids, want_d = oracle.topk(items, queries, k)
    knn = gpu.KnnQuery()
    ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(queries), k)
    assert_array_equal(ids, want_ids)
    assert_array_equal(d, want_d)
    filt = np.unique(rng.integers(0, ni, size=3000)).astype(np.int32)
    liked = sp.random(nq, ni, density=0.003, format="csr", dtype=np.float32, random_state=5)
    want_ids, want_d = oracle.topk(items, queries, k, filter_query_items=liked, filter_items=filt)
    ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(queries), k, query_filter=gpu.COOMatrix(liked.tocoo()),
                      item_filter=gpu.IntVector(filt))
    assert_array_equal(ids, want_ids)
    assert_array_equal(d, want_d)


def test_topk_zero_queries_and_all_filtered_tail(gpu, oracle):
    """Users without interactions (zero factors) and N larger than the unfiltered items
    (recommender_base_test.py:54-57): ids identical to the CPU path, filtered entries score -FLT_MAX."""
    rng = np.random.default_rng(9)
    items = (rng.standard_normal((300, 16)) * 0.2).astype(np.float32)
    queries = (rng.standard_normal((5, 16)) * 0.2).astype(np.float32)
    queries[2] = 0
    liked = sp.random(5, 300, density=0.3, format="csr", dtype=np.float32, random_state=2)
    filt = np.arange(0, 300, 3, dtype=np.int32)
    knn = gpu.KnnQuery()
    for k in (10, 150, 300):
        want_ids, want_d = oracle.topk(items, queries, k, filter_query_items=liked, filter_items=filt)
        ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(queries), k, query_filter=gpu.COOMatrix(liked.tocoo()),
                          item_filter=gpu.IntVector(filt))
        assert_array_equal(ids, want_ids, err_msg=f"k={k}")
        assert_allclose(d, want_d, rtol=2e-5, atol=1e-7)
    assert (d[:, -1] == -np.finfo(np.float32).max).all()
    # k > items.rows: only items.rows entries are written, the rest keep the zero initialisation
    ids, d = knn.topk(gpu.Matrix(items[:6]), gpu.Matrix(queries), 8)
    want_ids, want_d = oracle.topk(items[:6], queries, 8)
    assert_array_equal(ids, want_ids)
    assert_array_equal(ids[:, 6:], 0)
    assert_array_equal(d[:, 6:], 0)


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


@pytest.mark.parametrize("ni,k,f", [(6000, 10, 64), (26_744, 100, 256), (40_000, 10, 128)])
def test_fp16_factors_are_scored_as_stored(gpu, oracle, ni, k, f):
    """fp16 item / query factors go to the scoring GEMM as they are stored (converted in registers, fp32 accumulation --
    the reference feeds fp16 operands to cublasSgemmEx, knn.cu:117-128): same ids and scores as scoring fp32 copies of the
    same values, with norms and a per-query filter; small and large catalogues (emit path and materialising path)."""
    rng = np.random.default_rng(ni + k)
    items16 = (rng.standard_normal((ni, f)) * 0.1).astype(np.float16)
    q16 = (rng.standard_normal((70, f)) * 0.1).astype(np.float16)
    items32, q32 = items16.astype(np.float32), q16.astype(np.float32)
    liked = sp.random(70, ni, density=20.0 / ni, format="csr", random_state=3, dtype=np.float32)
    knn = gpu.KnnQuery()
    norms32 = gpu.calculate_norms(gpu.Matrix(items32))
    got_ids, got_d = knn.topk(gpu.Matrix(items16), gpu.Matrix(q16), k, item_norms=norms32, query_filter=gpu.COOMatrix(liked.tocoo()))
    ref_ids, ref_d = knn.topk(gpu.Matrix(items32), gpu.Matrix(q32), k, item_norms=norms32, query_filter=gpu.COOMatrix(liked.tocoo()))
    assert_array_equal(got_ids, ref_ids)          # the same MFMA accumulations on the same values: bit for bit
    assert_array_equal(got_d, ref_d)
    want_ids, want_d = oracle.topk(items32, q32, k, item_norms=norms32.to_numpy().reshape(-1), filter_query_items=liked)
    assert_allclose(got_d, want_d, rtol=3e-5)
    # ids against the oracle: identical, except where two scores are closer than the fp32 summation-order noise (k = 100 sorted
    # scores of fp16-valued factors sit close together) -- there the id the GPU chose must HAVE that rank's score in fp64
    i64, q64, n64 = items32.astype(np.float64), q32.astype(np.float64), norms32.to_numpy().reshape(-1).astype(np.float64)
    differ = got_ids != want_ids
    assert differ.mean() < 0.05
    for r, j in zip(*np.nonzero(differ)):
        exact = i64[got_ids[r, j]] @ q64[r] / n64[got_ids[r, j]]
        assert abs(exact - want_d[r, j]) <= 4 * f * np.finfo(np.float32).eps * max(abs(want_d[r, j]), 1e-30), (r, j)


def test_emit_path_covers_small_catalogues(gpu, oracle):
    """The score-matrix-free path runs wherever the candidate lists stay sparse (stride x k survivors <= items / 64, stride
    adapted to k, >= 8): 6 000 and 26 744 items at k = 10 take it, configs[4]'s similar_items shape (26 744 items, k = 100)
    and a 5 000-item catalogue stay on the materialising path -- all against the oracle, duplicated rows (exact ties) included."""
    rng = np.random.default_rng(8)
    for ni, k, f in ((26_744, 100, 256), (26_744, 10, 256), (6_000, 10, 64), (5_000, 100, 64)):
        items = (rng.standard_normal((ni, f)) * 0.1).astype(np.float32)
        items[::11] = items[5]                      # duplicated rows: exact ties, some at the k-th score
        q = items[rng.choice(ni, 130, replace=False)]
        norms = gpu.calculate_norms(gpu.Matrix(items))
        ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), k, item_norms=norms)
        nh = norms.to_numpy().reshape(-1)
        want_ids, want_d = oracle.topk(items, q, k, item_norms=nh)
        assert_allclose(d, want_d, rtol=3e-5)
        # ids identical to the oracle's -- exact ties (the duplicated rows) follow the reference heap's arrival-order rule --
        # except where two DIFFERENT scores are closer than the fp32 summation noise: there the id the GPU put at a rank
        # must have that rank's score in fp64
        differ = ids != want_ids
        assert differ.mean() < 0.05
        i64, q64 = items.astype(np.float64), q.astype(np.float64)
        for r, j in zip(*np.nonzero(differ)):
            exact = i64[ids[r, j]] @ q64[r] / nh[ids[r, j]]
            assert abs(exact - want_d[r, j]) <= 4 * f * np.finfo(np.float32).eps * max(abs(want_d[r, j]), 1e-30), (ni, k, r, j)


@pytest.mark.parametrize("f", [100, 10, 50])
def test_factor_counts_off_the_16_grid_take_the_fast_path(gpu, oracle, f):
    """f = 100 (the reference's CPU default) and other counts that are not a multiple of 16: rows are zero-padded to the
    next multiple (extra zero factors change no dot product) and scored by the direct-operand kernels -- fp32 and fp16
    storage, norms and a per-query filter, against the oracle on the unpadded factors."""
    rng = np.random.default_rng(f)
    ni, k = 40_000, 10
    items = (rng.standard_normal((ni, f)) * 0.1).astype(np.float32)
    q = (rng.standard_normal((90, f)) * 0.1).astype(np.float32)
    liked = sp.random(90, ni, density=15.0 / ni, format="csr", random_state=5, dtype=np.float32)
    norms = gpu.calculate_norms(gpu.Matrix(items))
    nh = norms.to_numpy().reshape(-1)
    ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), k, item_norms=norms, query_filter=gpu.COOMatrix(liked.tocoo()))
    want_ids, want_d = oracle.topk(items, q, k, item_norms=nh, filter_query_items=liked)
    assert_allclose(d, want_d, rtol=3e-5)
    ok = ~_near_tie_rows(want_d, f)
    assert ok.mean() > 0.9
    assert_array_equal(ids[ok], want_ids[ok])
    i16, q16 = items.astype(np.float16), q.astype(np.float16)
    ids16, d16 = gpu.KnnQuery().topk(gpu.Matrix(i16), gpu.Matrix(q16), k)
    w_ids, w_d = oracle.topk(i16.astype(np.float32), q16.astype(np.float32), k)
    assert_allclose(d16, w_d, rtol=3e-5)
    ok = ~_near_tie_rows(w_d, f)
    assert_array_equal(ids16[ok], w_ids[ok])


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_emit_path_over_several_query_batches(gpu, oracle, dtype):
    """More queries than one emit batch (2 048 rows): the fragment-ordered split copy of the query rows is indexed per batch and
    padded to whole 128-row blocks, the last batch is ragged (77 rows) -- fp32 and fp16 storage, per-query filter, against the
    oracle on the same (rounded) factors."""
    rng = np.random.default_rng(21)
    ni, nq, f, k = 8_000, 2 * 2048 + 77, 64, 10
    items = (rng.standard_normal((ni, f)) * 0.1).astype(dtype)
    q = (rng.standard_normal((nq, f)) * 0.1).astype(dtype)
    liked = sp.random(nq, ni, density=12.0 / ni, format="csr", random_state=3, dtype=np.float32)
    ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), k, query_filter=gpu.COOMatrix(liked.tocoo()))
    want_ids, want_d = oracle.topk(items.astype(np.float32), q.astype(np.float32), k, filter_query_items=liked)
    assert_allclose(d, want_d, rtol=3e-5, atol=1e-7)
    ok = ~_near_tie_rows(want_d, f)
    assert ok.mean() > 0.9
    assert_array_equal(ids[ok], want_ids[ok])


@pytest.mark.parametrize("item_scale,query_scale", [(1.0, 1.0), (1e-4, 30.0), (2e3, 1e-3), (1e-6, 1e-6)])
def test_fp16_form_follows_the_operands_magnitude(gpu, oracle, item_scale, query_scale):
    """The emit path scores fp32 factors as two fp16 terms per value, scaled per call by a power of two taken from the largest
    magnitudes (topk.hip, H2): trained-size, tiny and large factors keep the same relative accuracy, and rows of mixed size
    inside one matrix (every 7th item 1000 x smaller) are ranked like the oracle ranks them."""
    rng = np.random.default_rng(17)
    ni, nq, f, k = 30_000, 200, 128, 10
    items = (rng.standard_normal((ni, f)) * 0.1 * item_scale).astype(np.float32)
    items[::7] *= 1e-3
    q = (rng.standard_normal((nq, f)) * 0.1 * query_scale).astype(np.float32)
    q[::5] *= 1e-2
    ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), k)
    want_ids, want_d = oracle.topk(items, q, k + 1)
    assert_allclose(d, want_d[:, :k], rtol=3e-5, atol=0)
    ok = ~_near_tie_rows(want_d, f)
    assert ok.mean() > 0.9
    assert_array_equal(ids[ok], want_ids[ok, :k])


def test_fp16_form_with_outlier_item_rows(gpu, oracle):
    """Item rows 10^4 and 3 10^3 x the rest.  Round 5 scaled the items by a SAMPLED maximum: such rows overflowed fp16 and their
    query rows were re-scored by the six-product path.  Round 6 takes the exact maximum in the (cached) split pass
    (topk_resident.h): nothing overflows, the outliers are scored in the same form, the ordinary rows keep 22 bits below a scale
    set by the outlier.  Scores are compared with the float64 product under the fp32 dot product's own error bound (an outlier's
    score is a cancelling sum of terms 10^4 x larger than the result: rtol on the result alone is not a meaningful bar there)."""
    rng = np.random.default_rng(23)
    ni, nq, f, k = 20_000, 150, 64, 10
    items = (rng.standard_normal((ni, f)) * 0.1).astype(np.float32)
    items[7] *= 1e4
    items[4001] *= 3e3
    q = (rng.standard_normal((nq, f)) * 0.1).astype(np.float32)
    liked = sp.random(nq, ni, density=10.0 / ni, format="csr", random_state=9, dtype=np.float32)
    ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), k, query_filter=gpu.COOMatrix(liked.tocoo()))
    assert np.isfinite(d).all()
    want_ids, want_d = oracle.topk(items, q, k + 1, filter_query_items=liked)
    exact = np.einsum("qkf,qf->qk", items[ids].astype(np.float64), q.astype(np.float64))
    bound = 4 * f * np.finfo(np.float32).eps * np.einsum("qkf,qf->qk", np.abs(items[ids]).astype(np.float64), np.abs(q).astype(np.float64))
    assert (np.abs(d - exact) <= np.maximum(bound, 3e-5 * np.abs(exact))).all()
    plain = ~np.isin(ids, (7, 4001))
    assert_allclose(d[plain], want_d[:, :k][plain], rtol=3e-5, atol=1e-7)
    ok = ~_near_tie_rows(want_d, f)
    assert_array_equal(ids[ok], want_ids[ok, :k])
    assert (ids[:, 0] == 7).sum() > nq // 4      # the outlier leads wherever its score is positive
    # the same handle again: the item planes are cached -- same answer, bit for bit
    knn = gpu.KnnQuery()
    I, Q = gpu.Matrix(items), gpu.Matrix(q)
    first = knn.topk(I, Q, k)
    second = knn.topk(I, Q, k)
    assert_array_equal(first[0], second[0])
    assert_array_equal(first[1], second[1])


def test_item_plane_cache_follows_writes(gpu, oracle):
    """The fragment-ordered fp16 planes of the item matrix live in the KnnQuery handle across calls (topk_resident.h).  Every write
    to the matrix through the library must invalidate them: a host upload, assign_rows, a solver sweep over the rows.  A matrix
    whose device address was handed out is never cached from."""
    rng = np.random.default_rng(41)
    ni, nq, f, k = 16_000, 64, 64, 10
    a = (rng.standard_normal((ni, f)) * 0.1).astype(np.float32)
    b = (rng.standard_normal((ni, f)) * 0.1).astype(np.float32)
    q = (rng.standard_normal((nq, f)) * 0.1).astype(np.float32)
    from implicit_amd.synthetic import synthetic_csr

    def same(got, want):  # a stale plane set would give unrelated ids; fp32 near-ties may move a position or two
        return (got == want).mean() > 0.99

    knn, I, Q = gpu.KnnQuery(), gpu.Matrix(a), gpu.Matrix(q)
    ids_a, _ = knn.topk(I, Q, k)
    assert same(ids_a, oracle.topk(a, q, k)[0])
    assert_array_equal(knn.topk(I, Q, k)[0], ids_a)                      # cached planes
    I.copy_from_numpy(b)                                                  # same address, new contents
    ids_b, _ = knn.topk(I, Q, k)
    assert same(ids_b, oracle.topk(b, q, k)[0])
    rows = np.arange(0, ni, 3, dtype=np.int32)
    I.assign_rows(rows, gpu.Matrix(a[rows]))                              # partial overwrite
    mixed = b.copy()
    mixed[rows] = a[rows]
    assert same(knn.topk(I, Q, k)[0], oracle.topk(mixed, q, k)[0])
    # a CG sweep rewrites the item factors in place (what fit() does between recommend() calls)
    C = synthetic_csr(ni, nq, 40_000, seed=4)
    gram = gpu.Matrix.zeros(f, f)
    solver = gpu.LeastSquaresSolver()
    solver.calculate_yty(Q, gram, 0.05)
    solver.least_squares(gpu.CSRMatrix(C), I, gram, Q, 3)
    now = I.to_numpy()
    assert same(knn.topk(I, Q, k)[0], oracle.topk(now, q, k)[0])
    # an exposed address: nothing is kept (the planes are remade per call), answers stay right after an untracked change is
    # simulated by a tracked one
    _ = I.device_ptr
    assert same(knn.topk(I, Q, k)[0], oracle.topk(now, q, k)[0])
    I.copy_from_numpy(a)
    assert_array_equal(knn.topk(I, Q, k)[0], ids_a)


def test_fp16_form_on_heavy_tailed_item_norms(gpu, oracle):
    """Item rows whose magnitudes spread over four decades (log-uniform row scales): the sampled maximum is within the fp16 form's
    headroom of the true one or it is not -- either way the answer is the oracle's."""
    rng = np.random.default_rng(29)
    ni, nq, f, k = 24_000, 120, 64, 10
    items = (rng.standard_normal((ni, f)) * 0.1).astype(np.float32) * (10.0 ** rng.uniform(-2, 2, (ni, 1))).astype(np.float32)
    q = (rng.standard_normal((nq, f)) * 0.1).astype(np.float32)
    ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), k)
    want_ids, want_d = oracle.topk(items, q, k + 1)
    assert_allclose(d, want_d[:, :k], rtol=3e-5, atol=0)
    ok = ~_near_tie_rows(want_d, f)
    assert ok.mean() > 0.9
    assert_array_equal(ids[ok], want_ids[ok, :k])


@pytest.mark.parametrize("shape", [(20_000, 64, 300, 10), (3_000, 40, 70, 100), (40_000, 128, 1500, 10)])
def test_device_output_pointers(gpu, shape):
    """imp_knn_topk writes into DEVICE buffers when the caller passes device pointers (the reference detects where its output
    pointers live, knn.cu:40-54,147-164): every path -- emit (sparse candidate lists), materialising (k = 100 over a small
    catalogue), several query batches, with the liked-items filter -- must return what the host-pointer call returns."""
    ni, f, nq, k = shape
    rng = np.random.default_rng(ni)
    items = rng.standard_normal((ni, f)).astype(np.float32)
    queries = rng.standard_normal((nq, f)).astype(np.float32)
    liked = sp.random(nq, ni, density=20.0 / ni, format="csr", random_state=3, dtype=np.float32)
    knn = gpu.KnnQuery(max_temp_memory=50_000_000 if nq > 1000 else 0)
    I, Q = gpu.Matrix(items), gpu.Matrix(queries)
    for flt in (None, gpu.COOMatrix(liked.tocoo()), gpu.COOMatrix.from_csr_pattern(liked)):
        want_ids, want_d = knn.topk(I, Q, k, query_filter=flt)
        ids_d, dist_d = knn.topk_device(I, Q, k, query_filter=flt)
        assert_array_equal(ids_d.to_numpy().view(np.int32), want_ids)
        assert_array_equal(dist_d.to_numpy(), want_d)
        if flt is not None:  # nothing a user already liked comes back
            hit = [np.intersect1d(want_ids[r], liked.indices[liked.indptr[r]:liked.indptr[r + 1]]).size for r in range(nq)]
            assert not any(hit)


def test_coo_filter_from_a_csr_pattern(gpu):
    """COOMatrix.from_csr_pattern (what recommend() builds per batch: indptr + indices through page-locked staging, row ids expanded
    on the device) filters exactly like the COO built from scipy's tocoo() (knn.cu:197-214 reads row / col only) -- 32- and
    64-bit offsets, empty rows, one long row, an empty matrix; a decreasing indptr is a ValueError."""
    rng = np.random.default_rng(5)
    ni, f, nq, k = 30_000, 32, 300, 10
    items = (rng.standard_normal((ni, f)) * 0.1).astype(np.float32)
    queries = (rng.standard_normal((nq, f)) * 0.1).astype(np.float32)
    liked = sp.random(nq, ni, density=0.003, format="lil", dtype=np.float32, random_state=9)
    liked[7] = 0                                     # an empty row
    liked[11, : ni // 3] = 1.0                       # a long one: 10 000 liked items
    liked = liked.tocsr()
    liked.eliminate_zeros()
    I, Q, knn = gpu.Matrix(items), gpu.Matrix(queries), gpu.KnnQuery()
    want_ids, want_d = knn.topk(I, Q, k, query_filter=gpu.COOMatrix(liked.tocoo()))
    for dtype in (np.int32, np.int64):
        L = liked.copy()
        L.indptr = L.indptr.astype(dtype)
        for _ in range(2):                           # the second call re-uses the staging buffer
            ids, d = knn.topk(I, Q, k, query_filter=gpu.COOMatrix.from_csr_pattern(L))
            assert_array_equal(ids, want_ids)
            assert_array_equal(d, want_d)
    assert not np.isin(want_ids[11], np.arange(ni // 3)).any()
    empty = sp.csr_matrix((nq, ni), dtype=np.float32)
    ids, _ = knn.topk(I, Q, k, query_filter=gpu.COOMatrix.from_csr_pattern(empty))
    assert_array_equal(ids, knn.topk(I, Q, k)[0])
    bad = liked.copy()
    bad.indptr = bad.indptr.copy()
    bad.indptr[5] = bad.indptr[6] + 3
    with pytest.raises(ValueError):
        gpu.COOMatrix.from_csr_pattern(bad)


def test_screened_emit_pass_bounds_hold_on_awkward_rows(gpu, oracle):
    """The emit pass scores with ONE fp16 product and keeps everything within a rigorous error bound of the threshold; the
    select kernel re-scores what can still be among the best k in fp32 (topk_resident.h MODE 3).  Rows that stress the bound:
    queries of very different magnitude in one batch, a query that is tiny against its own rounding (all scores within the
    bound: hundreds of entries re-scored, or the row handed to the exact path), items with outlier norms that inflate the
    catalogue-wide bound, duplicated items (exact ties after re-scoring), fp16-stored factors.  Ids must be the oracle's outside
    fp32 near-ties, distances within the reference's own tolerance, tie rows bit for bit."""
    rng = np.random.default_rng(12)
    ni, f, nq, k = 40_000, 128, 256, 10
    items = (rng.standard_normal((ni, f)) * 0.05).astype(np.float32)
    items[::997] *= 40.0                               # outlier norms: N_max is 40 x the typical norm
    items[1::2000] = items[3]                          # duplicated rows: exact ties
    queries = (rng.standard_normal((nq, f)) * 0.1).astype(np.float32)
    queries[::5] *= 1e-4
    queries[1::5] *= 300.0
    queries[7] = 1e-30                                 # scores far below the bound's absolute terms
    queries[9] = items[3] * 2                          # the duplicated item wins: ties at the top
    queries[11] = 0.0                                  # all scores tie at zero
    want_ids, want_d = oracle.topk(items, queries, k + 1)
    knn = gpu.KnnQuery()
    ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(queries), k)
    audit = _near_tie_rows(want_d, f)
    ties = [9, 11]                                     # exact ties: the heap's rule depends on k -- compared with the oracle at k
    audit[ties] = True
    ok = ~audit
    assert_array_equal(ids[ok], want_ids[ok, :k])
    assert_allclose(d[ok], want_d[ok, :k], rtol=3e-5, atol=1e-30)
    tie_ids, tie_d = oracle.topk(items, queries[ties], k)
    assert_array_equal(ids[ties], tie_ids)
    assert_allclose(d[ties], tie_d, rtol=3e-5)
    for r in np.nonzero(audit)[0]:
        if r not in ties:
            assert len(set(ids[r]) ^ set(want_ids[r, :k])) <= 2
    # fp16-stored factors are re-scored as stored
    ih, qh = items.astype(np.float16), (queries[:64] * 0 + rng.standard_normal((64, f)) * 0.1).astype(np.float16)
    want_ids, want_d = oracle.topk(ih.astype(np.float32), qh.astype(np.float32), k + 1)
    ids, d = knn.topk(gpu.Matrix(ih), gpu.Matrix(qh), k)
    ok = ~_near_tie_rows(want_d, f)
    assert_array_equal(ids[ok], want_ids[ok, :k])
    assert_allclose(d[ok], want_d[ok, :k], rtol=2e-3)


def test_results_do_not_depend_on_the_adaptive_subset_stride(gpu, oracle):
    """The screened path widens the threshold pre-pass's subset stride (32 -> 64 -> 128) when a handle's candidate lists come
    out short -- heavy-tailed item norms, as trained factors have -- and narrows it again when they do not.  The ids and scores
    of every call are the oracle's whatever the stride was; a different catalogue size starts over."""
    rng = np.random.default_rng(21)
    ni, f, nq, k = 150_000, 64, 512, 10
    items = (rng.standard_normal((ni, f)) * 0.05).astype(np.float32) * rng.lognormal(0.0, 1.2, size=(ni, 1)).astype(np.float32)
    queries = (rng.standard_normal((nq, f)) * 0.1).astype(np.float32)
    want_ids, want_d = oracle.topk(items, queries, k + 1)
    ok = ~_near_tie_rows(want_d, f)
    knn, I, Q = gpu.KnnQuery(), gpu.Matrix(items), gpu.Matrix(queries)
    for call in range(5):                               # stride 32, 64, 128, 128, 128 if the lists stay short
        ids, d = knn.topk(I, Q, k)
        assert_array_equal(ids[ok], want_ids[ok, :k], err_msg=f"call {call}")
        assert_allclose(d[ok], want_d[ok, :k], rtol=3e-5)
    flat = (rng.standard_normal((60_000, f)) * 0.1).astype(np.float32)   # unstructured: long lists, the stride goes back / stays
    w_ids, w_d = oracle.topk(flat, queries, k + 1)
    ok2 = ~_near_tie_rows(w_d, f)
    F = gpu.Matrix(flat)
    for call in range(3):
        ids, d = knn.topk(F, Q, k)
        assert_array_equal(ids[ok2], w_ids[ok2, :k], err_msg=f"flat call {call}")


def test_concentrated_scores_defeat_the_screen_not_the_result(gpu):
    """All-positive, nearly parallel factors (a model one sweep from the default start): every score within 1e-3 of the next, so
    the one-product screen passes thousands of entries per row and the lists overflow.  The batch is then redone by the
    three-product emit pass -- decided by the batch's own outcome, so the same call returns the same bits every time -- and the
    answer is still the true top k (judged in float64: with scores this close, adjacent ranks are near-ties by construction)."""
    rng = np.random.default_rng(2)
    ni, f, nq, k = 120_000, 64, 300, 10
    items = (rng.random((ni, f), dtype=np.float32) * 0.01 + 0.02).astype(np.float32)
    queries = (rng.random((nq, f), dtype=np.float32) * 0.01 + 0.02).astype(np.float32)
    knn, I, Q = gpu.KnnQuery(), gpu.Matrix(items), gpu.Matrix(queries)
    ids, d = knn.topk(I, Q, k)
    ids2, d2 = knn.topk(I, Q, k)
    assert_array_equal(ids, ids2)
    assert_array_equal(d, d2)
    exact = queries.astype(np.float64) @ items.astype(np.float64).T
    best = -np.sort(-exact, axis=1)[:, :k]
    got = np.take_along_axis(exact, ids.astype(np.int64), axis=1)
    assert_allclose(got, best, rtol=2e-6)               # the returned items ARE the best k up to fp32 near-ties
    assert_allclose(d, got, rtol=3e-5)
    assert (np.diff(d, axis=1) <= 0).all()
