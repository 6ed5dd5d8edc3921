"""Pins the CPU oracle (oracle/als_oracle.c) to the reference:
 (a) against golden vectors produced by the compiled reference itself (tests/golden/als_golden.npz,
     generator tests/golden/make_golden.py) -- runs everywhere;
 (b) against the compiled reference modules in oracle/_ref on fresh seeded inputs -- runs where they
     have been built (this container; they also travel to the GPU box).
Tolerances are fp32 summation-order noise: the reference reaches OpenBLAS, the oracle plain loops.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from numpy.testing import assert_allclose, assert_array_equal

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "als_golden.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def csr(g, prefix):
    shape = tuple(g[prefix + "_shape"])
    return sp.csr_matrix((g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]), shape=shape)


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def test_golden_cg_and_cholesky(g, oracle):
    for name in g["als_cases"]:
        C = csr(g, f"{name}_C")
        X0, Y0 = g[f"{name}_X0"], g[f"{name}_Y0"]
        for steps in (1, 3):
            X = X0.copy()
            oracle.least_squares_cg(C, X, Y0, 0.05, cg_steps=steps)
            assert rel(X, g[f"{name}_cg{steps}_X"]) < 2e-5, (name, steps)
        Y = Y0.copy()
        oracle.least_squares_cg(C.T.tocsr(), Y, g[f"{name}_cg3_X"], 0.05)
        assert rel(Y, g[f"{name}_cg3_Y"]) < 2e-5, name
        X64 = oracle.least_squares_cg_f64(C, X0, Y0, 0.05)
        assert rel(X64, g[f"{name}_cg3_X_f64"]) < 1e-6, name   # fp32 gramian inputs differ in the last bit only
        X = X0.copy()
        oracle.least_squares(C, X, Y0, 0.05)
        assert rel(X, g[f"{name}_chol_X"]) < 2e-5, name
        empty = np.diff(C.indptr) == 0
        assert empty.any() and not X[empty].any()
        for reg, want in zip((0.0, 0.05, 10.0), g[f"{name}_loss"]):
            assert oracle.calculate_loss(C, g[f"{name}_cg3_X"], Y0, reg) == pytest.approx(want, rel=1e-5)


def test_golden_factorize_known_answer(g, oracle):
    """tests/als_test.py:142-186 of the reference: 15 iterations reconstruct the 7x6 matrix to 1e-3."""
    counts = csr(g, "factorize_counts")
    dense = counts.toarray()
    for solver, use_cg in (("cg", True), ("chol", False)):
        X, Y = oracle.fit(counts, 6, regularization=0.0, alpha=2.0, iterations=15, use_cg=use_cg, random_state=42,
                          num_threads=1)
        assert np.abs(X @ Y.T - dense).max() < 1e-3
        ref_rec = g[f"factorize_{solver}_X"] @ g[f"factorize_{solver}_Y"].T
        assert np.abs(ref_rec - dense).max() < 1e-3
        assert_allclose(X @ Y.T, ref_rec, atol=2e-3)


def test_golden_topk(g, oracle):
    items, query = g["topk_items"], g["topk_query"]
    liked, filt, norms = csr(g, "topk_liked"), g["topk_filter_items"], g["topk_norms"]
    variants = {"plain": {}, "norms": {"item_norms": norms},
                "filters": {"filter_query_items": liked, "filter_items": filt},
                "all": {"item_norms": norms, "filter_query_items": liked, "filter_items": filt}}
    for tag, kw in variants.items():
        for k in (1, 10, 64):
            ids, dist = oracle.topk(items, query, k, **kw)
            assert_array_equal(ids, g[f"topk_{tag}_k{k}_ids"])
            assert_allclose(dist, g[f"topk_{tag}_k{k}_dist"], rtol=2e-5, atol=1e-7)


def test_golden_select_tie_semantics(g, oracle):
    """select.h:12-40 arrival-order tie behaviour, bit for bit."""
    for i in range(int(g["n_ties"])):
        row = g[f"tie{i}_row"]
        for k in (2, 3, 5):
            ids, dist = oracle.topk(row.reshape(-1, 1), np.ones((1, 1), dtype=np.float32), k)
            assert_array_equal(ids, g[f"tie{i}_k{k}_ids"], err_msg=f"row {row} k={k}")
            assert_array_equal(dist, g[f"tie{i}_k{k}_dist"])
    ids, _ = oracle.select(np.array([[5, 5, 5, 9]], dtype=np.float32), 2)
    assert_array_equal(ids, [[3, 1]])


def test_loss_known_answers(oracle):
    """tests/als_test.py:304-324 of the reference."""
    ratings = sp.coo_matrix(([1.0], ([0], [0])), shape=(1, 2)).tocsr().astype(np.float32)
    item_factors = np.array([[0.0], [1.0]], dtype="float32")
    user_factors = np.array([[1.0]], dtype="float32")
    assert oracle.calculate_loss(ratings, user_factors, item_factors, 0) == pytest.approx(1.0)
    assert oracle.calculate_loss(ratings, user_factors, item_factors, 1.0) == pytest.approx(2.0)


def test_cholesky_failure_reports_row(oracle):
    C = sp.csr_matrix(np.ones((3, 4), dtype=np.float32))
    with pytest.raises(ValueError, match="posv failed"):
        oracle.least_squares(C, np.zeros((3, 8), dtype=np.float32), np.zeros((4, 8), dtype=np.float32), 0.0)


# ---- (b) live comparison with the compiled reference ------------------------------------------------
def _ref():
    from oracle import ref

    return ref.load()


needs_ref = pytest.mark.skipif(_ref()[0] is None, reason="oracle/_ref not built (python oracle/build_ref.py)")


@needs_ref
@pytest.mark.parametrize("f", [16, 64, 100, 128])
def test_oracle_vs_compiled_reference(oracle, f):
    from implicit_amd.synthetic import synthetic_csr

    als, topk = _ref()
    C = synthetic_csr(4000, 1500, 120_000, seed=f, neg_frac=0.05, empty_frac=0.01)
    Ct = C.T.tocsr()
    rng = np.random.default_rng(f)
    X = rng.random((4000, f), dtype=np.float32) * 0.01
    Y = rng.random((1500, f), dtype=np.float32) * 0.01
    # lockstep: the oracle sweep is compared with the reference's from the reference's state
    for it in range(3):
        for M, A, B in ((C, X, Y), (Ct, Y, X)):
            mine = A.copy()
            oracle.least_squares_cg(M, mine, B, 0.01)
            als.least_squares_cg(M, A, B, 0.01, num_threads=4, cg_steps=3)
            tol = 2e-3 if it == 0 else 5e-5  # cold first iteration: ill-conditioned, see SURVEY App. A.5
            assert rel(mine, A) < tol, (f, it)
    mine, theirs = X.copy(), X.copy()
    oracle.least_squares(C, mine, Y, 0.01)
    als.least_squares(C, theirs, Y, 0.01, num_threads=4)
    assert rel(mine, theirs) < 5e-5
    assert oracle.calculate_loss(C, X, Y, 0.01) == pytest.approx(als.calculate_loss(C, X, Y, 0.01), rel=1e-5)
    q = X[:50]
    ids, dist = oracle.topk(Y, q, 10)
    rids, rdist = topk.topk(Y, q, 10, num_threads=1)
    same = (ids == rids).all(axis=1)
    assert same.mean() > 0.97  # the rest are fp32 near-ties between the two GEMM orders
    assert_allclose(dist[same], rdist[same], rtol=2e-5, atol=1e-9)
