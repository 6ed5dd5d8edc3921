"""The golden vectors produced by the compiled reference (tests/golden/als_golden.npz, generator
tests/golden/make_golden.py) replayed on the HIP kernels through the C-ABI: the GPU is compared with the
REFERENCE's own outputs, not only with the restatement."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from numpy.testing import assert_allclose, assert_array_equal

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "als_golden.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def csr(g, prefix):
    shape = tuple(g[prefix + "_shape"])
    return sp.csr_matrix((g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]), shape=shape)


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def test_golden_solvers_and_loss(gpu, g):
    solver = gpu.LeastSquaresSolver()
    for name in g["als_cases"]:
        C = csr(g, f"{name}_C")
        X0, Y0 = g[f"{name}_X0"], g[f"{name}_Y0"]
        f = X0.shape[1]
        Yd = gpu.Matrix(Y0)
        gram = gpu.Matrix.zeros(f, f)
        solver.calculate_yty(Yd, gram, 0.05)
        Cd = gpu.CSRMatrix(C)
        for steps in (1, 3):
            Xd = gpu.Matrix(X0)
            solver.least_squares(Cd, Xd, gram, Yd, steps)
            err = rel(Xd.to_numpy(), g[f"{name}_cg{steps}_X"])
            assert err < 1e-4, (name, steps, err)
        # warm item sweep from the reference's user factors
        Xref = gpu.Matrix(g[f"{name}_cg3_X"])
        solver.calculate_yty(Xref, gram, 0.05)
        Yw = gpu.Matrix(Y0)
        solver.least_squares(gpu.CSRMatrix(C.T.tocsr()), Yw, gram, Xref, 3)
        assert rel(Yw.to_numpy(), g[f"{name}_cg3_Y"]) < 1e-4, name
        # Cholesky (unregularised gramian in, reg added inside)
        solver.calculate_yty(Yd, gram, 0.0)
        Xc = gpu.Matrix(np.zeros_like(X0))
        solver.least_squares_cholesky(Cd, Xc, gram, Yd, 0.05)
        assert rel(Xc.to_numpy(), g[f"{name}_chol_X"]) < 1e-4, name
        for reg, want in zip((0.0, 0.05, 10.0), g[f"{name}_loss"]):
            got = solver.calculate_loss(Cd, gpu.Matrix(g[f"{name}_cg3_X"]), Yd, reg)
            assert got == pytest.approx(want, rel=1e-4), (name, reg)


def test_golden_topk(gpu, g):
    items, query = g["topk_items"], g["topk_query"]
    liked, filt, norms = csr(g, "topk_liked"), g["topk_filter_items"], g["topk_norms"]
    knn = gpu.KnnQuery()
    I, Q = gpu.Matrix(items), gpu.Matrix(query)
    N = gpu.Matrix(norms.reshape(1, -1))
    coo = gpu.COOMatrix(liked.tocoo())
    fv = gpu.IntVector(filt)
    variants = {"plain": {}, "norms": {"item_norms": N}, "filters": {"query_filter": coo, "item_filter": fv},
                "all": {"item_norms": N, "query_filter": coo, "item_filter": fv}}
    for tag, kw in variants.items():
        for k in (1, 10, 64):
            ids, dist = knn.topk(I, Q, k, **kw)
            assert_array_equal(ids, g[f"topk_{tag}_k{k}_ids"], err_msg=f"{tag} k={k}")
            assert_allclose(dist, g[f"topk_{tag}_k{k}_dist"], rtol=2e-5, atol=1e-7)
