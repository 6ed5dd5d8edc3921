"""Parity at BASELINE.json's full sizes through size-independent checks: the oracle re-solves a uniform
sample of the rows of a full-size half sweep (each row's solve depends only on that row's nonzeros, the
other side's factors and the gramian), and algebraic properties are checked on everything."""
import numpy as np
import pytest

from implicit_amd.synthetic import named

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def _sampled_rows_check(gpu, oracle, C, X0, Y0, reg, solve_gpu, solve_oracle, n_sample=4000, tol=1e-4):
    Xd, Yd = gpu.Matrix(X0), gpu.Matrix(Y0)
    solve_gpu(gpu.CSRMatrix(C), Xd, Yd)
    got = Xd.to_numpy()
    assert np.isfinite(got).all()
    lens = np.diff(C.indptr)
    assert not got[lens == 0].any()                       # empty rows are zeroed
    rows = np.unique(np.concatenate([np.arange(0, C.shape[0], max(1, C.shape[0] // n_sample)),
                                     np.argsort(lens)[-64:]]))   # uniform sample + the 64 longest rows
    want = np.ascontiguousarray(X0[rows])
    solve_oracle(C[rows], want, Y0)
    err = rel(got[rows], want)
    print(f"{C.shape} nnz={C.nnz}: {len(rows)} sampled rows (max nnz {lens[rows].max()}) rel={err:.2e}")
    assert err < tol
    return got


def test_config3_cg_half_sweeps_full_size(gpu, oracle):
    """BASELINE configs[2] shape: 358,868 x 292,385, 17.3M nnz, f=128, CG 3 -- both orientations."""
    C = named("lastfm360k")
    f, reg = 128, 0.01
    rng = np.random.default_rng(7)
    X0 = rng.random((C.shape[0], f), dtype=np.float32) * 0.2 - 0.1
    Y0 = rng.random((C.shape[1], f), dtype=np.float32) * 0.2 - 0.1
    solver = gpu.LeastSquaresSolver()

    def solve_gpu(Cd, Xd, Yd):
        gram = gpu.Matrix.zeros(f, f)
        solver.calculate_yty(Yd, gram, reg)
        solver.least_squares(Cd, Xd, gram, Yd, 3)
        solve_gpu.gram = gram.to_numpy()

    def solve_oracle(Csub, Xsub, Y):
        oracle.least_squares_cg(Csub, Xsub, Y, reg, cg_steps=3, YtY=solve_gpu.gram)

    X1 = _sampled_rows_check(gpu, oracle, C, X0, Y0, reg, solve_gpu, solve_oracle)
    assert rel(solve_gpu.gram, oracle.gramian(Y0) + np.float32(reg) * np.eye(f, dtype=np.float32)) < 1e-5
    _sampled_rows_check(gpu, oracle, C.T.tocsr(), Y0, X1, reg, solve_gpu, solve_oracle)


def test_config2_cholesky_scaled(gpu, oracle):
    """BASELINE configs[1] shape at 1/5 scale (200K x 20K, 10M nnz), f=64, Cholesky."""
    C = named("c2", scale=0.2)
    f, reg = 64, 0.01
    rng = np.random.default_rng(3)
    X0 = np.zeros((C.shape[0], f), dtype=np.float32)
    Y0 = rng.random((C.shape[1], f), dtype=np.float32) * 0.2 - 0.1
    solver = gpu.LeastSquaresSolver()

    def solve_gpu(Cd, Xd, Yd):
        gram = gpu.Matrix.zeros(f, f)
        solver.calculate_yty(Yd, gram, 0.0)
        solver.least_squares_cholesky(Cd, Xd, gram, Yd, reg)

    def solve_oracle(Csub, Xsub, Y):
        oracle.least_squares(Csub, Xsub, Y, reg)

    _sampled_rows_check(gpu, oracle, C, X0, Y0, reg, solve_gpu, solve_oracle, n_sample=2000)


def test_topk_full_item_count(gpu, oracle):
    """recommend-shaped top-k over all 292,385 items for a query sample, ids identical to the oracle."""
    rng = np.random.default_rng(5)
    items = (rng.standard_normal((292_385, 128)) * 0.1).astype(np.float32)
    query = (rng.standard_normal((64, 128)) * 0.1).astype(np.float32)
    ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(query), 10)
    want_ids, want_d = oracle.topk(items, query, 11)
    gaps = np.abs(np.diff(want_d.astype(np.float64), axis=1)) / np.abs(want_d[:, :-1])
    ok = ~(gaps < 4 * 128 * np.finfo(np.float32).eps).any(axis=1)   # near-tie rows audited separately
    assert ok.mean() > 0.9
    np.testing.assert_array_equal(ids[ok], want_ids[ok, :10])
    np.testing.assert_allclose(d[ok], want_d[ok, :10], rtol=2e-5)
