"""The multi-GPU driver on ONE GPU: a world-size-1 RCCL communicator runs the same code path as N ranks (partial
gramian -> all-reduce -> chunked solve with the pipelined all-gather on the exchange stream -> ordering event), so the
stream / event plumbing and the chunk bookkeeping are checked on real hardware; the N = 2 logic runs on CPU in
tests/test_sharded_gloo.py."""
import numpy as np
import pytest

from implicit_amd.synthetic import synthetic_csr

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def test_pipelined_half_sweeps_match_the_plain_solver(gpu):
    from implicit_amd.gpu import sharded

    users, items, f, reg = 3000, 2000, 64, 0.05
    C = synthetic_csr(users, items, 90_000, seed=5, neg_frac=0.05, empty_frac=0.02)
    Ct = C.T.tocsr()
    rng = np.random.default_rng(1)
    X0 = rng.random((users, f), dtype=np.float32) * 0.1 - 0.05
    Y0 = rng.random((items, f), dtype=np.float32) * 0.1 - 0.05

    # plain single-GPU iteration
    solver = gpu.LeastSquaresSolver()
    X, Y, gram = gpu.Matrix(X0), gpu.Matrix(Y0), gpu.Matrix.zeros(f, f)
    Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
    for _ in range(2):
        solver.calculate_yty(Y, gram, reg)
        solver.least_squares(Cd, X, gram, Y, 3)
        solver.calculate_yty(X, gram, reg)
        solver.least_squares(Ctd, Y, gram, X, 3)
    want_x, want_y = X.to_numpy(), Y.to_numpy()

    comm = gpu.Comm(gpu.Comm.unique_id(), 1, 0)
    backend = sharded.GpuBackend(gpu)
    u_off, i_off = np.array([0, users]), np.array([0, items])
    for chunks in (1, 3):
        Xs, Ys, g = gpu.Matrix(X0), gpu.Matrix(Y0), gpu.Matrix.zeros(f, f)
        if chunks == 1:
            Cu, Ci = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
        else:
            Cu = [gpu.CSRMatrix(c) for c in sharded.split_rows(C, chunks)]
            Ci = [gpu.CSRMatrix(c) for c in sharded.split_rows(Ct, chunks)]
        for _ in range(2):
            sharded.iteration(backend, comm, Cu, Ci, Xs, Ys, u_off, i_off, g, reg, 3)
        gpu.synchronize()
        # same kernels on the same rows; only the long-row plan may differ between a chunk and the whole matrix
        assert rel(Xs.to_numpy(), want_x) < 1e-6 and rel(Ys.to_numpy(), want_y) < 1e-6
