"""GPU parity of the resident lock-step kernels of the f = 256 CG path (implicit_amd/csrc/als_cg_w256.hip): rows of up to 256
nonzeros keep their gathered factor rows in registers, R = 16 / WPR rows per workgroup share one matrix-core product with the
fp16-split gramian per pass.

Reference semantics: implicit/gpu/als.cu:23-111 == implicit/cpu/_als.pyx:152-248 (the oracle restates the latter).  Bar: 1e-4
relative PER ROW (a wrong team share or a wrong output tile would hide in a Frobenius norm over thousands of rows).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# every class boundary of the schedule (16 / 32 / 64 / 128 / 256 / 512) from both sides, whole and partial 4-entry steps
EDGE_LENGTHS = [1, 2, 3, 4, 5, 8, 15, 16, 17, 20, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 400, 512, 513, 1500, 0]


def _matrix(lengths, items, seed, neg_frac=0.1):
    rng = np.random.default_rng(seed)
    indptr, indices, data = [0], [], []
    for n in lengths:
        cols = np.sort(rng.choice(items, size=n, replace=False))
        c = 1 + 4 * rng.random(n)
        c[rng.random(n) < neg_frac] *= -1      # the "disliked" branch (_als.pyx:186-196)
        indices.append(cols)
        data.append(c)
        indptr.append(indptr[-1] + n)
    return sp.csr_matrix((np.concatenate(data).astype(np.float32), np.concatenate(indices).astype(np.int32), np.array(indptr)),
                         shape=(len(lengths), items))


def _row_errors(got, want):
    num = np.linalg.norm(got.astype(np.float64) - want, axis=1)
    return num / np.maximum(np.linalg.norm(want.astype(np.float64), axis=1), 1e-30)


def _solve(gpu, C, X, Y, reg, cg_steps):
    solver = gpu.LeastSquaresSolver()
    Xd, Yd = gpu.Matrix(X), gpu.Matrix(Y)
    gram = gpu.Matrix.zeros(X.shape[1], X.shape[1])
    solver.calculate_yty(Yd, gram, reg)
    solver.least_squares(gpu.CSRMatrix(C), Xd, gram, Yd, cg_steps)
    return Xd.to_numpy().astype(np.float32)


@pytest.mark.parametrize("cg_steps", [1, 3])
@pytest.mark.parametrize("f", [256, 192])
def test_every_class_boundary_row_by_row(gpu, oracle, f, cg_steps):
    items = 4000
    C = _matrix(EDGE_LENGTHS, items, seed=f + cg_steps)
    rng = np.random.default_rng(11)
    Y = ((rng.random((items, f), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    X = ((rng.random((C.shape[0], f), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.05, cg_steps=cg_steps)
    got = _solve(gpu, C, X.copy(), Y, 0.05, cg_steps)
    err = _row_errors(got, want)
    print("f", f, "cg", cg_steps, "per-row rel max %.2e" % err.max(), "at length", EDGE_LENGTHS[int(err.argmax())])
    assert err.max() < TOL
    assert np.array_equal(got[EDGE_LENGTHS.index(0)], np.zeros(f, np.float32))  # empty row -> zeros


def test_many_groups_per_class_and_ragged_ends(gpu, oracle):
    """Several lock-step groups per class, group counts that are not multiples of the rows per workgroup, rows that stop early
    (zero residual: x already solves the system is not constructible cheaply -- rows with b = 0 and x = 0 are: all-negative rows)."""
    f, items = 256, 3000
    rng = np.random.default_rng(3)
    lengths = list(rng.integers(1, 300, size=397)) + [0] * 5
    C = _matrix(lengths, items, seed=5, neg_frac=0.05)
    # ten rows with every confidence negative and a zero iterate: b = 0, r = -A x = 0 -> rsold < 1e-20, x untouched
    dead = rng.choice(397, size=10, replace=False)
    for r in dead:
        C.data[C.indptr[r]:C.indptr[r + 1]] = -np.abs(C.data[C.indptr[r]:C.indptr[r + 1]])
    Y = (rng.standard_normal((items, f)) * 0.1).astype(np.float32)
    X = (rng.standard_normal((C.shape[0], f)) * 0.1).astype(np.float32)
    X[dead] = 0
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.01, cg_steps=3)
    got = _solve(gpu, C, X.copy(), Y, 0.01, 3)
    err = _row_errors(got, want)
    live = np.setdiff1d(np.arange(len(lengths)), dead)
    print("per-row rel max %.2e" % err[live].max(), "at length", lengths[int(live[err[live].argmax()])])
    assert err[live].max() < TOL
    assert np.array_equal(got[dead], np.zeros((10, f), np.float32))


def test_gramian_and_operand_magnitudes(gpu, oracle):
    """The fp16 split scales the gramian to 2^13..2^14 and every published operand to 2^14: factors of 1e-4 and of 30 (gramians
    of 1e-5 .. 1e6), a regularisation that dwarfs the gramian, confidences of 400.  Bar: the oracle's own distance from the
    float64 answer (these systems are not all well conditioned)."""
    f, items = 256, 2500
    lengths = [5, 30, 60, 120, 250, 14, 90]
    for scale, reg, cmul in ((1e-4, 1e-6, 1.0), (30.0, 50.0, 1.0), (0.1, 1e3, 1.0), (0.1, 0.01, 100.0)):
        C = _matrix(lengths, items, seed=8, neg_frac=0.0)
        C.data[::5] *= cmul
        rng = np.random.default_rng(2)
        Y = (rng.standard_normal((items, f)) * scale).astype(np.float32)
        X = (rng.standard_normal((len(lengths), f)) * scale).astype(np.float32)
        want = X.copy()
        oracle.least_squares_cg(C, want, Y, reg, cg_steps=3)
        exact = oracle.least_squares_cg_f64(C, X, Y, reg)
        got = _solve(gpu, C, X.copy(), Y, reg, 3)
        e_gpu, e_oracle = _row_errors(got, exact), _row_errors(want, exact)
        print("scale", scale, "reg", reg, "c x", cmul, "gpu vs fp64 %.2e" % e_gpu.max(), "oracle vs fp64 %.2e" % e_oracle.max())
        assert np.isfinite(got).all()
        assert (e_gpu < np.maximum(TOL, 2.0 * e_oracle)).all()


def test_old_f256_kernel_still_agrees():
    """IMP_F256_OLD=1 (round-2 streamed kernel) and the default give the same rows within the parity bar."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r + "/tests")
import implicit_amd.gpu as gpu
from test_gpu_w256 import _matrix, _solve, EDGE_LENGTHS
C = _matrix(EDGE_LENGTHS * 3, 5000, seed=1)
rng = np.random.default_rng(0)
Y = (rng.standard_normal((5000, 256)) * 0.1).astype(np.float32)
X = (rng.standard_normal((C.shape[0], 256)) * 0.1).astype(np.float32)
np.save(sys.argv[1], _solve(gpu, C, X, Y, 0.05, 3))
""" % (ROOT, ROOT)
    import tempfile
    out = []
    with tempfile.TemporaryDirectory() as d:
        for tag, env in (("new", {}), ("old", {"IMP_F256_OLD": "1"})):
            path = os.path.join(d, tag + ".npy")
            subprocess.run([sys.executable, "-c", code, path], check=True, env={**os.environ, **env}, timeout=600)
            out.append(np.load(path))
    err = np.linalg.norm(out[0] - out[1], axis=1) / np.maximum(np.linalg.norm(out[1], axis=1), 1e-30)
    print("new vs old per-row rel max %.2e" % err.max())
    assert err.max() < TOL
    assert not np.array_equal(out[0], out[1])  # they ARE different kernels
