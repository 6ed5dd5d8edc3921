"""GPU parity of the normal-matrix kernels (implicit_amd/csrc/als_cg_nm.hip): the rows of more than 512 nonzeros of the f = 64 / 128
CG path, whose A_u = YtY + sum (c - 1) y y^T is built explicitly on the matrix cores from fp16-split operands.

Reference semantics: implicit/gpu/als.cu:23-111 == implicit/cpu/_als.pyx:152-248 (the oracle restates the latter).  Bar: 1e-4
relative, per ROW here (a wrong tile or a wrong segment sum would hide in a Frobenius norm over thousands of short rows).
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


def _long_row_matrix(lengths, items, seed, conf=lambda rng, n: 1 + 4 * rng.random(n)):
    """One row per entry of `lengths` (sorted column ids, as scipy CSR), confidences from `conf`."""
    rng = np.random.default_rng(seed)
    indptr, indices, data = [0], [], []
    for n in lengths:
        cols = np.sort(rng.choice(items, size=n, replace=False))
        indices.append(cols)
        data.append(conf(rng, n))
        indptr.append(indptr[-1] + n)
    return sp.csr_matrix((np.concatenate(data).astype(np.float32), np.concatenate(indices).astype(np.int32), np.array(indptr)),
                         shape=(len(lengths), items))


def _row_errors(got, want):
    num = np.linalg.norm(got.astype(np.float64) - want, axis=1)
    return num / np.maximum(np.linalg.norm(want.astype(np.float64), axis=1), 1e-30)


def _solve(gpu, C, X, Y, reg, cg_steps, dtype="float32"):
    solver = gpu.LeastSquaresSolver()
    if dtype == "float16":
        Xd, Yd = gpu.Matrix(X.astype(np.float16)), gpu.Matrix(Y.astype(np.float16))
    else:
        Xd, Yd = gpu.Matrix(X), gpu.Matrix(Y)
    gram = gpu.Matrix.zeros(X.shape[1], X.shape[1])
    solver.calculate_yty(Yd, gram, reg)
    solver.least_squares(gpu.CSRMatrix(C), Xd, gram, Yd, cg_steps)
    return Xd.to_numpy().astype(np.float32)


LENGTHS = [513, 514, 600, 1024, 1025, 2047, 2048, 2049, 3000, 4096, 4097, 7001, 12000, 30, 200, 0, 512]


@pytest.mark.parametrize("f", [64, 128])
@pytest.mark.parametrize("cg_steps", [1, 3])
def test_long_rows_match_the_oracle_row_by_row(gpu, oracle, f, cg_steps):
    items = 20000
    C = _long_row_matrix(LENGTHS, items, seed=f + cg_steps)
    rng = np.random.default_rng(11)
    Y = ((rng.random((items, f), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    X = ((rng.random((C.shape[0], f), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.05, cg_steps=cg_steps)
    got = _solve(gpu, C, X.copy(), Y, 0.05, cg_steps)
    err = _row_errors(got, want)
    print("f", f, "cg", cg_steps, "per-row rel max %.2e" % err.max(), "at length", LENGTHS[int(err.argmax())])
    assert err.max() < TOL
    assert np.array_equal(got[LENGTHS.index(0)], np.zeros(f, np.float32))  # empty row -> zeros (als.cu zero-fills via the solve of b = 0)


@pytest.mark.parametrize("segment", [256, 1000])
def test_rows_cut_into_many_segments(gpu, oracle, monkeypatch, segment):
    """IMP_NM_SEGMENT forces short segments: every long row goes through the partial matrices and the finishing kernel."""
    monkeypatch.setenv("IMP_NM_SEGMENT", str(segment))
    f, items = 128, 9000
    lengths = [513, 777, 1000, 1001, 2500, 6000, 300]
    C = _long_row_matrix(lengths, items, seed=3)
    rng = np.random.default_rng(5)
    Y = (rng.standard_normal((items, f)) * 0.1).astype(np.float32)
    X = (rng.standard_normal((C.shape[0], f)) * 0.1).astype(np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.01)
    got = _solve(gpu, C, X.copy(), Y, 0.01, 3)
    err = _row_errors(got, want)
    print("segment", segment, "per-row rel max %.2e" % err.max())
    assert err.max() < TOL


def test_confidence_range_and_signs(gpu, oracle):
    """Confidences of 1e5 (the weights leave the fp16 range unless the segment is rescaled), confidences below one (negative
    weights), negative confidences (weight |c| - 1, no share in b: _als.pyx:186-196) and explicit ones (weight 0)."""
    f, items = 128, 8000

    def conf(rng, n):
        c = 1 + 4 * rng.random(n)
        c[::7] = 1e5 * (1 + rng.random(len(c[::7])))
        c[1::7] = 0.25
        c[2::7] = -3.0
        c[3::7] = 1.0
        return c

    C = _long_row_matrix([600, 1500, 5000], items, seed=9, conf=conf)
    rng = np.random.default_rng(2)
    Y = (rng.standard_normal((items, f)) * 0.1).astype(np.float32)
    X = (rng.standard_normal((3, f)) * 0.1).astype(np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.01)
    exact = oracle.least_squares_cg_f64(C, X, Y, 0.01)
    got = _solve(gpu, C, X.copy(), Y, 0.01, 3)
    e_gpu, e_oracle = _row_errors(got, exact), _row_errors(want, exact)
    print("vs oracle", _row_errors(got, want), "gpu vs fp64", e_gpu, "oracle vs fp64", e_oracle)
    assert np.isfinite(got).all()
    # 1e5-weighted systems are badly conditioned: the bar is the oracle's own distance from the exact answer
    assert (e_gpu < np.maximum(TOL, 2.0 * e_oracle)).all()


def test_tiny_and_mixed_magnitudes(gpu, oracle):
    """Factors of 1e-6 .. 1 in one row: the small ones fall into the fp16 subnormals of the split, an ABSOLUTE error of 3e-8 each."""
    f, items = 64, 6000
    C = _long_row_matrix([900, 2100], items, seed=4)
    rng = np.random.default_rng(8)
    Y = (rng.standard_normal((items, f)) * np.float32(10.0) ** rng.integers(-6, 1, size=(items, 1))).astype(np.float32)
    X = np.zeros((2, f), np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.1)
    got = _solve(gpu, C, X.copy(), Y, 0.1, 3)
    err = _row_errors(got, want)
    print("per-row rel", err)
    assert err.max() < TOL


def test_operands_beyond_the_fp16_range_are_resolved_in_fp32(gpu, oracle):
    """confidences of 1e6 on factors of +-40 put the fp16-split operands (~ sqrt(2 w) |y| 2^k) past 65504.  The
    kernel must notice (the CG scalars stop being finite), store nothing, and leave the row to the fp32 fix-up kernel: finite
    output, parity with the oracle at the oracle's own fp64 distance, and a non-zero imp_solver_fixup_rows.  The rows of the same
    launch whose operands stay in range are solved by the matrix-core kernel as usual."""
    f, items = 128, 6000

    def conf(rng, n):
        return 1 + 4 * rng.random(n)

    C = _long_row_matrix([600, 900, 2500, 5000], items, seed=12, conf=conf).tolil()
    rng = np.random.default_rng(7)
    Y = (rng.standard_normal((items, f)) * 0.1).astype(np.float32)
    C = C.tocsr()
    # rows 1 and 3: a tenth of their entries with confidence 1e6 on item rows of magnitude 40
    hot = np.arange(0, items, 10)
    Y[hot] = (rng.standard_normal((len(hot), f)) * 40).astype(np.float32)
    for r in (1, 3):
        lo, hi = C.indptr[r], C.indptr[r + 1]
        sel = np.isin(C.indices[lo:hi], hot)
        C.data[lo:hi][sel] = 1e6
    # rows 0 and 2 must not touch the hot items at all (their operands stay in range)
    for r in (0, 2):
        lo, hi = C.indptr[r], C.indptr[r + 1]
        C.data[lo:hi][np.isin(C.indices[lo:hi], hot)] = 1.0   # weight 0
    X = (rng.standard_normal((4, f)) * 0.1).astype(np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.01)
    exact = oracle.least_squares_cg_f64(C, X, Y, 0.01)
    gpu.fixup_rows(reset=True)
    got = _solve(gpu, C, X.copy(), Y, 0.01, 3)
    n_fix = gpu.fixup_rows(reset=True)
    e_gpu, e_oracle = _row_errors(got, exact), _row_errors(want, exact)
    print("rows re-solved in fp32:", n_fix, "gpu vs fp64", e_gpu, "oracle vs fp64", e_oracle)
    assert np.isfinite(got).all()
    assert n_fix == 2
    assert (e_gpu < np.maximum(TOL, 2.0 * e_oracle)).all()
    # and an ordinary launch leaves the counter alone
    _solve(gpu, _long_row_matrix([700, 3000], items, seed=1), X[:2].copy(), (Y * np.float32(0.01)), 0.01, 3)
    assert gpu.fixup_rows(reset=True) == 0


@pytest.mark.parametrize("scale", [0.01, 1e-3, 1e-4])
def test_small_factors_keep_their_bits(gpu, oracle, scale):
    """Factors of magnitude 0.01 and below (symmetric about zero: a well-conditioned system): the launch's operand scale
    (nm_gram_image_kernel) keeps the low fp16 halves out of the subnormals; every row at least as close to the float64 answer as the
    fp32 oracle (measured 2e-7 .. 4e-7 with the scale, 4e-7 .. 1.3e-6 without, oracle 7e-7 .. 2.9e-6; profiles/scripts/r5b_nm_coldstart.py)."""
    f, items = 128, 9000
    C = _long_row_matrix([600, 1300, 4000, 9000], items, seed=21)
    rng = np.random.default_rng(4)
    Y = ((rng.random((items, f)) - 0.5) * scale).astype(np.float32)
    X = ((rng.random((4, f)) - 0.5) * scale).astype(np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.01)
    exact = oracle.least_squares_cg_f64(C, X, Y, 0.01)
    got = _solve(gpu, C, X.copy(), Y, 0.01, 3)
    e_gpu, e_oracle = _row_errors(got, exact), _row_errors(want, exact)
    print("scale", scale, "gpu vs fp64", e_gpu, "oracle vs fp64", e_oracle)
    assert gpu.fixup_rows(reset=True) == 0
    assert (e_gpu < np.maximum(1e-6, 1.5 * e_oracle)).all()


def test_all_positive_cold_start_factors(gpu, oracle):
    """The first sweep of every fit starts from factors uniform in (0, 0.01) (implicit/gpu/als.py:98-101): A_u is then a rank-one
    matrix plus a perturbation four orders of magnitude smaller, and three CG steps amplify fp32 rounding to 1e-4 .. 1e-2 of the
    answer in EVERY implementation (the fp32 oracle itself: 8e-5 .. 1.3e-2 per row from float64).  The explicit normal matrix
    rounds its entries once more than the reference's implicit product does; measured 3e-3 .. 4e-2 per row here against 9e-4 ..
    1.2e-2 for the streamed fp32 kernels (IMP_NM=0) -- independent of the operand scale, i.e. conditioning, not the fp16 split.
    The bar is the oracle's own distance over the sweep, within a factor that says "same regime"; from the second sweep on the
    lock-step fit test holds every half sweep to 1e-4 (tests/test_gpu_als.py)."""
    f, items = 128, 9000
    C = _long_row_matrix([600, 1300, 4000, 9000], items, seed=21)
    rng = np.random.default_rng(4)
    Y = (rng.random((items, f)) * 0.01).astype(np.float32)
    X = (rng.random((4, f)) * 0.01).astype(np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.01)
    exact = oracle.least_squares_cg_f64(C, X, Y, 0.01)
    got = _solve(gpu, C, X.copy(), Y, 0.01, 3)
    fro = lambda a: np.linalg.norm(a.astype(np.float64) - exact) / np.linalg.norm(exact)  # noqa: E731
    print("per row: gpu vs fp64", _row_errors(got, exact), "oracle vs fp64", _row_errors(want, exact), "sweep: gpu %.2e oracle %.2e" % (fro(got), fro(want)))
    assert np.isfinite(got).all() and gpu.fixup_rows(reset=True) == 0
    assert fro(got) < 10.0 * fro(want)


@pytest.mark.parametrize("f", [64, 128])
def test_fp16_factor_storage(gpu, oracle, f):
    """fp16 factors: y is an fp16 number, the split needs one half; result = the fp32 solve of the rounded inputs, rounded once."""
    items = 7000
    C = _long_row_matrix([520, 1800, 4500, 100], items, seed=6)
    rng = np.random.default_rng(3)
    Y = (rng.standard_normal((items, f)) * 0.1).astype(np.float16).astype(np.float32)
    X = (rng.standard_normal((4, f)) * 0.1).astype(np.float16).astype(np.float32)
    want = X.copy()
    oracle.least_squares_cg(C, want, Y, 0.05)
    got = _solve(gpu, C, X.copy(), Y, 0.05, 3, dtype="float16")
    # one fp16 rounding of the result: half an ulp of the row's largest element
    bar = np.abs(want).max(axis=1, keepdims=True) * 2.0 ** -10
    assert (np.abs(got - want) <= bar + 1e-4 * np.abs(want)).all()


def test_old_long_row_kernels_still_agree():
    """IMP_NM=0 (one streamed pass per CG step) and the default give the same rows within the parity bar."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r + "/tests")
import implicit_amd.gpu as gpu
from test_gpu_nm import _long_row_matrix, _solve, LENGTHS
C = _long_row_matrix(LENGTHS, 20000, seed=1)
rng = np.random.default_rng(0)
Y = (rng.standard_normal((20000, 128)) * 0.1).astype(np.float32)
X = (rng.standard_normal((C.shape[0], 128)) * 0.1).astype(np.float32)
np.save(sys.argv[1], _solve(gpu, C, X, Y, 0.05, 3))
""" % (ROOT, ROOT)
    import tempfile
    out = []
    with tempfile.TemporaryDirectory() as d:
        for tag, env in (("nm", {}), ("old", {"IMP_NM": "0"})):
            path = os.path.join(d, tag + ".npy")
            subprocess.run([sys.executable, "-c", code, path], check=True, env={**os.environ, **env}, timeout=600)
            out.append(np.load(path))
    err = np.linalg.norm(out[0] - out[1], axis=1) / np.maximum(np.linalg.norm(out[1], axis=1), 1e-30)
    print("nm vs old per-row rel max %.2e" % err.max())
    assert err.max() < TOL
    assert not np.array_equal(out[0], out[1])  # they ARE different kernels
