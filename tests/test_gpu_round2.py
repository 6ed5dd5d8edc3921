"""Round-2 additions: hygiene of the C-ABI (empty inputs, malformed CSR, int64 offsets / row blocks, failed_row),
oracle parity of the fold-in paths, the A/B environment switches, concurrent callers, and the model-level multi-GPU
entry with a one-rank communicator."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import scipy.sparse as sp

from implicit_amd.synthetic import synthetic_csr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def test_gramian_of_an_empty_matrix_is_reg_times_identity(gpu):
    """calculate_yty on a 0-row Matrix (an empty shard of the multi-GPU driver): reg * I, as the reference's GEMM +
    l2_regularize pair leaves (als.cu:122-152)."""
    solver = gpu.LeastSquaresSolver()
    for f in (16, 128):
        out = gpu.Matrix(np.ones((f, f), dtype=np.float32))
        solver.calculate_yty(gpu.Matrix(np.zeros((0, f), dtype=np.float32)), out, 0.5)
        np.testing.assert_array_equal(out.to_numpy(), 0.5 * np.eye(f, dtype=np.float32))
        full = gpu.Matrix(np.ones((10, f), dtype=np.float32))
        solver.calculate_yty(full[3:3], out, 0.25)      # empty row-range view
        np.testing.assert_array_equal(out.to_numpy(), 0.25 * np.eye(f, dtype=np.float32))


def test_malformed_csr_is_rejected_on_the_host(gpu):
    good = synthetic_csr(50, 40, 400, seed=1)
    bad = good.copy()
    bad.indices = bad.indices.copy()
    bad.indices[5] = 40                                  # == cols: out of range
    with pytest.raises(ValueError, match="column index out of range"):
        gpu.CSRMatrix(bad)
    bad.indices[5] = -1
    with pytest.raises(ValueError, match="column index out of range"):
        gpu.CSRMatrix(bad)
    bad = good.copy()
    bad.indptr = bad.indptr.copy()
    bad.indptr[3], bad.indptr[4] = bad.indptr[4], bad.indptr[3] - 1
    with pytest.raises(ValueError):
        gpu.CSRMatrix(bad)


def test_cholesky_failure_reports_the_row(gpu):
    """A non-positive-definite system raises ValueError (as _als.pyx:136-138) and names the smallest failing row."""
    f = 16
    C = sp.csr_matrix(np.array([[2.0, 0, 3.0], [0, 2.0, 0], [1.5, 0, 0]], dtype=np.float32))
    Y = gpu.Matrix(np.zeros((3, f), dtype=np.float32))          # YtY = 0 and reg = 0: singular for every row
    X = gpu.Matrix.zeros(3, f)
    gram = gpu.Matrix.zeros(f, f)
    with pytest.raises(ValueError, match="cholesky") as info:
        gpu.LeastSquaresSolver().least_squares_cholesky(gpu.CSRMatrix(C), X, gram, Y, 0.0)
    assert info.value.failed_row == 0


@pytest.mark.parametrize("solver_kind", ["cg", "cholesky"])
def test_int64_offsets_and_row_blocks_match_the_int32_matrix(gpu, solver_kind, monkeypatch):
    """imp_csr_create64: int64 indptr (what scipy hands over beyond 2^31 nonzeros; the CPU reference accepts it,
    _als.pyx:76).  With the block limit lowered the same matrix is held as several row blocks: identical results."""
    f, reg = 64, 0.05
    C = synthetic_csr(4000, 900, 120_000, seed=2, neg_frac=0.05, empty_frac=0.02)
    rng = np.random.default_rng(1)
    X0 = rng.random((4000, f), dtype=np.float32) * 0.1
    Y0 = rng.random((900, f), dtype=np.float32) * 0.1
    solver = gpu.LeastSquaresSolver()
    Yd = gpu.Matrix(Y0)
    gram = gpu.Matrix.zeros(f, f)
    solver.calculate_yty(Yd, gram, reg if solver_kind == "cg" else 0.0)

    def solve(Cd):
        Xd = gpu.Matrix(X0)
        if solver_kind == "cg":
            solver.least_squares(Cd, Xd, gram, Yd, 3)
        else:
            solver.least_squares_cholesky(Cd, Xd, gram, Yd, reg)
        return Xd.to_numpy(), solver.calculate_loss(Cd, Xd, Yd, reg)

    want, want_loss = solve(gpu.CSRMatrix(C))
    C64 = C.copy()
    C64.indptr = C64.indptr.astype(np.int64)
    C64.indices = C64.indices.astype(np.int64)
    got, got_loss = solve(gpu.CSRMatrix(C64))
    np.testing.assert_array_equal(got, want)
    monkeypatch.setenv("IMP_CSR_PART_NNZ", "25000")      # ~5 row blocks
    blocks = gpu.CSRMatrix(C64)
    monkeypatch.delenv("IMP_CSR_PART_NNZ")
    got, got_loss = solve(blocks)
    assert rel(got, want) < 1e-6                           # the row schedule differs per block, the arithmetic per row does not
    assert got_loss == pytest.approx(want_loss, rel=1e-5)


def test_fold_in_matches_the_oracle(gpu, oracle):
    """recalculate_user / recalculate_item / partial_fit_* against the oracle's solvers on the same rows (the round-1
    tests compared the model with itself): Cholesky fold-in as cpu/als.py:221-241."""
    from implicit_amd.als import AlternatingLeastSquares

    C = synthetic_csr(600, 300, 9000, seed=4)
    model = AlternatingLeastSquares(factors=32, regularization=0.05, random_state=3, use_gpu=True, iterations=3)
    model.fit(C, show_progress=False)
    Xh, Yh = model.user_factors.to_numpy(), model.item_factors.to_numpy()
    users = np.array([5, 17, 99, 400])
    got = model.recalculate_user(users, C[users]).to_numpy()
    want = np.zeros((len(users), 32), dtype=np.float32)
    oracle.least_squares(C[users], want, Yh, 0.05)
    assert rel(got, want) < 1e-4
    one = model.recalculate_user(7, C[7]).to_numpy()
    want1 = np.zeros((1, 32), dtype=np.float32)
    oracle.least_squares(C[7], want1, Yh, 0.05)
    assert rel(one, want1) < 1e-4
    Ct = C.T.tocsr()
    items = np.array([0, 3, 250])
    got = model.recalculate_item(items, Ct[items]).to_numpy()
    want = np.zeros((len(items), 32), dtype=np.float32)
    oracle.least_squares(Ct[items], want, Xh, 0.05)
    assert rel(got, want) < 1e-4
    # partial_fit_users: new user ids beyond the model grow the matrix; rows equal the fold-in solution
    new_ids = np.array([600, 602])
    model.partial_fit_users(new_ids, C[[1, 2]])
    after = model.user_factors.to_numpy()
    assert after.shape == (603, 32) and not after[601].any()
    want = np.zeros((2, 32), dtype=np.float32)
    oracle.least_squares(C[[1, 2]], want, Yh, 0.05)
    assert rel(after[new_ids], want) < 1e-4
    np.testing.assert_array_equal(after[:600], Xh)
    # the cached unregularised gramian is dropped when the factors it was built from change
    assert model._XtX0 is None                               # partial_fit_users changed the user factors
    grown = sp.csr_matrix((Ct[2].data, Ct[2].indices, Ct[2].indptr), shape=(1, 603))
    model.recalculate_item(2, grown)
    assert model._XtX0 is not None and model._YtY0 is not None  # both unregularised gramians are cached now ...
    model.partial_fit_items(np.array([2]), grown)
    assert model._YtY0 is None and model._YtY is None           # ... and the item-side ones are dropped again


_SWITCH_SCRIPT = r"""
import sys, numpy as np, warnings
sys.path.insert(0, {root!r})
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import synthetic_csr
from oracle import oracle
oracle.build()
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))
solver = gpu.LeastSquaresSolver()
for f in (64, 128, 50, 100):   # 50 / 100: zero-padded onto the f = 64 / 128 kernels (IMP_NO_PAD=1: the generic kernels)
    C = synthetic_csr(3000, 700, 150_000, seed=6, neg_frac=0.05, empty_frac=0.01)   # item side: rows > 512 nnz
    for M in (C, C.T.tocsr()):
        rng = np.random.default_rng(2)
        X0 = rng.random((M.shape[0], f), dtype=np.float32) * 0.2 - 0.1
        Y0 = rng.random((M.shape[1], f), dtype=np.float32) * 0.2 - 0.1
        Xd, Yd, gram = gpu.Matrix(X0), gpu.Matrix(Y0), gpu.Matrix.zeros(f, f)
        solver.calculate_yty(Yd, gram, 0.05)
        assert rel(gram.to_numpy(), oracle.gramian(Y0) + np.float32(0.05) * np.eye(f, dtype=np.float32)) < 1e-6
        solver.least_squares(gpu.CSRMatrix(M), Xd, gram, Yd, 3)
        want = X0.copy()
        oracle.least_squares_cg(M, want, Y0, 0.05, cg_steps=3, YtY=gram.to_numpy())
        assert rel(Xd.to_numpy(), want) < 1e-4, ("cg", f, rel(Xd.to_numpy(), want))
        if f in (64, 128):   # float16 storage: packed 64-entry tiles 
            X16, Y16 = X0.astype(np.float16), Y0.astype(np.float16)
            Xh, Yh = gpu.Matrix(X16), gpu.Matrix(Y16)
            solver.calculate_yty(Yh, gram, 0.05)
            solver.least_squares(gpu.CSRMatrix(M), Xh, gram, Yh, 3)
            want = X16.astype(np.float32)
            oracle.least_squares_cg(M, want, Y16.astype(np.float32), 0.05, cg_steps=3, YtY=gram.to_numpy())
            assert rel(Xh.to_numpy().astype(np.float32), want) < 1e-3, ("cg fp16", f)
    if f == 64:
        Xd = gpu.Matrix.zeros(C.shape[0], f)
        solver.calculate_yty(Yd if Yd.shape[0] == C.shape[1] else gpu.Matrix(Y0), gram, 0.0)
Cw = synthetic_csr(1200, 900, 60_000, seed=8, neg_frac=0.05)          # f = 256: workgroup-shared gramian / generic kernel
rng = np.random.default_rng(3)
X0 = rng.random((1200, 256), dtype=np.float32) * 0.2 - 0.1
Y0 = rng.random((900, 256), dtype=np.float32) * 0.2 - 0.1
Xd, Yd, gram = gpu.Matrix(X0), gpu.Matrix(Y0), gpu.Matrix.zeros(256, 256)
solver.calculate_yty(Yd, gram, 0.05)
solver.least_squares(gpu.CSRMatrix(Cw), Xd, gram, Yd, 3)
want = X0.copy()
oracle.least_squares_cg(Cw, want, Y0, 0.05, cg_steps=3, YtY=gram.to_numpy())
assert rel(Xd.to_numpy(), want) < 1e-4, ("cg f=256", rel(Xd.to_numpy(), want))
Cc = synthetic_csr(2000, 500, 60_000, seed=3)
rng = np.random.default_rng(5)
Y0 = rng.random((500, 64), dtype=np.float32) * 0.2 - 0.1
Yd, gram, Xd = gpu.Matrix(Y0), gpu.Matrix.zeros(64, 64), gpu.Matrix.zeros(2000, 64)
solver.calculate_yty(Yd, gram, 0.0)
solver.least_squares_cholesky(gpu.CSRMatrix(Cc), Xd, gram, Yd, 0.05)
want = np.zeros((2000, 64), dtype=np.float32)
oracle.least_squares(Cc, want, Y0, 0.05)
assert rel(Xd.to_numpy(), want) < 1e-4, ("cholesky", rel(Xd.to_numpy(), want))
Y1 = rng.random((500, 100), dtype=np.float32) * 0.2 - 0.1   # f = 100: the workgroup-per-row kernel 
Yd, gram, Xd = gpu.Matrix(Y1), gpu.Matrix.zeros(100, 100), gpu.Matrix.zeros(2000, 100)
solver.calculate_yty(Yd, gram, 0.0)
solver.least_squares_cholesky(gpu.CSRMatrix(Cc), Xd, gram, Yd, 0.05)
want = np.zeros((2000, 100), dtype=np.float32)
oracle.least_squares(Cc, want, Y1, 0.05)
assert rel(Xd.to_numpy(), want) < 1e-4, ("cholesky f=100", rel(Xd.to_numpy(), want))
Y2 = rng.random((500, 128), dtype=np.float32) * 0.2 - 0.1   # f = 128: normal matrices on the matrix cores + LDS factorisation (IMP_CHOL_NM=0: workgroup kernel)
Yd, gram, Xd = gpu.Matrix(Y2), gpu.Matrix.zeros(128, 128), gpu.Matrix.zeros(2000, 128)
solver.calculate_yty(Yd, gram, 0.0)
solver.least_squares_cholesky(gpu.CSRMatrix(Cc), Xd, gram, Yd, 0.05)
want = np.zeros((2000, 128), dtype=np.float32)
oracle.least_squares(Cc, want, Y2, 0.05)
assert rel(Xd.to_numpy(), want) < 1e-4, ("cholesky f=128", rel(Xd.to_numpy(), want))
items = (rng.standard_normal((5000, 64)) * 0.1).astype(np.float32)
q = (rng.standard_normal((40, 64)) * 0.1).astype(np.float32)
ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), 10)
wi, wd = oracle.topk(items, q, 10)
assert (ids == wi).mean() > 0.99
items = (rng.standard_normal((12000, 64)) * 0.1).astype(np.float32)   # enough items for the emit path (resident fp16 form / IMP_TOPK_RESIDENT=0: the six-product kernel)
ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(q), 10)
wi, wd = oracle.topk(items, q, 10)
assert (ids == wi).mean() > 0.99 and np.allclose(d, wd, rtol=3e-5)
print("switch ok")
"""


@pytest.mark.parametrize("switch", ["IMP_STRIPE=0", "IMP_SEGMENT=128", "IMP_TOPK_NO_FAST=1", "IMP_TOPK_NO_EMIT=1", "IMP_TOPK_RESIDENT=0",
                                    "IMP_TOPK_SCREEN=0", "IMP_TOPK_FP32_MFMA=1", "IMP_OVERSUB=3", "IMP_NO_PAD=1", "IMP_NM=0", "IMP_NM_SEGMENT=128", "IMP_F256_OLD=1",
                                    "IMP_CHOL_NM=0", "IMP_CHOL_PAD=0"])
def test_ab_switch_paths_keep_parity(gpu, switch):
    """Every debug / A-B environment switch selects kernels the default run does not take (they are read once per
    process, hence the subprocess): CG both orientations at f = 64 / 128, Cholesky f = 64 and top-k against the oracle."""
    name, value = switch.split("=")
    env = dict(os.environ, **{name: value})
    out = subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT.format(root=ROOT)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "switch ok" in out.stdout, out.stderr[-2000:]


def test_concurrent_callers_on_one_device(gpu, oracle):
    """ctypes releases the GIL around every C-ABI call and the model caches ONE KnnQuery / solver: calls on a device are
    serialised inside the library (per-device lock), so threads sharing a model get the single-threaded answers."""
    rng = np.random.default_rng(0)
    items = gpu.Matrix((rng.standard_normal((20_000, 64)) * 0.1).astype(np.float32))
    queries = [(rng.standard_normal((64, 64)) * 0.1).astype(np.float32) for _ in range(8)]
    knn = gpu.KnnQuery()
    want = [knn.topk(items, gpu.Matrix(q), 10) for q in queries]
    C = synthetic_csr(1500, 20_000, 40_000, seed=9)
    Y = items
    solver = gpu.LeastSquaresSolver()
    gram = gpu.Matrix.zeros(64, 64)
    solver.calculate_yty(Y, gram, 0.1)
    X0 = rng.random((1500, 64), dtype=np.float32) * 0.01
    Xref = gpu.Matrix(X0)
    Cd = gpu.CSRMatrix(C)
    solver.least_squares(Cd, Xref, gram, Y, 3)
    want_x = Xref.to_numpy()
    errors = []

    def knn_worker(i):
        try:
            for _ in range(5):
                ids, d = knn.topk(items, gpu.Matrix(queries[i]), 10)
                np.testing.assert_array_equal(ids, want[i][0])
                np.testing.assert_array_equal(d, want[i][1])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def solve_worker():
        try:
            for _ in range(5):
                Xd = gpu.Matrix(X0)
                solver.calculate_yty(Y, gram, 0.1)
                solver.least_squares(Cd, Xd, gram, Y, 3)
                np.testing.assert_array_equal(Xd.to_numpy(), want_x)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=knn_worker, args=(i,)) for i in range(8)] + [threading.Thread(target=solve_worker)] * 1
    threads += [threading.Thread(target=solve_worker)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]


def test_model_fit_with_a_communicator(gpu, oracle):
    """AlternatingLeastSquares(comm=...): the multi-GPU fit entry.  One real rank here (the N = 2 logic runs on CPU in
    tests/test_sharded_gloo.py): a one-rank communicator is the plain fit; IMP_FORCE... is not needed -- nranks == 1
    short-circuits to the single-GPU loop -- so the sharded body is driven directly with chunks."""
    from implicit_amd.als import AlternatingLeastSquares
    from implicit_amd.gpu import sharded

    C = synthetic_csr(800, 500, 20_000, seed=12)
    comm = gpu.Comm(gpu.Comm.unique_id(), 1, 0)
    plain = AlternatingLeastSquares(factors=64, regularization=0.05, random_state=5, use_gpu=True, iterations=2)
    plain.fit(C, show_progress=False)
    model = AlternatingLeastSquares(factors=64, regularization=0.05, random_state=5, use_gpu=True, iterations=2)
    model.comm = comm
    model.fit(C, show_progress=False)                      # nranks == 1: same loop
    np.testing.assert_array_equal(model.user_factors.to_numpy(), plain.user_factors.to_numpy())
    # the sharded body itself (what every rank of an N-GPU job runs), chunked exchange included
    model2 = AlternatingLeastSquares(factors=64, regularization=0.05, random_state=5, use_gpu=True, iterations=2)
    Cui = C.astype(np.float32)
    model2._initial_factors(*C.shape)
    sharded.fit_sharded(model2, Cui, comm, chunks=3)     # one rank: its block of user rows is the whole matrix
    assert gpu.get_oversubscribe() == 1                   # the driver restores the launch shape it found
    # the model-level entry every rank of an N-GPU job goes through (block sizes all-reduced, initial factors agreed through a
    # sum all-reduce, set-up exchange, iterations, per-rank loss), here with the one rank there is
    model3 = AlternatingLeastSquares(factors=64, regularization=0.05, random_state=5, use_gpu=True, iterations=2,
                                     calculate_training_loss=True)
    model3.comm = comm
    seen = []
    model3._fit_sharded(Cui, lambda it, dt, loss: seen.append(it))
    assert seen == [0, 1]
    assert rel(model3.user_factors.to_numpy(), plain.user_factors.to_numpy()) < 1e-5
    assert rel(model3.item_factors.to_numpy(), plain.item_factors.to_numpy()) < 1e-5
    assert rel(model2.user_factors.to_numpy(), plain.user_factors.to_numpy()) < 1e-5
    assert rel(model2.item_factors.to_numpy(), plain.item_factors.to_numpy()) < 1e-5
    import pickle

    assert pickle.loads(pickle.dumps(model)).comm is None


def test_topk_emit_path_and_its_fallbacks(gpu, oracle):
    """The score-matrix-free top-k (>= 32768 items): random rows with both filters, and the rows it must hand back to the
    materialising path -- exact ties at the k-th score (all-equal rows: the reference heap's arrival-order rule), rows
    whose whole threshold subset is filtered (no valid lower bound), candidate overflow (thousands of tied maxima)."""
    rng = np.random.default_rng(11)
    ni, f, k = 40_000, 64, 10
    items = (rng.standard_normal((ni, f)) * 0.1).astype(np.float32)
    items[::7] = items[3]                          # thousands of identical items: ties everywhere they score high
    queries = (rng.standard_normal((70, f)) * 0.1).astype(np.float32)
    queries[5] = 0.0                               # all scores 0: every item ties
    queries[6] = items[3] * 4                      # the duplicated item is the best: > 5000 candidates tie at the top
    # small tie groups scattered over the columns: the tie at the k-th score is resolved INSIDE the candidate list with the
    # heap's arrival-order rule (no fallback): 25 copies of one vector, some of them filtered for query 8
    group = np.sort(rng.choice(np.setdiff1d(np.arange(ni), np.arange(0, ni, 7)), 25, replace=False))
    items[group] = items[group[0]]
    queries[7] = items[group[0]] * 3
    queries[8] = items[group[0]] * 3 + items[11] * 0.5
    # per-query filter: query 0 filters the whole threshold subset (every 32nd 128-item block), query 1 a random set
    sub = np.concatenate([np.arange(b * 128, min(ni, b * 128 + 128)) for b in range(0, (ni + 127) // 128, 32)])
    rows = np.concatenate([np.zeros(len(sub), dtype=np.int64), np.ones(500, dtype=np.int64), np.full(6, 8, dtype=np.int64)])
    cols = np.concatenate([sub, rng.choice(ni, 500, replace=False), group[::4][:6]])
    liked = sp.csr_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(70, ni))
    banned = rng.choice(ni, 300, replace=False).astype(np.int32)
    ids, d = gpu.KnnQuery().topk(gpu.Matrix(items), gpu.Matrix(queries), k, query_filter=gpu.COOMatrix(liked.tocoo()),
                                 item_filter=gpu.IntVector(banned))
    want_ids, want_d = oracle.topk(items, queries, k, filter_query_items=liked, filter_items=banned)
    np.testing.assert_allclose(d, want_d, rtol=2e-5, atol=1e-7)
    exact_rows = [5, 6, 7, 8]                      # tie rows: ids must follow select.h bit for bit
    np.testing.assert_array_equal(ids[exact_rows], want_ids[exact_rows])
    assert not (set(ids[0]) & set(sub.tolist())) and not (set(ids.ravel().tolist()) & set(banned.tolist()))
    plain = [r for r in range(70) if r not in exact_rows]
    assert (ids[plain] == want_ids[plain]).mean() > 0.97     # duplicated items tie: order inside a tie group may differ
    for r in plain:
        assert sorted(d[r], reverse=True) == list(d[r])


@pytest.mark.parametrize("f", [64, 128])
def test_native_fp16_factor_storage(gpu, oracle, f):
    """fp16 factor storage is read and written by the f = 64 / 128 kernels themselves (half2 / 8-byte loads converted in
    registers or inside the FMA, fp32 arithmetic and CG state, as implicit/gpu/als.cu:41,55,109): equal to solving an fp32
    copy of the fp16 matrices and rounding the result -- bit for bit where both storages share
    a kernel, to the last fp16 place elsewhere -- and within 1e-3 of the oracle run on the fp16-rounded inputs (the result is
    stored in fp16)."""
    C = synthetic_csr(4000, 1500, 200_000, seed=2, neg_frac=0.05, empty_frac=0.01)   # item side has rows > 512 nnz
    rng = np.random.default_rng(4)
    for M in (C, C.T.tocsr()):
        X16 = (rng.random((M.shape[0], f), dtype=np.float32) * 0.2 - 0.1).astype(np.float16)
        Y16 = (rng.random((M.shape[1], f), dtype=np.float32) * 0.2 - 0.1).astype(np.float16)
        solver = gpu.LeastSquaresSolver()
        Xd, Yd, gram = gpu.Matrix(X16), gpu.Matrix(Y16), gpu.Matrix.zeros(f, f)
        assert Xd.itemsize == 2
        solver.calculate_yty(Yd, gram, 0.05)
        gram32 = gpu.Matrix.zeros(f, f)
        solver.calculate_yty(gpu.Matrix(Y16.astype(np.float32)), gram32, 0.05)
        np.testing.assert_array_equal(gram.to_numpy(), gram32.to_numpy())   # the products are fp32 either way
        solver.least_squares(gpu.CSRMatrix(M), Xd, gram, Yd, 3)
        got = Xd.to_numpy()
        assert got.dtype == np.float16
        # the same solve on an fp32 copy, rounded at the end
        X32, Y32 = gpu.Matrix(X16.astype(np.float32)), gpu.Matrix(Y16.astype(np.float32))
        solver.least_squares(gpu.CSRMatrix(M), X32, gram, Y32, 3)
        want16 = X32.to_numpy().astype(np.float16)
        # Round 4: the mid-row classes of fp16 storage run on HALF the wavefronts with a packed 64-entry tile
        # (als_cg_qh.hip), so a row's partial sums associate differently from the fp32 kernels': the fp32 results agree to
        # rounding noise and the rounded fp16 values to one unit in the last place on a small fraction of the elements.  The
        # short rows share their kernels with fp32 storage and stay bit-identical.  The long rows (normal-matrix kernels,
        # als_cg_nm.hip) split an fp32 factor into two fp16 halves and take an fp16 factor as it is: the same numbers unless
        # the scaled factor 2^e y falls into the fp16 subnormals, where the fp32 path keeps one more bit -- last-place
        # differences there too.
        lens = np.diff(M.indptr)
        same_kernel = lens <= 32
        np.testing.assert_array_equal(got[same_kernel], want16[same_kernel])
        a, b = got.astype(np.float32), want16.astype(np.float32)
        # one unit in the last place at the element's magnitude -- for elements much smaller than their row (cancellation in
        # the last x update) at 1/64 of the row's largest: there an fp32 difference of 1e-7 of the row is several fp16 ulps
        scale = np.maximum(np.abs(b), np.abs(b).max(axis=1, keepdims=True) / 64)
        ulp = np.spacing(scale.astype(np.float16)).astype(np.float32)
        assert (np.abs(a - b) <= ulp).all() and (a != b).mean() < 0.02
        want = X16.astype(np.float32)
        oracle.least_squares_cg(M, want, Y16.astype(np.float32), 0.05, cg_steps=3, YtY=gram.to_numpy())
        assert rel(got.astype(np.float32), want) < 1e-3
        empty = np.diff(M.indptr) == 0
        assert not got[empty].any()


def _cg_fp64_row(row, x0, Y, A0, cg_steps):
    """One row of the oracle's CG (implicit/cpu/_als.pyx:179-244) in float64; A0 = YtY + reg I."""
    idx, c = row.indices, row.data.astype(np.float64)
    Yu, A0, x = Y[idx].astype(np.float64), A0.astype(np.float64), x0.astype(np.float64)
    cm1 = np.abs(c) - 1.0
    r = -A0 @ x + Yu.T @ (np.where(c > 0, c, 0.0) - cm1 * (Yu @ x))
    p, rsold = r.copy(), r @ r
    if rsold < 1e-20:
        return x
    for _ in range(cg_steps):
        Ap = A0 @ p + Yu.T @ (cm1 * (Yu @ p))
        alpha = rsold / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        rsnew = r @ r
        if rsnew < 1e-20:
            break
        p, rsold = r + (rsnew / rsold) * p, rsnew
    return x


@pytest.mark.parametrize("f", [64, 128])
@pytest.mark.parametrize("cg_steps", [3, 0])
def test_cluster_resident_long_rows(gpu, oracle, f, cg_steps):
    """Rows of 513 .. 4096 nonzeros are solved resident across a cluster of 4 / 8 / 16 workgroups that exchange one f-vector
    per workgroup and pass through L2 (als_cg_cluster.hip); longer rows stay on the streamed path.  Row lengths sit on and
    around every class boundary; more rows than clusters in every class, so the persistent loop and the double-buffered
    exchange slots are exercised; an already-solved row (residual 0) takes the early exit in the whole cluster.  Checked per
    row against the oracle's CG evaluated in float64 and, as a whole, against the oracle itself."""
    rng = np.random.default_rng(11)
    lens = np.concatenate([np.arange(500, 530), [1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 5000, 9000],
                           rng.integers(513, 1025, 150), rng.integers(1025, 2049, 90), rng.integers(2049, 4097, 70),
                           rng.integers(1, 400, 300), [0, 0]])
    rng.shuffle(lens)
    cols = 12_000
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens] + [[]]).astype(np.int32)
    data = (1.0 + 4.0 * rng.random(len(indices), dtype=np.float32)).astype(np.float32)
    data[rng.random(len(data)) < 0.03] *= -1
    C = sp.csr_matrix((data, indices, indptr), shape=(len(lens), cols))
    X0 = rng.random((C.shape[0], f), dtype=np.float32) * 0.2 - 0.1
    Y0 = rng.random((cols, f), dtype=np.float32) * 0.2 - 0.1
    solver = gpu.LeastSquaresSolver()
    Xd, Yd, gram = gpu.Matrix(X0), gpu.Matrix(Y0), gpu.Matrix.zeros(f, f)
    solver.calculate_yty(Yd, gram, 0.05)
    Cd = gpu.CSRMatrix(C)
    solver.least_squares(Cd, Xd, gram, Yd, cg_steps)
    got = Xd.to_numpy()
    gram_h = gram.to_numpy()
    long_rows = np.flatnonzero(lens > 512)
    worst = 0.0
    for r in long_rows:
        exact = _cg_fp64_row(C[int(r)], X0[r], Y0, gram_h, cg_steps)
        worst = max(worst, rel(got[r], exact))
    assert worst < 1e-4, worst
    want = X0.copy()
    oracle.least_squares_cg(C, want, Y0, 0.05, cg_steps=cg_steps, YtY=gram_h)
    assert rel(got, want) < 1e-4
    assert not got[lens == 0].any()
    # a second sweep from the solution: deterministic (same bits on a re-run from the same start)
    Xa, Xb = gpu.Matrix(got), gpu.Matrix(got)
    solver.least_squares(Cd, Xa, gram, Yd, 3)
    solver.least_squares(Cd, Xb, gram, Yd, 3)
    np.testing.assert_array_equal(Xa.to_numpy(), Xb.to_numpy())


@pytest.mark.parametrize("f", [64, 128])
def test_oversubscribed_launches_are_bitwise_identical(gpu, f):
    """imp_set_oversubscribe only changes how many workgroups share the row schedule (the multi-GPU driver uses 4x so that
    slots held by RCCL's kernels delay small shares only): every row's arithmetic is unchanged, with or without a foreign
    kernel parked on the device (imp_debug_occupy) -- cluster exchanges included."""
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(1, 600, 4000), rng.integers(513, 4097, 200), [5000, 7000], [0, 0, 0]])
    rng.shuffle(lens)
    cols = 9000
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens] + [[]]).astype(np.int32)
    data = (1.0 + 4.0 * rng.random(len(indices), dtype=np.float32)).astype(np.float32)
    C = gpu.CSRMatrix(sp.csr_matrix((data, indices, indptr), shape=(len(lens), cols)))
    X0 = rng.random((len(lens), f), dtype=np.float32) * 0.2 - 0.1
    Yd = gpu.Matrix(rng.random((cols, f), dtype=np.float32) * 0.2 - 0.1)
    solver, gram = gpu.LeastSquaresSolver(), gpu.Matrix.zeros(f, f)
    solver.calculate_yty(Yd, gram, 0.05)
    results = []
    try:
        for factor, occupy in ((1, 0), (4, 0), (4, 24), (1, 24)):
            gpu.set_oversubscribe(factor)
            if occupy:
                gpu.debug_occupy(occupy, 30_000)
            Xd = gpu.Matrix(X0)
            solver.least_squares(C, Xd, gram, Yd, 3)
            results.append(Xd.to_numpy())
    finally:
        gpu.set_oversubscribe(1)
        gpu.synchronize()
    for other in results[1:]:
        np.testing.assert_array_equal(results[0], other)
    with pytest.raises(ValueError):
        gpu.set_oversubscribe(0)


def test_cholesky_long_rows_are_segment_parallel(gpu, oracle):
    """f = 64 Cholesky: the A-build of a row with more than 1024 nonzeros is cut into 1024-nnz segments built by separate
    wavefronts and summed by the row's wavefront (als_cholesky_f64_partial_kernel) -- lengths on both sides of the
    threshold and of the segment boundaries, negative confidences, against the oracle's Cholesky solve."""
    rng = np.random.default_rng(21)
    lens = np.concatenate([[1023, 1024, 1025, 2047, 2048, 2049, 3000, 5000, 9000], rng.integers(1025, 4000, 40),
                           rng.integers(1, 1024, 400), [0]])
    rng.shuffle(lens)
    cols, f = 12_000, 64
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(cols, int(n), replace=False)) for n in lens] + [[]]).astype(np.int32)
    data = (1.0 + 4.0 * rng.random(len(indices), dtype=np.float32)).astype(np.float32)
    data[rng.random(len(data)) < 0.03] *= -1
    C = sp.csr_matrix((data, indices, indptr), shape=(len(lens), cols))
    Y0 = rng.random((cols, f), dtype=np.float32) * 0.2 - 0.1
    solver = gpu.LeastSquaresSolver()
    Yd, gram, Xd = gpu.Matrix(Y0), gpu.Matrix.zeros(f, f), gpu.Matrix.zeros(C.shape[0], f)
    solver.calculate_yty(Yd, gram, 0.0)
    solver.least_squares_cholesky(gpu.CSRMatrix(C), Xd, gram, Yd, 0.05)
    got = Xd.to_numpy()
    want = np.zeros((C.shape[0], f), dtype=np.float32)
    oracle.least_squares(C, want, Y0, 0.05)
    long_rows = np.flatnonzero(lens > 1024)
    assert max(rel(got[r], want[r]) for r in long_rows) < 2e-4
    assert rel(got, want) < 1e-4
    assert not got[lens == 0].any()




