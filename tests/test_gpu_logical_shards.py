"""N LOGICAL ranks on one GPU (the harness for boxes that expose a single device): the production
multi-GPU path -- `AlternatingLeastSquares(comm=).fit(this rank's block)`, i.e. `sharded.fit_sharded` with `shard_transpose`,
K = 4 row chunks per half sweep, 4x oversubscribed persistent kernels, deferred iterations, the cluster kernels on the
ranks' popular item rows -- runs with the REAL HIP kernels for N = 2, 4 and 8; only the transport differs (device-to-device
copies between the ranks' replicas through `local_comm.LocalComm`, with a parked foreign kernel per exchange standing in for
RCCL's resident send / recv kernels).  The result must equal the unsharded fit of the same matrix."""
import numpy as np
import pytest
import scipy.sparse as sp

from implicit_amd.synthetic import grid_shards

pytestmark = pytest.mark.gpu

# BASELINE configs[3] in miniature: 8 x 8 block grid, popular items spread over the item ranges; item rows average 750
# nonzeros (the popular ones several thousand: cluster classes and the streamed remainder), user rows 100
USERS, ITEMS, NNZ, GRID, F = 24_000, 3_200, 2_400_000, 8, 128


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def unsharded(gpu):
    """(warm state X1, Y1; its two-iteration continuation X2, Y2) of the one-GPU model.  The comparison runs from a WARM
    state: the cold first sweeps (factors 0.01 * U(0,1)) are ill conditioned and amplify the fp32 association noise of the
    gramian to 1e-4 .. 2e-4 between ANY two correct evaluations (the oracle itself sits 2.2e-4 from its
    fp64 twin there), which would hide a real discrepancy of that size."""
    from implicit_amd.als import AlternatingLeastSquares

    C = sp.vstack([grid_shards(r, GRID, USERS, ITEMS, NNZ, GRID, gamma=2.0, seed=11)[0] for r in range(GRID)]).tocsr()
    lens = np.diff(C.T.tocsr().indptr)
    assert lens.max() > 4096 and (lens > 512).sum() > 100  # the item side does reach the cluster / streamed classes
    cold = AlternatingLeastSquares(factors=F, regularization=0.05, random_state=3, use_gpu=True, iterations=2)
    cold.fit(C, show_progress=False)
    X1, Y1 = cold.user_factors.to_numpy(), cold.item_factors.to_numpy()
    warm = AlternatingLeastSquares(factors=F, regularization=0.05, use_gpu=True, iterations=2)
    warm.user_factors, warm.item_factors = gpu.Matrix(X1), gpu.Matrix(Y1)
    warm.fit(C, show_progress=False)
    return X1, Y1, warm.user_factors.to_numpy(), warm.item_factors.to_numpy()


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_logical_ranks_reproduce_the_unsharded_fit(gpu, unsharded, nranks):
    from implicit_amd.als import AlternatingLeastSquares
    from implicit_amd.gpu import local_comm

    before = gpu.get_oversubscribe()
    X1, Y1, want_x, want_y = unsharded

    def rank_body(comm):
        # every rank generates ONLY its block of user rows, as bench.py --gpus N and a production launcher do
        block = grid_shards(comm.rank, comm.nranks, USERS, ITEMS, NNZ, GRID, gamma=2.0, seed=11)[0]
        model = AlternatingLeastSquares(factors=F, regularization=0.05, use_gpu=True, iterations=2, comm=comm)
        # rank 0's initial factors win (the others' are overwritten by the set-up broadcast)
        model.user_factors = gpu.Matrix(X1 if comm.rank == 0 else np.full_like(X1, 7.0))
        model.item_factors = gpu.Matrix(Y1 if comm.rank == 0 else np.full_like(Y1, -7.0))
        model.fit(block, show_progress=False)
        return model.user_factors.to_numpy(), model.item_factors.to_numpy(), comm.stats

    out = local_comm.run(nranks, rank_body, gpu=gpu, occupy=(32, 1500))
    assert gpu.get_oversubscribe() == before  # the launch shape of later single-GPU calls is what it was
    X0, Y0, stats = out[0]
    for X, Y, _ in out[1:]:  # replicas identical bit for bit
        np.testing.assert_array_equal(X, X0)
        np.testing.assert_array_equal(Y, Y0)
    # two iterations: every rank received every other rank's rows of both factor matrices, chunk by chunk
    assert stats["allgather_rows_copied"] == 2 * (nranks - 1) * (USERS + ITEMS)
    assert stats["alltoall_rows_copied"] > 0 and stats["occupied"] >= 2 * 2 * 4
    # same kernels on the same rows; what differs: the gramian's association across ranks and the long-row plans of the chunks
    ex, ey = rel(X0, want_x), rel(Y0, want_y)
    print(f"logical ranks {nranks}: rel X {ex:.2e} Y {ey:.2e}")
    assert ex < 5e-5 and ey < 5e-5


def test_cold_start_through_the_logical_ranks(gpu):
    """The whole production entry from a cold start, random_state differing per rank (rank 0's draw wins): N = 8 against the
    one-GPU fit, at the bar the cold sweeps allow (see `unsharded`)."""
    from implicit_amd.als import AlternatingLeastSquares
    from implicit_amd.gpu import local_comm

    users, items, nnz = 6000, 1600, 450_000
    C = sp.vstack([grid_shards(r, GRID, users, items, nnz, GRID, gamma=2.0, seed=5)[0] for r in range(GRID)]).tocsr()
    one = AlternatingLeastSquares(factors=64, regularization=0.05, random_state=3, use_gpu=True, iterations=3)
    one.fit(C, show_progress=False)

    def rank_body(comm):
        block = grid_shards(comm.rank, comm.nranks, users, items, nnz, GRID, gamma=2.0, seed=5)[0]
        model = AlternatingLeastSquares(factors=64, regularization=0.05, random_state=3 + comm.rank, use_gpu=True, iterations=3,
                                        comm=comm)
        model.fit(block, show_progress=False)
        return model.user_factors.to_numpy(), model.item_factors.to_numpy()

    out = local_comm.run(8, rank_body, gpu=gpu)
    for X, Y in out[1:]:
        np.testing.assert_array_equal(X, out[0][0])
        np.testing.assert_array_equal(Y, out[0][1])
    ex, ey = rel(out[0][0], one.user_factors.to_numpy()), rel(out[0][1], one.item_factors.to_numpy())
    print(f"cold start, 8 logical ranks: rel X {ex:.2e} Y {ey:.2e}")
    assert ex < 1e-3 and ey < 1e-3


def test_copy_rows_checks_its_arguments(gpu):
    a, b = gpu.Matrix(np.arange(40, dtype=np.float32).reshape(10, 4)), gpu.Matrix.zeros(6, 4)
    b.copy_rows_from(1, a, 7, 3)
    got = b.to_numpy()
    np.testing.assert_array_equal(got[1:4], np.arange(28, 40, dtype=np.float32).reshape(3, 4))
    assert not got[0].any() and not got[4:].any()
    with pytest.raises(IndexError):
        b.copy_rows_from(4, a, 0, 3)
    with pytest.raises(ValueError):
        b.copy_rows_from(0, gpu.Matrix.zeros(3, 5), 0, 1)


def test_sharded_recommend_and_similar_items_on_logical_ranks(gpu, unsharded):
    """Inference on N ranks: queries cut into contiguous slices, replicated factors, no collective
    (`sharded.recommend` / `sharded.similar_items`); the slices of 8 logical ranks concatenated are the one-GPU call's
    result bit for bit -- liked-items filter, an item filter and a ragged query count included."""
    from implicit_amd.als import AlternatingLeastSquares
    from implicit_amd.gpu import local_comm, sharded

    _, _, X2, Y2 = unsharded
    C = sp.vstack([grid_shards(r, GRID, USERS, ITEMS, NNZ, GRID, gamma=2.0, seed=11)[0] for r in range(GRID)]).tocsr()
    users = np.arange(5, USERS, 23)[:1003]  # 1003 queries over 8 ranks: 126 / 125 per rank
    liked = C[users]
    banned = [0, 1, 2, 17]

    def build():
        m = AlternatingLeastSquares(factors=F, use_gpu=True)
        m.user_factors, m.item_factors = gpu.Matrix(X2), gpu.Matrix(Y2)
        return m

    one = build()
    want = one.recommend(users, liked, N=10, filter_items=banned)
    want_sim = one.similar_items(np.arange(0, ITEMS, 3), N=20)

    def rank_body(comm):
        model = build()  # every rank its own replica of the factors, as after fit_sharded
        return (sharded.recommend(model, comm, users, liked, N=10, filter_items=banned),
                sharded.similar_items(model, comm, np.arange(0, ITEMS, 3), N=20))

    out = local_comm.run(8, rank_body, gpu=gpu, oversubscribe=0)
    rec, sim = [o[0] for o in out], [o[1] for o in out]
    assert [r[0] for r in rec[1:]] == [r[1] for r in rec[:-1]] and rec[0][0] == 0 and rec[-1][1] == len(users)
    np.testing.assert_array_equal(np.vstack([r[2] for r in rec]), want[0])
    np.testing.assert_array_equal(np.vstack([r[3] for r in rec]), want[1])
    np.testing.assert_array_equal(np.vstack([s[2] for s in sim]), want_sim[0])
    np.testing.assert_array_equal(np.vstack([s[3] for s in sim]), want_sim[1])
