"""The in-process N-rank communicator (implicit_amd/gpu/local_comm.py) on CPU: host stand-ins for the device matrix and
the solver backend (the oracle's CG), the REAL driver (`sharded.fit_sharded`: shard_transpose with its personalised
exchange, chunked half sweeps, all-reduce of the gramian).  N = 4 and 8 logical ranks must end with identical replicas equal
to a single-process oracle fit of the same matrix.  The GPU twin (tests/test_gpu_logical_shards.py) drives the HIP kernels
through the same communicator."""
import numpy as np
import pytest

from test_sharded_gloo import NumpyBackend, _Model


class HostMatrix(np.ndarray):
    """A numpy array with the four Matrix methods the communicator uses (views keep sharing storage)."""

    def to_numpy(self):
        return np.asarray(self)

    def copy_from_numpy(self, a):
        self[...] = a

    def copy_rows_from(self, dst_row, other, src_row, rows):
        self[dst_row:dst_row + rows] = np.asarray(other)[src_row:src_row + rows]


class HostBackend(NumpyBackend):
    @staticmethod
    def upload(array):
        return np.ascontiguousarray(array, dtype=np.float32).copy().view(HostMatrix)

    @staticmethod
    def download(M):
        return np.asarray(M)


class _StubGpu:
    """What local_comm.run touches of the gpu module."""

    occupied = 0

    def get_oversubscribe(self):
        return 1

    def set_oversubscribe(self, n):
        pass

    def set_deferred_sync(self, on):
        pass

    def synchronize(self):
        pass

    def debug_occupy(self, workgroups, microseconds):
        _StubGpu.occupied += 1


@pytest.mark.parametrize("nranks,chunks", [(4, 1), (4, 3), (8, 2)])
def test_logical_ranks_match_the_single_process_fit(oracle, nranks, chunks):
    from implicit_amd.gpu import local_comm, sharded
    from implicit_amd.synthetic import synthetic_csr

    users, items, f = 900, 350, 32
    C = synthetic_csr(users, items, 16_000, seed=31, neg_frac=0.05, empty_frac=0.03)
    cuts = sharded.shard_offsets(users, nranks, weights=np.diff(C.indptr))
    rng = np.random.default_rng(3)
    X0 = rng.random((users, f), dtype=np.float32) * 0.1 - 0.05
    Y0 = rng.random((items, f), dtype=np.float32) * 0.1 - 0.05

    def rank_body(comm):
        block = C[int(cuts[comm.rank]):int(cuts[comm.rank + 1])]
        model = _Model(X0.copy().view(HostMatrix), Y0.copy().view(HostMatrix), f, 2)
        u_off, i_off = sharded.fit_sharded(model, block, comm, chunks=chunks, backend=HostBackend(oracle), csr=lambda c: c)
        return np.asarray(model.user_factors), np.asarray(model.item_factors), u_off, i_off, comm.stats

    _StubGpu.occupied = 0
    out = local_comm.run(nranks, rank_body, gpu=_StubGpu(), occupy=(8, 100))
    for X, Y, u_off, i_off, _ in out[1:]:
        np.testing.assert_array_equal(X, out[0][0])
        np.testing.assert_array_equal(Y, out[0][1])
        np.testing.assert_array_equal(u_off, cuts)
    stats = out[0][4]
    # every rank copied every other rank's rows of both sides in both iterations
    assert stats["allgather_rows_copied"] == 2 * (nranks - 1) * (users + items)
    assert stats["alltoall_rows_copied"] > 0 and _StubGpu.occupied == stats["occupied"] > 0
    Xs, Ys = oracle.fit(C, f, regularization=0.05, iterations=2, user_factors=X0.copy(), item_factors=Y0.copy())
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(out[0][0], Xs) < 5e-5 and rel(out[0][1], Ys) < 5e-5


def test_a_failing_rank_does_not_hang_the_others():
    from implicit_amd.gpu import local_comm

    def rank_body(comm):
        if comm.rank == 2:
            raise RuntimeError("rank 2 gave up")
        comm.barrier()
        return comm.rank

    with pytest.raises(RuntimeError, match="rank 2 gave up"):
        local_comm.run(4, rank_body, gpu=_StubGpu())
