"""N > 1 path on CPU: world_size-2 `gloo` run of implicit_amd.gpu.sharded.iteration with the oracle as
the per-shard solver (tests may use the oracle; the product never does).  Checks the shard plan, the
view/offset bookkeeping and the exchange order: after each iteration every rank must hold the same
full X, Y and they must equal a single-process oracle fit of the global matrix (only the gramian's
cross-rank summation order differs)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyBackend:
    """Per-shard solver for the CPU test: the oracle's CG on numpy arrays (views share storage)."""

    def __init__(self, oracle):
        self.o = oracle

    def calculate_yty(self, rows, gram, reg):
        f = rows.shape[1]
        gram[...] = self.o.gramian(np.ascontiguousarray(rows)) + np.float32(reg) * np.eye(f, dtype=np.float32)

    def least_squares(self, C, X_rows, gram, Y, cg_steps):
        assert X_rows.flags.c_contiguous and C.shape[0] == X_rows.shape[0]
        self.o.least_squares_cg(C, X_rows, Y, 0.0, cg_steps=cg_steps, YtY=gram)

    @staticmethod
    def rows(M, start, stop):
        return M[int(start):int(stop)]

    @staticmethod
    def upload(array):
        return np.ascontiguousarray(array, dtype=np.float32).copy()

    @staticmethod
    def download(M):
        return M

    def deferred(self, on):
        pass

    def fence(self):
        pass


class GlooComm:
    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch
        self.rank, self.nranks = dist.get_rank(), dist.get_world_size()

    def allreduce_sum(self, M):
        t = self.torch.from_numpy(M)
        self.dist.all_reduce(t)

    def allgather_rows(self, M, offs):
        for r in range(self.nranks):
            part = self.torch.from_numpy(M[int(offs[r]):int(offs[r + 1])])
            self.dist.broadcast(part, src=r)

    def allgather_rows_begin(self, M, lo, hi):  # the pipelined form, executed eagerly here
        for r in range(self.nranks):
            part = self.torch.from_numpy(M[int(lo[r]):int(hi[r])])
            self.dist.broadcast(part, src=r)

    def allgather_rows_end(self):
        pass

    def alltoall_rows(self, send, send_lo, send_hi, recv, recv_lo, recv_hi):
        for src in range(self.nranks):
            for dst in range(self.nranks):
                if src == dst == self.rank:
                    recv[int(recv_lo[src]):int(recv_hi[src])] = send[int(send_lo[dst]):int(send_hi[dst])]
                elif self.rank == src and src != dst and send_hi[dst] > send_lo[dst]:
                    self.dist.send(self.torch.from_numpy(send[int(send_lo[dst]):int(send_hi[dst])]), dst=dst)
                elif self.rank == dst and src != dst and recv_hi[src] > recv_lo[src]:
                    self.dist.recv(self.torch.from_numpy(recv[int(recv_lo[src]):int(recv_hi[src])]), src=src)

    def barrier(self):
        self.dist.barrier()


def _worker(rank, world, port, out_dir, chunks):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
    import torch
    import torch.distributed as dist

    from implicit_amd.gpu import sharded
    from implicit_amd.synthetic import synthetic_csr
    from oracle import oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    f, reg = 32, 0.05
    C = synthetic_csr(600, 400, 12_000, seed=8, neg_frac=0.05, empty_frac=0.02)
    Ct = C.T.tocsr()
    lens_u, lens_i = np.diff(C.indptr), np.diff(Ct.indptr)
    u_off = sharded.shard_offsets(C.shape[0], world, weights=lens_u)
    i_off = sharded.shard_offsets(Ct.shape[0], world, weights=lens_i)
    rng = np.random.default_rng(3)
    X = rng.random((600, f), dtype=np.float32) * 0.1 - 0.05
    Y = rng.random((400, f), dtype=np.float32) * 0.1 - 0.05
    Cui_shard = C[u_off[rank]:u_off[rank + 1]]
    Ciu_shard = Ct[i_off[rank]:i_off[rank + 1]]
    if chunks > 1:  # pipelined exchange: K row chunks per half sweep
        Cui_shard, Ciu_shard = sharded.split_rows(Cui_shard, chunks), sharded.split_rows(Ciu_shard, chunks)
    gram = np.zeros((f, f), dtype=np.float32)
    comm = GlooComm(dist, torch)
    backend = NumpyBackend(oracle)
    for _ in range(3):
        sharded.iteration(backend, comm, Cui_shard, Ciu_shard, X, Y, u_off, i_off, gram, reg, 3)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), X=X, Y=Y, u_off=u_off, i_off=i_off)
    dist.barrier()
    dist.destroy_process_group()


class _Model:
    """What fit_sharded reads of a model."""

    def __init__(self, X, Y, factors, iterations):
        self.user_factors, self.item_factors, self.factors, self.iterations = X, Y, factors, iterations
        self.regularization, self.cg_steps = 0.05, 3


def _worker_from_user_blocks(rank, world, port, out_dir):
    """fit_sharded from THIS RANK's block of user rows only: the item-side shard comes out of shard_transpose (local
    transpose of the block + one personalised exchange), no rank holds or transposes the full matrix."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
    import torch
    import torch.distributed as dist

    from implicit_amd.gpu import sharded
    from implicit_amd.synthetic import synthetic_csr
    from oracle import oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    f = 32
    C = synthetic_csr(700, 300, 14_000, seed=21, neg_frac=0.05, empty_frac=0.03)
    cut = [0, 260, 700][rank:rank + 2]  # uneven blocks, chosen by the caller
    block = C[cut[0]:cut[1]]
    comm, backend = GlooComm(dist, torch), NumpyBackend(oracle)
    Ciu_rows, u_off, i_off = sharded.shard_transpose(comm, backend, block)
    want = C.T.tocsr()[int(i_off[rank]):int(i_off[rank + 1])]
    want.sort_indices()
    assert list(u_off) == [0, 260, 700] and Ciu_rows.shape == want.shape
    np.testing.assert_array_equal(Ciu_rows.indptr, want.indptr)
    np.testing.assert_array_equal(Ciu_rows.indices, want.indices)
    np.testing.assert_array_equal(Ciu_rows.data, want.data)
    rng = np.random.default_rng(3)
    X = rng.random((700, f), dtype=np.float32) * 0.1 - 0.05
    Y = rng.random((300, f), dtype=np.float32) * 0.1 - 0.05
    model = _Model(X, Y, f, 2)
    sharded.fit_sharded(model, block, comm, chunks=2, backend=backend, csr=lambda c: c)
    np.savez(os.path.join(out_dir, f"blocks_rank{rank}.npz"), X=X, Y=Y)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fit_from_user_blocks_only(tmp_path, oracle):
    import torch.multiprocessing as mp

    from implicit_amd.synthetic import synthetic_csr

    world = 2
    mp.spawn(_worker_from_user_blocks, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "blocks_rank0.npz"), np.load(tmp_path / "blocks_rank1.npz")
    np.testing.assert_array_equal(r0["X"], r1["X"])
    np.testing.assert_array_equal(r0["Y"], r1["Y"])
    C = synthetic_csr(700, 300, 14_000, seed=21, neg_frac=0.05, empty_frac=0.03)
    rng = np.random.default_rng(3)
    X = rng.random((700, 32), dtype=np.float32) * 0.1 - 0.05
    Y = rng.random((300, 32), dtype=np.float32) * 0.1 - 0.05
    Xs, Ys = oracle.fit(C, 32, regularization=0.05, iterations=2, user_factors=X, item_factors=Y)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    # the gramian is summed rank by rank here and row by row in the single process: fp32 association noise only
    assert rel(r0["X"], Xs) < 5e-5 and rel(r0["Y"], Ys) < 5e-5


def test_allreduce_ints_is_exact_beyond_fp32():
    from implicit_amd.gpu import sharded

    class OneRank:
        nranks, rank = 1, 0

        @staticmethod
        def allreduce_sum(M):
            pass

    v = np.array([0, 1, 2**24 + 1, 2**31 + 5, 2**40 + 12345], dtype=np.int64)
    np.testing.assert_array_equal(sharded.allreduce_ints(OneRank, NumpyBackend(None), v), v)


def _worker_grid(rank, world, port, out_dir):
    """BASELINE configs[3] in miniature: every rank generates ONLY its own user rows and item rows of the block-composed
    matrix (implicit_amd.synthetic.grid_shards), as bench.py --gpus N does."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
    import torch
    import torch.distributed as dist

    from implicit_amd.gpu import sharded
    from implicit_amd.synthetic import grid_shards
    from oracle import oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    users, items, nnz, grid, f, reg = 2000, 240, 20_000, 4, 32, 0.05
    Cui_shard, Ciu_shard, u_off, i_off = grid_shards(rank, world, users, items, nnz, grid, gamma=2.0, seed=11)
    assert Cui_shard.shape == (u_off[rank + 1] - u_off[rank], items) and Ciu_shard.shape == (i_off[rank + 1] - i_off[rank], users)
    rng = np.random.default_rng(3)
    X = rng.random((users, f), dtype=np.float32) * 0.1 - 0.05
    Y = rng.random((items, f), dtype=np.float32) * 0.1 - 0.05
    gram = np.zeros((f, f), dtype=np.float32)
    comm, backend = GlooComm(dist, torch), NumpyBackend(oracle)
    Cu, Ci = sharded.split_rows(Cui_shard, 2), sharded.split_rows(Ciu_shard, 2)
    for _ in range(2):
        sharded.iteration(backend, comm, Cu, Ci, X, Y, u_off, i_off, gram, reg, 3)
    np.savez(os.path.join(out_dir, f"grid_rank{rank}.npz"), X=X, Y=Y)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_on_a_config4_shaped_miniature(tmp_path, oracle):
    import scipy.sparse as sp
    import torch.multiprocessing as mp

    from implicit_amd.synthetic import grid_shards

    world = 2
    mp.spawn(_worker_grid, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "grid_rank0.npz"), np.load(tmp_path / "grid_rank1.npz")
    np.testing.assert_array_equal(r0["X"], r1["X"])
    np.testing.assert_array_equal(r0["Y"], r1["Y"])
    C = sp.vstack([grid_shards(r, world, 2000, 240, 20_000, 4, gamma=2.0, seed=11)[0] for r in range(world)]).tocsr()
    rng = np.random.default_rng(3)
    X = rng.random((2000, 32), dtype=np.float32) * 0.1 - 0.05
    Y = rng.random((240, 32), dtype=np.float32) * 0.1 - 0.05
    Xs, Ys = oracle.fit(C, 32, regularization=0.05, iterations=2, user_factors=X, item_factors=Y)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(r0["X"], Xs) < 1e-5 and rel(r0["Y"], Ys) < 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("chunks", [1, 3])
def test_two_rank_gloo_matches_single_process(tmp_path, oracle, chunks):
    import torch.multiprocessing as mp

    from implicit_amd.synthetic import synthetic_csr

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), chunks), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["X"], r1["X"])  # replicas identical bit for bit
    np.testing.assert_array_equal(r0["Y"], r1["Y"])
    assert 0 < r0["u_off"][1] < 600 and 0 < r0["i_off"][1] < 400

    C = synthetic_csr(600, 400, 12_000, seed=8, neg_frac=0.05, empty_frac=0.02)
    rng = np.random.default_rng(3)
    X = rng.random((600, 32), dtype=np.float32) * 0.1 - 0.05
    Y = rng.random((400, 32), dtype=np.float32) * 0.1 - 0.05
    Xs, Ys = oracle.fit(C, 32, regularization=0.05, iterations=3, user_factors=X, item_factors=Y)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(r0["X"], Xs) < 1e-5 and rel(r0["Y"], Ys) < 1e-5


def test_chunk_offsets_cover_every_shard():
    from implicit_amd.gpu.sharded import chunk_offsets

    offs = np.array([0, 10, 10, 25], dtype=np.int64)  # an empty shard in the middle
    cuts = chunk_offsets(offs, 4)
    assert cuts.shape == (3, 5)
    assert (cuts[:, 0] == offs[:-1]).all() and (cuts[:, -1] == offs[1:]).all()
    assert (np.diff(cuts, axis=1) >= 0).all()
    assert list(cuts[2]) == [10, 13, 17, 21, 25] or (np.diff(cuts[2]) >= 0).all()
    # equal shares on request; by default the shares FALL (the last chunk's exchange is the exposed one)
    assert list(chunk_offsets(offs, 4, ratio=1.0)[2]) == [10, 13, 17, 21, 25]
    sizes = np.diff(chunk_offsets(np.array([0, 4000]), 4)[0])
    assert (np.diff(sizes) < 0).all() and sizes.sum() == 4000 and sizes[-1] < 0.16 * 4000


def test_chunk_count_and_projection_model(monkeypatch):
    from implicit_amd.gpu import sharded

    monkeypatch.delenv("IMP_SHARD_CHUNKS", raising=False)
    assert [sharded.default_chunks(n) for n in (1, 2, 4, 8)] == [1, 4, 4, 6]
    monkeypatch.setenv("IMP_SHARD_CHUNKS", "3")
    assert sharded.default_chunks(8) == 3
    monkeypatch.delenv("IMP_SHARD_CHUNKS")
    f = sharded.chunk_fractions(6)
    assert f[0] == 0.0 and abs(f[-1] - 1.0) < 1e-12 and (np.diff(np.diff(f)) < 0).all()
    # one rank: no exchange, no resident RCCL kernels
    ms, hidden = sharded.project_iteration_ms([8.0, 8.0], [0.0, 0.0], 1)
    assert ms == hidden == 16.0
    # 8 ranks: never better than the hidden-exchange bound, worse with slower links, and the exposed part is the last chunk
    slow, hid = sharded.project_iteration_ms([8.0, 8.0], [4.5e9, 0.45e9], 8, link_GBps=25.0)
    fast, _ = sharded.project_iteration_ms([8.0, 8.0], [4.5e9, 0.45e9], 8, link_GBps=100.0)
    assert hid == 16.0 and slow > fast > hid
    k1, _ = sharded.project_iteration_ms([8.0, 8.0], [4.5e9, 0.45e9], 8, chunks=1, link_GBps=50.0)
    k6, _ = sharded.project_iteration_ms([8.0, 8.0], [4.5e9, 0.45e9], 8, chunks=6, link_GBps=50.0)
    assert k6 < k1


def _worker_scheme_a(rank, world, port, out_dir):
    """north_star's literal split (SURVEY 8e scheme A): users sharded only; the item half sweep as a replicated-state CG with an
    all-reduce of the I x f partial sums per pass."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from implicit_amd.gpu import sharded
    from implicit_amd.synthetic import synthetic_csr
    from oracle import oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    f, reg = 32, 0.05
    C = synthetic_csr(600, 400, 12_000, seed=8, neg_frac=0.05, empty_frac=0.02)
    C = C.tolil()
    C[:, 7] = 0  # an item nobody touched: its row of Y must come out zero
    C = C.tocsr()
    C.eliminate_zeros()
    u_off = sharded.shard_offsets(C.shape[0], world, weights=np.diff(C.indptr))
    block = C[int(u_off[rank]):int(u_off[rank + 1])]
    Ciu_local = block.T.tocsr()  # every item, local users only
    rng = np.random.default_rng(3)
    X = rng.random((600, f), dtype=np.float32) * 0.1 - 0.05
    Y = rng.random((400, f), dtype=np.float32) * 0.1 - 0.05
    comm, backend = GlooComm(dist, torch), NumpyBackend(oracle)
    X_mine = X[int(u_off[rank]):int(u_off[rank + 1])]
    gram = np.zeros((f, f), dtype=np.float32)
    backend.calculate_yty(X_mine, gram, reg if rank == 0 else 0.0)
    comm.allreduce_sum(gram)
    nnz_i = sharded.allreduce_ints(comm, backend, np.diff(Ciu_local.indptr))
    sharded.scheme_a_item_half_sweep(comm, Ciu_local, X_mine, Y, gram, 3, nnz_i)
    np.savez(os.path.join(out_dir, f"schemeA_rank{rank}.npz"), Y=Y)
    dist.barrier()
    dist.destroy_process_group()


def test_scheme_a_item_half_sweep_matches_the_oracle(tmp_path, oracle):
    import torch.multiprocessing as mp

    from implicit_amd.synthetic import synthetic_csr

    world = 2
    mp.spawn(_worker_scheme_a, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "schemeA_rank0.npz"), np.load(tmp_path / "schemeA_rank1.npz")
    np.testing.assert_array_equal(r0["Y"], r1["Y"])  # replicated state: identical bit for bit
    C = synthetic_csr(600, 400, 12_000, seed=8, neg_frac=0.05, empty_frac=0.02).tolil()
    C[:, 7] = 0
    C = C.tocsr()
    C.eliminate_zeros()
    rng = np.random.default_rng(3)
    X = rng.random((600, 32), dtype=np.float32) * 0.1 - 0.05
    Y = rng.random((400, 32), dtype=np.float32) * 0.1 - 0.05
    oracle.least_squares_cg(C.T.tocsr(), Y, X, 0.05, cg_steps=3)
    assert not Y[7].any() and not r0["Y"][7].any()
    rel = np.linalg.norm(r0["Y"] - Y) / np.linalg.norm(Y)
    assert rel < 1e-5, rel


def test_shard_offsets_balance_by_weight():
    from implicit_amd.gpu.sharded import shard_offsets

    assert list(shard_offsets(10, 3)) == [0, 4, 7, 10]
    w = np.array([100, 1, 1, 1, 1, 1, 1, 1, 1, 92])
    offs = shard_offsets(10, 2, weights=w)
    assert offs[0] == 0 and offs[-1] == 10 and 1 <= offs[1] <= 9
    left, right = w[:offs[1]].sum(), w[offs[1]:].sum()
    assert abs(left - right) <= 100
    offs = shard_offsets(3, 8)  # more ranks than rows: empty shards are legal
    assert offs[0] == 0 and offs[-1] == 3 and (np.diff(offs) >= 0).all()


# ---- sharded inference: queries split over the ranks, no collective ------------------------------------------------------


class _OracleModel:
    """recommend / similar_items of the reference's model layer on the oracle's top-k (the product's model needs a GPU)."""

    def __init__(self, oracle, X, Y):
        self.o, self.X, self.Y = oracle, X, Y

    def recommend(self, userid, user_items, N=10, filter_already_liked_items=True):
        return self.o.topk(self.Y, self.X[userid], N, filter_query_items=user_items if filter_already_liked_items else None)

    def similar_items(self, itemid, N=10):
        norms = np.linalg.norm(self.Y, axis=1).astype(np.float32)
        ids, d = self.o.topk(self.Y, self.Y[itemid], N, item_norms=norms)
        return ids, d / norms[itemid][:, None]


def _recommend_problem():
    from implicit_amd.synthetic import synthetic_csr

    rng = np.random.default_rng(5)
    X = rng.standard_normal((300, 24)).astype(np.float32)
    Y = rng.standard_normal((500, 24)).astype(np.float32)
    C = synthetic_csr(300, 500, 6000, seed=4)
    users = np.arange(3, 300, 7)   # 43 queries: the two ranks get 22 and 21
    return X, Y, C, users


def _worker_recommend(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from implicit_amd.gpu import sharded
    from oracle import oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    X, Y, C, users = _recommend_problem()
    comm = GlooComm(dist, torch)
    model = _OracleModel(oracle, X, Y)
    lo, hi, ids, scores = sharded.recommend(model, comm, users, C[users], N=7)
    slo, shi, sids, sscores = sharded.similar_items(model, comm, np.arange(0, 500, 11), N=5)
    elo, ehi, eids, _ = sharded.recommend(model, comm, users[:1], C[users[:1]], N=7)  # fewer queries than ranks
    np.savez(os.path.join(out_dir, f"rec_rank{rank}.npz"), lo=lo, hi=hi, ids=ids, scores=scores, slo=slo, shi=shi, sids=sids,
             sscores=sscores, elo=elo, ehi=ehi, eids=eids)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_recommend_is_the_single_process_result_cut_in_two(tmp_path, oracle):
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_worker_recommend, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rec_rank{k}.npz") for k in range(world)]
    X, Y, C, users = _recommend_problem()
    model = _OracleModel(oracle, X, Y)
    want_ids, want_scores = model.recommend(users, C[users], N=7)
    assert (int(r[0]["lo"]), int(r[0]["hi"]), int(r[1]["lo"]), int(r[1]["hi"])) == (0, 22, 22, 43)
    np.testing.assert_array_equal(np.vstack([r[0]["ids"], r[1]["ids"]]), want_ids)
    np.testing.assert_array_equal(np.vstack([r[0]["scores"], r[1]["scores"]]), want_scores)
    want_sids, want_ss = model.similar_items(np.arange(0, 500, 11), N=5)
    np.testing.assert_array_equal(np.vstack([r[0]["sids"], r[1]["sids"]]), want_sids)
    np.testing.assert_array_equal(np.vstack([r[0]["sscores"], r[1]["sscores"]]), want_ss)
    # one query, two ranks: rank 0 answers, rank 1 returns an empty slice of the right shape
    assert (int(r[0]["elo"]), int(r[0]["ehi"]), int(r[1]["elo"]), int(r[1]["ehi"])) == (0, 1, 1, 1)
    assert r[0]["eids"].shape == (1, 7) and r[1]["eids"].shape == (0, 7)
