"""Multi-GPU ALS: one process per GPU, rows sharded, RCCL over xGMI for the exchange.

No counterpart exists in the reference ("TODO: multi-gpu support", implicit/gpu/als.cu:169).  The
scheme keeps every per-row solve byte-for-byte the single-GPU kernel (multi-GPU):

  * users AND items are cut into contiguous shards, one per rank; rank g holds the CSR rows of its
    users (global item ids) and of its items (global user ids), plus full replicas of X and Y in HBM
    (C4: 5.1 GB + 0.5 GB of 288 GB);
  * user half sweep: every rank computes the partial gramian of ITS item rows, the f x f partials
    are all-reduced (RCCL, 64 KiB at f=128), then it solves its users against the replica of Y and
    the updated X rows are all-gathered;
  * item half sweep: the same with the roles swapped.

Per iteration and rank: two f x f all-reduces and two all-gathers moving (N-1)/N of X and of Y.
The only arithmetic difference from one GPU is the summation order of the gramian across ranks.

The driver below is written against two small interfaces so that its logic (shard plan, exchange
order, views) is exercised on CPU by tests/test_sharded_gloo.py with a gloo communicator and the
oracle as the per-shard solver:

  comm    : .nranks .rank .allreduce_sum(M) .allgather_rows(M, row_offsets) .barrier()
            .allgather_rows_begin(M, row_lo, row_hi) .allgather_rows_end()   (pipelined form)
  backend : .calculate_yty(F_rows, gram, reg) .least_squares(C, X_rows, gram, Y, cg_steps)
            .rows(M, start, stop) -> view sharing storage
"""
import time

import numpy as np


def shard_offsets(n_rows, nranks, weights=None):
    """Contiguous row ranges, one per rank.  With `weights` (e.g. nnz per row) the cut points balance
    the cumulative weight instead of the row count (partition by nnz)."""
    if weights is None:
        base, extra = divmod(n_rows, nranks)
        sizes = [base + (1 if r < extra else 0) for r in range(nranks)]
        return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(np.asarray(weights, dtype=np.float64))])
    targets = csum[-1] * np.arange(1, nranks) / nranks
    cuts = np.searchsorted(csum, targets, side="left")
    offs = np.concatenate([[0], cuts, [n_rows]]).astype(np.int64)
    return np.maximum.accumulate(offs)


class GpuBackend:
    """The real thing: implicit_amd.gpu objects (HIP kernels through the C-ABI)."""

    def __init__(self, gpu):
        self.gpu = gpu
        self.solver = gpu.LeastSquaresSolver()

    def calculate_yty(self, rows, gram, reg):
        self.solver.calculate_yty(rows, gram, reg)

    def least_squares(self, C, X_rows, gram, Y, cg_steps):
        self.solver.least_squares(C, X_rows, gram, Y, cg_steps)

    @staticmethod
    def rows(M, start, stop):
        return M[int(start):int(stop)]


def chunk_offsets(offsets, chunks):
    """Cut every rank's row range [offsets[r], offsets[r+1]) into `chunks` pieces by row count; every rank computes
    the same (nranks, chunks+1) table, which is what keeps the grouped send/recv of a chunk matched across ranks."""
    offsets = np.asarray(offsets, dtype=np.int64)
    lens = np.diff(offsets)
    k = np.arange(chunks + 1, dtype=np.int64)
    return offsets[:-1, None] + (lens[:, None] * k[None, :]) // chunks


def split_rows(C_shard, chunks):
    """This rank's CSR rows cut the same way (scipy CSR in, list of scipy CSR out)."""
    cuts = chunk_offsets([0, C_shard.shape[0]], chunks)[0]
    return [C_shard[int(cuts[k]):int(cuts[k + 1])] for k in range(chunks)]


def half_sweep(backend, comm, C_shard, X_full, x_offsets, Y_full, y_offsets, gram, reg, cg_steps):
    """Solve this rank's rows of X given the replica of Y; leaves every rank with the full new X.

    `C_shard` is either one CSR handle (solve, then one blocking all-gather) or a list of K handles for the row
    chunks of `split_rows`: chunk k's freshly solved rows are queued for exchange (RCCL on a second stream) while
    chunk k+1 is being solved; the half sweep ends by ordering all exchanges before the next kernels."""
    r = comm.rank
    y_mine = backend.rows(Y_full, y_offsets[r], y_offsets[r + 1])
    # regularisation is added exactly once across the ranks
    backend.calculate_yty(y_mine, gram, reg if r == 0 else 0.0)
    comm.allreduce_sum(gram)
    if not isinstance(C_shard, (list, tuple)):
        x_mine = backend.rows(X_full, x_offsets[r], x_offsets[r + 1])
        backend.least_squares(C_shard, x_mine, gram, Y_full, cg_steps)
        comm.allgather_rows(X_full, x_offsets)
        return
    cuts = chunk_offsets(x_offsets, len(C_shard))
    for k, C_k in enumerate(C_shard):
        x_k = backend.rows(X_full, cuts[r, k], cuts[r, k + 1])
        backend.least_squares(C_k, x_k, gram, Y_full, cg_steps)
        comm.allgather_rows_begin(X_full, cuts[:, k], cuts[:, k + 1])
    comm.allgather_rows_end()


def iteration(backend, comm, Cui_shard, Ciu_shard, X_full, Y_full, u_offsets, i_offsets, gram, reg, cg_steps):
    half_sweep(backend, comm, Cui_shard, X_full, u_offsets, Y_full, i_offsets, gram, reg, cg_steps)
    half_sweep(backend, comm, Ciu_shard, Y_full, i_offsets, X_full, u_offsets, gram, reg, cg_steps)


# ---- synthetic weak-scaling workload + benchmark driver (bench.py --gpus N) -----------------------------


def weak_scaling_shards(rank, nranks, users, items, nnz, gamma, seed=42):
    """Rank `rank`'s pieces of the global (nranks*users) x (nranks*items) matrix whose user block s is
    synthetic_csr(users, nranks*items, nnz, seed=seed+s).  Every rank regenerates all blocks (cheap,
    deterministic, no host-side exchange) and keeps (a) its own user block as CSR with global item
    ids and (b) the columns of its item range from every block, transposed, as CSR with global user
    ids."""
    import scipy.sparse as sp

    from ..synthetic import synthetic_csr

    total_items = nranks * items
    i0, i1 = rank * items, (rank + 1) * items
    pieces, mine, total_nnz = [], None, 0
    for s in range(nranks):
        block = synthetic_csr(users, total_items, nnz, gamma=gamma, seed=seed + s)
        total_nnz += block.nnz
        if s == rank:
            mine = block
        pieces.append(block[:, i0:i1].T.tocsr())  # items x users-of-block-s
    item_rows = sp.hstack(pieces, format="csr").astype(np.float32)
    item_rows.sort_indices()
    if item_rows.indices.dtype != np.int32:
        item_rows.indices = item_rows.indices.astype(np.int32)
        item_rows.indptr = item_rows.indptr.astype(np.int32)
    return mine, item_rows, total_nnz


def bench(args, gpu, users, items, nnz_target, gamma, factors, reg, cg_steps, roofline_fn=None):
    """Weak-scaling benchmark body for WORLD_SIZE > 1 (launched by torch.distributed.run).  torch is
    used ONLY for rendezvous (RCCL unique id), the barrier around the timed region and the max over
    ranks; the data path is RCCL inside libimplicit_hip.so."""
    import datetime
    import os

    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
    box = [gpu.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = gpu.Comm(box[0], world, rank)

    t0 = time.time()
    Cui, Ciu, total_nnz = weak_scaling_shards(rank, world, users, items, nnz_target, gamma)
    u_off = np.arange(world + 1, dtype=np.int64) * users
    i_off = np.arange(world + 1, dtype=np.int64) * items
    rng = np.random.default_rng(7)
    X0 = rng.random((world * users, factors), dtype=np.float32) * 0.01  # same seed on every rank: replicas agree
    Y0 = rng.random((world * items, factors), dtype=np.float32) * 0.01
    t_gen = time.time() - t0

    backend = GpuBackend(gpu)
    # K row chunks per half sweep: chunk k is exchanged over xGMI while chunk k+1 is solved (one chunk = blocking form)
    pipelined = world > 1 or os.environ.get("IMP_FORCE_SHARDED")
    chunks = max(1, int(os.environ.get("IMP_SHARD_CHUNKS", "4"))) if pipelined else 1
    if chunks > 1:
        Cui_d = [gpu.CSRMatrix(c) for c in split_rows(Cui, chunks)]
        Ciu_d = [gpu.CSRMatrix(c) for c in split_rows(Ciu, chunks)]
    else:
        Cui_d, Ciu_d = gpu.CSRMatrix(Cui), gpu.CSRMatrix(Ciu)
    X, Y = gpu.Matrix(X0), gpu.Matrix(Y0)
    del X0, Y0
    gram = gpu.Matrix.zeros(factors, factors)

    def step():
        iteration(backend, comm, Cui_d, Ciu_d, X, Y, u_off, i_off, gram, reg, cg_steps)

    def fence():
        gpu.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize(local_rank)
        dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    # event pairs only for the dominant kernel family inside the timed region (they cost stream time), as in bench.py
    timed_filter = "als_cg_team" if factors in (64, 128) else None
    gpu.Profiler.reset()
    gpu.Profiler.enable(True, only=timed_filter)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    gpu.Profiler.enable(False)
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    kernels = {}
    for name in gpu.Profiler.names():
        ms, n = gpu.Profiler.get(name)
        kernels[name] = ms / args.steps
    step_s = elapsed / args.steps
    result = {
        "metric": "ALS user+item updates/sec per iteration (factors=128)",
        "value": world * (users + items) / step_s,
        "unit": "updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * step_s,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"weak scaling: one BASELINE configs[2]-shaped shard per GPU ({args.shape}), "
                        f"global {world * users} users x {world * items} items, ALS CG cg_steps={cg_steps}",
            "users": world * users, "items": world * items, "nnz": int(total_nnz), "factors": factors,
            "regularization": reg, "solver": "cg", "cg_steps": cg_steps,
            "parallelism": f"row-sharded x{world}, RCCL all-reduce(f x f) + all-gather(factor shards) pipelined in "
                           f"{chunks} row chunk(s) per half sweep",
        },
        "nnz_visits_per_s": 2 * int(total_nnz) / step_s,
        "roofline": roofline_fn(Cui, Ciu, {k: gpu.Profiler.get(k) for k in gpu.Profiler.names()}, args.steps)
        if (roofline_fn and rank == 0) else None,
        "kernels_ms_per_step_rank0": kernels,
        "setup_s": {"generate": t_gen},
    }
    dist.barrier()
    dist.destroy_process_group()
    return result
