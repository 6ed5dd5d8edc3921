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
(north_star's literal scheme -- users sharded only, the item side as a replicated-state CG with an all-reduce
of an I x f buffer per CG pass -- moves fewer bytes over xGMI at U >> I but re-streams every gathered factor row
1 + cg_steps times from HBM, because no row's nonzeros are local to one GPU any more; DESIGN.md section 6 has the
numbers for BASELINE configs[3].)

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
    """The real thing: implicit_amd.gpu objects (HIP kernels through the C-ABI).

    With more than one rank the exchange of chunk k (RCCL send / recv kernels on the exchange stream) is resident on
    the device while the kernels of chunk k + 1 run.  The row kernels are persistent -- one workgroup per slot the
    device has, each with a fixed share of the rows -- so a workgroup that has to wait for a slot held by RCCL would do
    its whole share after everybody else has finished.  `nranks > 1` therefore launches them 4x oversubscribed
    (imp_set_oversubscribe; IMP_SHARD_OVERSUB overrides): shares a quarter the size, balanced by the hardware
    dispatcher over whatever slots are free (measured with a stand-in resident kernel: profiles/micro/oversub.py)."""

    def __init__(self, gpu, solver=None, nranks=1):
        import os

        self.gpu = gpu
        self.solver = solver or gpu.LeastSquaresSolver()
        if nranks > 1:
            gpu.set_oversubscribe(int(os.environ.get("IMP_SHARD_OVERSUB", "4")))

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


# ---- model-level entry: AlternatingLeastSquares(..., comm=...).fit ------------------------------------------------------


def fit_sharded(model, Cui, Ciu, comm, callback=None, chunks=None):
    """The iterations of AlternatingLeastSquares.fit on `comm.nranks` GPUs.  Every rank calls it with the SAME full
    matrices (scipy CSR, users x items and its transpose) and the same initial factors in model.user_factors /
    item_factors; rows are cut by nnz weight, every rank keeps its own rows of both orientations on the device, and
    all ranks finish with identical full factor matrices."""
    import time

    import implicit_amd.gpu as gpu

    r, n = comm.rank, comm.nranks
    u_off = shard_offsets(Cui.shape[0], n, weights=np.diff(Cui.indptr))
    i_off = shard_offsets(Ciu.shape[0], n, weights=np.diff(Ciu.indptr))
    mine_u, mine_i = Cui[u_off[r]:u_off[r + 1]], Ciu[i_off[r]:i_off[r + 1]]
    if chunks is None:
        chunks = 4 if n > 1 else 1
    if chunks > 1:
        Cu = [gpu.CSRMatrix(c) for c in split_rows(mine_u, chunks)]
        Ci = [gpu.CSRMatrix(c) for c in split_rows(mine_i, chunks)]
    else:
        Cu, Ci = gpu.CSRMatrix(mine_u), gpu.CSRMatrix(mine_i)
    backend = GpuBackend(gpu, solver=model.solver, nranks=n)
    gram = gpu.Matrix.zeros(model.factors, model.factors)
    X, Y = model.user_factors, model.item_factors
    for it in range(model.iterations):
        t0 = time.time()
        iteration(backend, comm, Cu, Ci, X, Y, u_off, i_off, gram, model.regularization, model.cg_steps)
        if callback:
            gpu.synchronize()
            callback(it, time.time() - t0, None)
    gpu.synchronize()
    comm.barrier()


# ---- synthetic workloads + benchmark driver (bench.py --gpus N) -------------------------------------------------------


def rank_sum(comm, gpu, value):
    """Exact sum of a non-negative integer over the ranks through the fp32 all-reduce (split into 16-bit limbs)."""
    limbs = [(int(value) >> (16 * k)) & 0xFFFF for k in range(4)]
    m = gpu.Matrix(np.array([limbs], dtype=np.float32))
    comm.allreduce_sum(m)
    back = m.to_numpy()[0]
    return sum(int(round(float(back[k]))) << (16 * k) for k in range(4))


def rank_max(comm, gpu, value):
    """Max of a float over the ranks: every rank writes its slot of a zero vector, the all-reduce fills in the rest."""
    slots = np.zeros((1, comm.nranks), dtype=np.float32)
    slots[0, comm.rank] = value
    m = gpu.Matrix(slots)
    comm.allreduce_sum(m)
    return float(m.to_numpy().max())


def bench(args, gpu, shapes, factors, reg, cg_steps, roofline_fn=None):
    """Benchmark body for WORLD_SIZE > 1, one process per GPU (launched by torch.distributed.run, whose environment
    variables are all that is used of it: the rendezvous is a TCP hand-off of the RCCL id, barriers and the max over
    ranks go through RCCL).  Default: STRONG scaling on BASELINE configs[3] (10 M users x 1 M items x 500 M nnz,
    f = 128, user- and item-sharded); --weak: one BASELINE configs[2]-shaped shard per GPU."""
    import os

    from ..synthetic import grid_shards
    from . import rendezvous

    rank, world, local_rank = rendezvous.env_world()
    comm = rendezvous.init_comm(gpu, rank, world, local_rank)
    t0 = time.time()
    if args.weak:
        users, items, nnz_target, gamma = shapes[args.shape]
        users_total, items_total, nnz_total, grid = world * users, world * items, world * nnz_target, world
        label = (f"weak scaling: one BASELINE configs[2]-shaped shard per GPU ({args.shape}), global {users_total} users x "
                 f"{items_total} items")
        scaling = "weak"
    else:
        shape = args.shape if args.shape != "lastfm360k" else "c4"
        users_total, items_total, nnz_total, gamma = shapes[shape]
        grid = 8 if 8 % world == 0 else world  # the matrix is a function of the grid, not of the rank count
        label = (f"BASELINE configs[3]: {users_total} users x {items_total} items, synthetic CSR composed of {grid} x {grid} "
                 f"blocks, users and items sharded over {world} GPU(s)")
        scaling = "strong"
    if args.scale != 1.0:
        users_total, items_total, nnz_total = (int(users_total * args.scale), int(items_total * args.scale),
                                               int(nnz_total * args.scale))
    Cui, Ciu, u_off, i_off = grid_shards(rank, world, users_total, items_total, nnz_total, grid, gamma=gamma, seed=42)
    t_gen = time.time() - t0

    backend = GpuBackend(gpu, nranks=world)
    # K row chunks per half sweep: chunk k is exchanged over xGMI while chunk k+1 is solved (one chunk = blocking form)
    pipelined = world > 1 or os.environ.get("IMP_FORCE_SHARDED")
    chunks = max(1, int(os.environ.get("IMP_SHARD_CHUNKS", "4"))) if pipelined else 1
    if chunks > 1:
        Cui_d = [gpu.CSRMatrix(c) for c in split_rows(Cui, chunks)]
        Ciu_d = [gpu.CSRMatrix(c) for c in split_rows(Ciu, chunks)]
    else:
        Cui_d, Ciu_d = gpu.CSRMatrix(Cui), gpu.CSRMatrix(Ciu)
    # replicas drawn on the device from the same Philox seed on every rank (5 GB at configs[3]: no host copy)
    X = gpu.RandomState(7).uniform(users_total, factors, 0.0, 0.01)
    Y = gpu.RandomState(8).uniform(items_total, factors, 0.0, 0.01)
    gram = gpu.Matrix.zeros(factors, factors)
    total_nnz = rank_sum(comm, gpu, Cui.nnz)

    def step():
        iteration(backend, comm, Cui_d, Ciu_d, X, Y, u_off, i_off, gram, reg, cg_steps)

    def fence():
        gpu.synchronize()
        comm.barrier()

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
    timed = {k: gpu.Profiler.get(k) for k in gpu.Profiler.names()}
    elapsed = rank_max(comm, gpu, elapsed)

    # untimed extra iteration with events on everything: where rank 0's time goes (kernels vs exchange)
    gpu.Profiler.reset()
    gpu.Profiler.enable(True)
    step()
    fence()
    gpu.Profiler.enable(False)
    kernels = {name: gpu.Profiler.get(name)[0] for name in gpu.Profiler.names()}
    compute_ms = float(sum(ms for name, ms in kernels.items() if not name.startswith("rccl")))
    step_s = elapsed / args.steps
    result = {
        "metric": "ALS user+item updates/sec per iteration (factors=128)",
        "value": (users_total + items_total) / step_s,
        "unit": "updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * step_s,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{label}, ALS CG cg_steps={cg_steps}" + ("" if args.scale == 1.0 else f" (scaled x{args.scale})"),
            "users": users_total, "items": items_total, "nnz": int(total_nnz), "factors": factors,
            "regularization": reg, "solver": "cg", "cg_steps": cg_steps,
            "parallelism": f"row-sharded x{world}, RCCL all-reduce(f x f) + all-gather(factor shards) pipelined in "
                           f"{chunks} row chunk(s) per half sweep",
        },
        "nnz_visits_per_s": 2 * int(total_nnz) / step_s,
        "roofline": roofline_fn(Cui, Ciu, timed, args.steps) if (roofline_fn and rank == 0) else None,
        "kernels_ms_per_step_rank0": kernels,
        # how the step divides on rank 0: its own kernels (HIP events) against the wall time of the step; the difference is
        # what the exchange (and launch gaps) left exposed after pipelining
        "rank0_compute_ms_per_step": compute_ms,
        "rank0_exposed_exchange_ms_per_step": max(0.0, 1e3 * step_s - compute_ms),
        "exchange_GB_received_per_rank_per_step": 4.0 * factors * (users_total + items_total) * (world - 1) / world / 1e9,
        "oversubscription": int(os.environ.get("IMP_SHARD_OVERSUB", "4")) if world > 1 else 1,
        "rank0_shard": {"user_rows": int(Cui.shape[0]), "item_rows": int(Ciu.shape[0]), "user_nnz": int(Cui.nnz),
                        "item_nnz": int(Ciu.nnz)},
        "setup_s": {"generate": t_gen},
    }
    fence()
    return result
