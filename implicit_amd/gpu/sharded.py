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
            .alltoall_rows(send, send_lo, send_hi, recv, recv_lo, recv_hi)   (set-up: pieces of the transposed shard)
  backend : .calculate_yty(F_rows, gram, reg) .least_squares(C, X_rows, gram, Y, cg_steps)
            .rows(M, start, stop) -> view sharing storage
            .upload(float32 array (n, c)) -> M   .download(M) -> float32 array   (set-up only)
            .deferred(on) .fence()   queue-only solver calls: one host wait per iteration instead of one per call
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
        self._restore_oversub = None
        if nranks > 1:
            self._restore_oversub = gpu.get_oversubscribe()
            gpu.set_oversubscribe(int(os.environ.get("IMP_SHARD_OVERSUB", "4")))

    def close(self):
        """Undo what the constructor changed on the device: later single-GPU calls of the process (fold-ins, other
        models) run with the launch shape they had before."""
        self.deferred(False)
        if self._restore_oversub is not None:
            self.gpu.set_oversubscribe(self._restore_oversub)
            self._restore_oversub = None

    def deferred(self, on):
        self.gpu.set_deferred_sync(on)

    def fence(self):
        """Host wait for everything queued.  A cluster exchange lost since the last one is a warning on stderr, not an error: its rows
        were re-solved on the device by the fix-up kernel (`gpu.fixup_rows()` counts them)."""
        self.gpu.synchronize()

    def upload(self, array):
        return self.gpu.Matrix(np.ascontiguousarray(array, dtype=np.float32))

    @staticmethod
    def download(M):
        return M.to_numpy()

    def calculate_yty(self, rows, gram, reg):
        self.solver.calculate_yty(rows, gram, reg)

    def least_squares(self, C, X_rows, gram, Y, cg_steps):
        self.solver.least_squares(C, X_rows, gram, Y, cg_steps)

    @staticmethod
    def rows(M, start, stop):
        return M[int(start):int(stop)]


def default_chunks(nranks):
    """Row chunks per half sweep of the pipelined exchange.  Only the LAST chunk's exchange is exposed (nothing is left to
    solve beside it), so more ranks -- more bytes per rank to receive -- get more, smaller chunks; every chunk is a full set
    of row-class launches (seven kernels), which bounds K from above.  IMP_SHARD_CHUNKS overrides."""
    import os

    if os.environ.get("IMP_SHARD_CHUNKS"):
        return max(1, int(os.environ["IMP_SHARD_CHUNKS"]))
    if nranks <= 1:
        return 1
    return 4 if nranks <= 4 else 6


def chunk_fractions(chunks, ratio=None):
    """Cumulative row fractions 0 = c_0 < c_1 < ... < c_K = 1 of the K chunks of a half sweep: sizes fall geometrically
    (chunk k+1 = ratio x chunk k, default 0.75, IMP_SHARD_CHUNK_RATIO), so the LAST chunk -- the one whose exchange nothing
    hides -- is the smallest: K = 4 leaves 15 % of the rows in it instead of 25 %, K = 6 leaves 7 % instead of 17 %."""
    import os

    if ratio is None:
        ratio = float(os.environ.get("IMP_SHARD_CHUNK_RATIO", "0.75"))
    ratio = min(1.0, max(0.25, ratio))
    w = ratio ** np.arange(chunks, dtype=np.float64)
    return np.concatenate([[0.0], np.cumsum(w) / w.sum()])


def chunk_offsets(offsets, chunks, ratio=None):
    """Cut every rank's row range [offsets[r], offsets[r+1]) into `chunks` pieces by row count (sizes as chunk_fractions);
    every rank computes the same (nranks, chunks+1) table, which is what keeps the grouped send/recv of a chunk matched
    across ranks."""
    offsets = np.asarray(offsets, dtype=np.int64)
    lens = np.diff(offsets)
    frac = chunk_fractions(chunks, ratio)
    cuts = np.floor(lens[:, None] * frac[None, :] + 1e-9).astype(np.int64)
    cuts[:, 0], cuts[:, -1] = 0, lens
    cuts = np.maximum.accumulate(cuts, axis=1)
    return offsets[:-1, None] + cuts


def split_rows(C_shard, chunks, ratio=None):
    """This rank's CSR rows cut the same way (scipy CSR in, list of scipy CSR out)."""
    cuts = chunk_offsets([0, C_shard.shape[0]], chunks, ratio)[0]
    return [C_shard[int(cuts[k]):int(cuts[k + 1])] for k in range(chunks)]


def project_iteration_ms(half_sweep_compute_ms, half_sweep_recv_bytes, nranks, chunks=None, ratio=None, link_GBps=50.0,
                         resident_rccl_cost=0.15):
    """Model of one sharded iteration's wall time (what bench.py prints beside the optimistic bound): for every half sweep
    the K chunks are solved one after the other -- compute x (1 + resident_rccl_cost) while RCCL's send / recv kernels hold
    part of the device (measured +15 % with a stand-in resident kernel) -- and chunk k's rows travel over the mesh
    while chunk k+1 is solved; an exchange starts when its chunk is solved AND the previous exchange is done; the half sweep
    ends with the last exchange.  xGMI is a full mesh: a rank receives from its N-1 peers at once, one link each, so an
    exchange takes (bytes received from ONE peer) / link rate.  `half_sweep_recv_bytes`: bytes a rank receives per half sweep
    from all peers together.  Returns (modelled ms, ms if every exchange were hidden)."""
    if chunks is None:
        chunks = default_chunks(nranks)
    frac = np.diff(chunk_fractions(chunks, ratio))
    total = hidden = 0.0
    for compute_ms, recv_bytes in zip(half_sweep_compute_ms, half_sweep_recv_bytes):
        per_peer = recv_bytes / max(1, nranks - 1)
        t = xchg_done = 0.0
        for k in range(chunks):
            t += compute_ms * frac[k] * (1.0 + (resident_rccl_cost if k > 0 and nranks > 1 else 0.0))
            xchg_done = max(t, xchg_done) + 1e3 * per_peer * frac[k] / (link_GBps * 1e9)
        total += max(t, xchg_done) if nranks > 1 else t
        hidden += compute_ms
    return total, hidden


def allreduce_ints(comm, backend, values):
    """Exact element-wise sum over the ranks of an array of non-negative integers (< 2^63), through the fp32 sum
    all-reduce: four 16-bit limbs per value, each limb's sum stays below 2^24 for up to 256 ranks."""
    v = np.asarray(values, dtype=np.int64).reshape(-1)
    limbs = np.stack([(v >> (16 * k)) & 0xFFFF for k in range(4)], axis=1).astype(np.float32)
    m = backend.upload(limbs)
    comm.allreduce_sum(m)
    back = np.rint(backend.download(m)).astype(np.int64).reshape(-1, 4)
    return sum(back[:, k] << (16 * k) for k in range(4))


def shard_transpose(comm, backend, Cui_rows, users=None):
    """From user-sharded input to the two shards a rank solves: every rank passes ITS contiguous block of user rows
    (scipy CSR, global item ids; blocks in rank order) and gets back

        Ciu_rows  its block of ITEM rows (global user ids, columns sorted),
        u_off     the user cut points (from the block sizes),
        i_off     the item cut points (balanced by the global nonzeros per item).

    Only the shard is transposed locally (items x my users); the piece of that transpose inside rank p's item range goes
    to p in one personalised exchange (counts per row, columns, values), and since user blocks are in rank order the
    received pieces concatenate row by row into sorted rows.  No rank ever holds the full matrix."""
    import scipy.sparse as sp

    r, n = comm.rank, comm.nranks
    n_local, items = Cui_rows.shape
    sizes = np.zeros(n, dtype=np.int64)
    sizes[r] = n_local
    u_off = np.concatenate([[0], np.cumsum(allreduce_ints(comm, backend, sizes))]).astype(np.int64)
    if users is not None and int(u_off[-1]) != int(users):
        raise ValueError("the user blocks do not add up to the global number of users the caller counted")
    from ..utils import transpose_csr

    T = transpose_csr(Cui_rows)  # items x my users
    T.sort_indices()
    lens = allreduce_ints(comm, backend, np.diff(T.indptr))
    i_off = shard_offsets(items, n, weights=lens)
    # nonzeros of the piece rank q sends to rank p: table[q, p]
    table = np.zeros((n, n), dtype=np.int64)
    table[r] = T.indptr[i_off[1:]] - T.indptr[i_off[:-1]]
    table = allreduce_ints(comm, backend, table).reshape(n, n)
    my_items = int(i_off[r + 1] - i_off[r])
    # one segment per peer: [row counts | columns (global user ids) | values], 4-byte words riding in an fp32 column
    segs, send_lo, send_hi, at = [], [], [], 0
    for p in range(n):
        lo, hi = int(T.indptr[i_off[p]]), int(T.indptr[i_off[p + 1]])
        counts = np.diff(T.indptr[i_off[p]:i_off[p + 1] + 1]).astype(np.int32)
        cols = (T.indices[lo:hi].astype(np.int64) + u_off[r]).astype(np.int32)
        segs += [counts.view(np.float32), cols.view(np.float32), T.data[lo:hi].astype(np.float32, copy=False)]
        send_lo.append(at)
        at += counts.size + 2 * (hi - lo)
        send_hi.append(at)
    send = backend.upload(np.concatenate(segs).reshape(-1, 1) if at else np.zeros((1, 1), np.float32))
    recv_len = [my_items + 2 * int(table[q, r]) for q in range(n)]
    recv_lo = np.concatenate([[0], np.cumsum(recv_len)])[:-1]
    recv_hi = recv_lo + np.asarray(recv_len)
    recv = backend.upload(np.zeros((max(1, int(recv_hi[-1])), 1), dtype=np.float32))
    comm.alltoall_rows(send, send_lo, send_hi, recv, recv_lo, recv_hi)
    words = backend.download(recv).reshape(-1)
    pieces = []
    for q in range(n):
        seg = words[int(recv_lo[q]):int(recv_hi[q])]
        k = int(table[q, r])
        pieces.append((seg[:my_items].view(np.int32).astype(np.int64), seg[my_items:my_items + k].view(np.int32),
                       seg[my_items + k:my_items + 2 * k]))
    row_len = sum(c for c, _, _ in pieces) if pieces else np.zeros(my_items, np.int64)
    indptr = np.concatenate([[0], np.cumsum(row_len)]).astype(np.int64)
    indices = np.empty(int(indptr[-1]), dtype=np.int32)
    data = np.empty(int(indptr[-1]), dtype=np.float32)
    start = indptr[:-1].copy()  # where the next piece's entries of each row go
    for counts, cols, vals in pieces:
        k = cols.size
        if k:
            piece_ptr = np.concatenate([[0], np.cumsum(counts)])[:-1]
            dest = np.repeat(start - piece_ptr, counts) + np.arange(k, dtype=np.int64)
            indices[dest] = cols
            data[dest] = vals
        start += counts
    idx_dtype = np.int32 if indptr[-1] < 2**31 else np.int64
    Ciu_rows = sp.csr_matrix((data, indices, indptr.astype(idx_dtype)), shape=(my_items, int(u_off[-1])))
    return Ciu_rows, u_off, i_off


def take_rank_rows(Cui_full, comm):
    """Convenience for callers that do hold the whole matrix on every rank: this rank's block of user rows, cut by nnz."""
    u_off = shard_offsets(Cui_full.shape[0], comm.nranks, weights=np.diff(Cui_full.indptr))
    return Cui_full[int(u_off[comm.rank]):int(u_off[comm.rank + 1])]


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


# ---- scheme (A): users sharded only, the item half sweep as a replicated-state distributed CG --------------------------------
#
# north_star's literal split (scheme A): rank g holds its users' rows and the transpose of THAT block,
# Ciu_g = (Cui[U_g, :])^T (every item, local users only); Y, r, p live replicated on every rank.  One CG pass = a local
# SPARSE-PARTIAL product (for every item, the sum over its LOCAL users) followed by an all-reduce of an I x f buffer, then the
# dense per-row update, identical on every rank (XtX p from the all-reduced gramian, alpha, beta, the oracle's early exits as
# per-row masks): 1 + cg_steps all-reduces of I x f per iteration instead of one all-gather of X.  What runs on the GPUs is
# scheme (B) above (DESIGN.md section 6 has the byte counts and why); this is the alternative kept ONE KERNEL AWAY: the
# driver, the exchange pattern and the row arithmetic below are complete and tested against the oracle over gloo
# (tests/test_sharded_gloo.py::test_scheme_a_item_half_sweep_matches_the_oracle); `partial` is the one device kernel a GPU
# version needs (the user kernels' gather-dot-axpy body with the output scattered per item instead of reduced per row).


def scheme_a_sparse_partial(Ciu_local, X_mine, V, first):
    """out[i] = sum over the LOCAL users u of item i of w_iu x_u, float32:
       first pass : w = c+ - (|c| - 1) (x_u . v_i)      (implicit/cpu/_als.pyx:190-201, v = the iterate)
       later      : w = (|c| - 1) (x_u . v_i)           (_als.pyx:214-222, v = the search direction)
    numpy statement of the device kernel scheme (A) would need; rows of `Ciu_local` are items, columns local user ids."""
    import scipy.sparse as sp

    C = Ciu_local.tocsr()
    rows = np.repeat(np.arange(C.shape[0]), np.diff(C.indptr))
    c = C.data.astype(np.float32)
    d = np.einsum("ij,ij->i", V[rows].astype(np.float32), X_mine[C.indices].astype(np.float32)).astype(np.float32)
    cm1 = np.abs(c) - np.float32(1.0)
    w = (np.maximum(c, np.float32(0.0)) - cm1 * d) if first else cm1 * d
    W = sp.csr_matrix((w.astype(np.float32), C.indices, C.indptr), shape=C.shape)
    return np.asarray(W @ X_mine, dtype=np.float32)


def scheme_a_item_half_sweep(comm, Ciu_local, X_mine, Y_full, gram, cg_steps, item_nnz_global, partial=scheme_a_sparse_partial):
    """One item half sweep under scheme (A).  `gram` = XtX + reg I, already all-reduced; `item_nnz_global[i]` = nonzeros of
    item i over ALL ranks (rows without any are zeroed, _als.pyx:182-184); Y_full (numpy float32, replicated) is updated in
    place and ends identical on every rank.  1 + cg_steps all-reduces of an I x f float32 buffer through
    `comm.allreduce_sum`."""
    f32 = np.float32
    Y = Y_full
    empty = np.asarray(item_nnz_global) == 0
    S = np.ascontiguousarray(partial(Ciu_local, X_mine, Y, True), dtype=f32)
    comm.allreduce_sum(S)
    r = (S - Y @ gram.T.astype(f32)).astype(f32)          # r = -A0 y + sum (c+ - (|c|-1) y.x) x
    p = r.copy()
    rsold = np.einsum("ij,ij->i", r, r).astype(f32)
    active = (~empty) & (rsold >= f32(1e-20))             # rsold < 1e-20: the row keeps its iterate (_als.pyx:206)
    x = Y.copy()
    for _ in range(cg_steps):
        S = np.ascontiguousarray(partial(Ciu_local, X_mine, p, False), dtype=f32)
        comm.allreduce_sum(S)                             # every rank takes part in every pass, whatever its rows' state
        Ap = (p @ gram.T.astype(f32) + S).astype(f32)
        pAp = np.einsum("ij,ij->i", p, Ap).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            alpha = np.where(active, rsold / pAp, f32(0.0)).astype(f32)
        x = np.where(active[:, None], x + alpha[:, None] * p, x).astype(f32)
        r = np.where(active[:, None], r - alpha[:, None] * Ap, r).astype(f32)
        rsnew = np.einsum("ij,ij->i", r, r).astype(f32)
        cont = active & (rsnew >= f32(1e-20))             # rsnew < 1e-20: break (_als.pyx:235)
        with np.errstate(divide="ignore", invalid="ignore"):
            beta = np.where(cont, rsnew / rsold, f32(0.0)).astype(f32)
        p = np.where(cont[:, None], r + beta[:, None] * p, p).astype(f32)
        rsold = np.where(cont, rsnew, rsold).astype(f32)
        active = cont
    x[empty] = 0.0
    Y[...] = x


# ---- model-level entry: AlternatingLeastSquares(..., comm=...).fit ------------------------------------------------------


def fit_sharded(model, Cui_rows, comm, callback=None, chunks=None, backend=None, csr=None, users=None):
    """The iterations of AlternatingLeastSquares.fit on `comm.nranks` GPUs.  Every rank calls it with ITS block of user
    rows (scipy CSR, all item columns; blocks in rank order -- `take_rank_rows` cuts one out of a full matrix) and with
    model.user_factors / item_factors holding the same full initial factors on every rank.  The item-side shard is built
    by `shard_transpose` (no rank holds or transposes the full matrix); all ranks finish with identical full factors.
    Returns (u_off, i_off)."""
    import time

    if backend is None:
        import implicit_amd.gpu as gpu

        backend, csr = GpuBackend(gpu, solver=model.solver, nranks=comm.nranks), gpu.CSRMatrix
        gram = gpu.Matrix.zeros(model.factors, model.factors)
    else:
        gram = backend.upload(np.zeros((model.factors, model.factors), dtype=np.float32))
    n = comm.nranks
    try:
        mine_i, u_off, i_off = shard_transpose(comm, backend, Cui_rows, users=users)
        X, Y = model.user_factors, model.item_factors
        if X.shape[0] != u_off[-1] or Y.shape[0] != Cui_rows.shape[1]:
            raise ValueError("user_factors / item_factors do not match the global matrix the shards add up to")
        if chunks is None:
            chunks = default_chunks(n)
        if chunks > 1:
            Cu = [csr(c) for c in split_rows(Cui_rows, chunks)]
            Ci = [csr(c) for c in split_rows(mine_i, chunks)]
        else:
            Cu, Ci = csr(Cui_rows), csr(mine_i)
        backend.deferred(True)  # one host wait per iteration: nothing below reads device results on the host
        for it in range(model.iterations):
            t0 = time.time()
            iteration(backend, comm, Cu, Ci, X, Y, u_off, i_off, gram, model.regularization, model.cg_steps)
            backend.fence()
            if callback:
                callback(it, time.time() - t0, None)
        backend.deferred(False)
        comm.barrier()
    finally:
        if hasattr(backend, "close"):
            backend.close()
    return u_off, i_off


# ---- inference on N GPUs: queries shard trivially -----------------------------------------------------
#
# After fit_sharded every rank holds the full user and item factors, so recommend / similar_items need no exchange at all:
# the batch of queries is cut into contiguous slices, one per rank, each rank scores ITS slice against its replica of the
# item factors with the single-GPU scorer (KnnQuery.topk, knn.cu:131-265 semantics) and returns the slice's bounds with the
# results; whoever needs the whole batch concatenates the slices in rank order (they are row-independent, so the result is
# the one-GPU call's bit for bit).  Same arguments as the model's own methods (gpu/matrix_factorization_base.py:34-101).


def query_slice(n_queries, comm):
    """[lo, hi) of this rank's share of a batch of `n_queries` (contiguous, sizes differing by at most one)."""
    offs = shard_offsets(int(n_queries), comm.nranks)
    return int(offs[comm.rank]), int(offs[comm.rank + 1])


def recommend(model, comm, userids, user_items, N=10, **kwargs):
    """This rank's slice of model.recommend(userids, user_items, N, ...): returns (lo, hi, ids, scores) with ids / scores of
    userids[lo:hi].  `user_items` has one row per entry of `userids` (as the model's method wants it) or is None when neither
    filter_already_liked_items nor recalculate_user needs it."""
    userids = np.asarray(userids)
    if userids.ndim != 1:
        raise ValueError("the sharded recommend takes a batch (1-d array) of user ids")
    if user_items is not None and user_items.shape[0] != len(userids):
        raise ValueError("user_items must contain 1 row for every user in userids")
    lo, hi = query_slice(len(userids), comm)
    if hi == lo:
        return lo, hi, np.zeros((0, N), dtype=np.int32), np.zeros((0, N), dtype=np.float32)
    rows = user_items[lo:hi] if user_items is not None else None
    ids, scores = model.recommend(userids[lo:hi], rows, N=N, **kwargs)
    return lo, hi, ids, scores


def similar_items(model, comm, itemids, N=10, **kwargs):
    """This rank's slice of model.similar_items(itemids, N, ...): (lo, hi, ids, scores)."""
    itemids = np.asarray(itemids)
    if itemids.ndim != 1:
        raise ValueError("the sharded similar_items takes a batch (1-d array) of item ids")
    lo, hi = query_slice(len(itemids), comm)
    if hi == lo:
        return lo, hi, np.zeros((0, N), dtype=np.int32), np.zeros((0, N), dtype=np.float32)
    ids, scores = model.similar_items(itemids[lo:hi], N=N, **kwargs)
    return lo, hi, ids, scores


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
    # both communicators (library stream / exchange stream) must connect exactly the ranks the launcher started: N processes
    # that each found only themselves would "scale" as N independent replicas
    ranks_seen = comm.ranks_seen() if hasattr(comm, "ranks_seen") else (world, world)
    if tuple(ranks_seen) != (world, world):
        raise RuntimeError(f"bench.py --gpus {world}: RCCL connects {ranks_seen} ranks (library / exchange communicator), expected {world}")
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
    no_iter_fence = bool(os.environ.get("IMP_SHARD_NO_ITER_FENCE"))
    # K row chunks per half sweep: chunk k is exchanged over xGMI while chunk k+1 is solved (one chunk = blocking form)
    pipelined = world > 1 or os.environ.get("IMP_FORCE_SHARDED")
    chunks = (default_chunks(world) if world > 1 else max(1, int(os.environ.get("IMP_SHARD_CHUNKS", "4")))) if pipelined else 1
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

    backend.deferred(True)  # a whole iteration is queued without a host wait (K chunks, their exchanges, two all-reduces)

    def step():
        iteration(backend, comm, Cui_d, Ciu_d, X, Y, u_off, i_off, gram, reg, cg_steps)
        if not no_iter_fence:
            gpu.synchronize()  # as fit_sharded: one host wait per iteration (reports a timed-out cluster exchange)

    def fence():
        gpu.synchronize()
        comm.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    # event pairs only for the dominant kernel family inside the timed region (they cost stream time), as in bench.py
    timed_filter = "als_cg_half_sweep"
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
    # `als_cg_half_sweep` brackets the kernels of a least_squares call: an umbrella, not a kernel of its own
    compute_ms = float(sum(ms for name, ms in kernels.items() if not name.startswith("rccl") and name != "als_cg_half_sweep"))
    step_s = elapsed / args.steps
    result = {
        "metric": "ALS user+item updates/sec per iteration (factors=128); top-k recs/sec",
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
        "rccl_ranks_seen": list(ranks_seen),
        "nnz_visits_per_s": 2 * int(total_nnz) / step_s,
        "roofline": roofline_fn(Cui, Ciu, timed, args.steps) if (roofline_fn and rank == 0) else None,
        "kernels_ms_per_step_rank0": kernels,
        # how the step divides on rank 0: its own kernels (HIP events) against the wall time of the step; the difference is
        # what the exchange (and launch gaps) left exposed after pipelining
        "rank0_compute_ms_per_step": compute_ms,
        "rank0_exposed_exchange_ms_per_step": max(0.0, 1e3 * step_s - compute_ms),
        "exchange_GB_received_per_rank_per_step": 4.0 * factors * (users_total + items_total) * (world - 1) / world / 1e9,
        # (B) is what runs -- rows of both sides sharded, solved rows all-gathered; (A) is north_star's
        # literal scheme -- users sharded, the item half sweep as a distributed CG with a ring all-reduce of the I x f buffer per
        # pass -- not built (every pass would re-gather X from HBM and wait for a collective that cannot overlap)
        "exchange_schemes_GB_per_rank_per_step": {
            "B_built_allgather_of_solved_rows": 4.0 * factors * (users_total + items_total) * (world - 1) / world / 1e9,
            "A_not_built_allreduce_of_item_factors_per_cg_pass": (1 + cg_steps) * 2.0 * 4.0 * factors * items_total * (world - 1) / world / 1e9},
        "oversubscription": int(os.environ.get("IMP_SHARD_OVERSUB", "4")) if world > 1 else 1,
        "rank0_shard": {"user_rows": int(Cui.shape[0]), "item_rows": int(Ciu.shape[0]), "user_nnz": int(Cui.nnz),
                        "item_nnz": int(Ciu.nnz)},
        "setup_s": {"generate": t_gen},
    }
    fence()
    backend.close()
    return result
