"""AlternatingLeastSquares on MI355X.

Public surface of implicit/gpu/als.py:14-341 (fit / recalculate_user / recalculate_item /
partial_fit_users / partial_fit_items / YtY / XtX / save / load / pickle, fit_callback) with the
numerics of the reference's *CPU* path, which is the parity oracle (implicit/cpu/als.py:98-202):

  * `alpha` scales the confidence matrix inside fit (cpu/als.py:133-134; the reference GPU fit
    silently ignores alpha -- SURVEY 3.2);
  * factors start as rng.random((n, f), float32) * 0.01, users first then items, from
    numpy.random.default_rng(random_state) (cpu/als.py:144-147) unless `init="device"` asks for the
    reference-GPU style U(-0.5/f, 0.5/f) drawn on the device (gpu/als.py:129-139);
  * `use_cg=False` selects the Cholesky solver (cpu/als.py:418-423), which the reference GPU path
    does not have; recalculate_user/item use Cholesky as the CPU path does (cpu/als.py:221-241)
    (any factors <= 1024; the reference GPU path runs CG to `factors` steps there, gpu/als.py:188-195);
  * NaN factors after fit raise ModelFitError (cpu/als.py:202).

Multi-GPU: `AlternatingLeastSquares(..., comm=implicit_amd.gpu.Comm(...))`, one process per GPU, every rank calling
fit() with ITS contiguous block of user rows (blocks in rank order; implicit_amd/gpu/sharded.py: the item-side shard is
exchanged at set-up, RCCL all-reduce of the gramian and all-gather of the solved factor rows per half sweep; all ranks
end with identical full factors).  `rendezvous.init_comm(gpu)` builds the
communicator from the launcher's RANK / WORLD_SIZE / MASTER_* environment.  No reference counterpart
(implicit/gpu/als.cu:169 "TODO: multi-gpu support").
"""
import logging
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import implicit_amd.gpu as gpu

from ..utils import check_csr, check_random_state, random_factors, transpose_csr
from .matrix_factorization_base import MatrixFactorizationBase

log = logging.getLogger("implicit_amd")

_CHOLESKY_MAX_FACTORS = 1024  # what the library's Cholesky takes (beyond 256: the triangle in a device workspace)


class AlternatingLeastSquares(MatrixFactorizationBase):
    def __init__(self, factors=64, regularization=0.01, alpha=1.0, dtype=np.float32, iterations=15,
                 calculate_training_loss=False, random_state=None, use_cg=True, cg_steps=3, init="numpy", comm=None):
        if not gpu.HAS_CUDA:
            raise ValueError("No usable HIP device / extension, can't train on GPU.")
        super().__init__()
        self.factors = factors
        self.regularization = regularization
        self.alpha = alpha
        self.dtype = np.dtype(dtype)
        self.iterations = iterations
        self.calculate_training_loss = calculate_training_loss
        self.random_state = random_state
        self.use_cg = use_cg
        self.cg_steps = cg_steps
        self.init = init
        self.comm = comm
        self.fit_callback = None
        self._solver = None
        self._YtY = self._XtX = None      # regularised gramians (the reference's cached properties)
        self._YtY0 = self._XtX0 = None    # unregularised ones, for the Cholesky fold-in

    # ---- training ----------------------------------------------------------------------------------
    def _initial_factors(self, users, items):
        if self.init == "device":
            from .matrix_factorization_base import check_random_state as device_rs

            rs = device_rs(self.random_state)
            lo, hi = -0.5 / self.factors, 0.5 / self.factors
            if self.user_factors is None:
                self.user_factors = rs.uniform(users, self.factors, low=lo, high=hi).astype(self.dtype)
            if self.item_factors is None:
                self.item_factors = rs.uniform(items, self.factors, low=lo, high=hi).astype(self.dtype)
            return
        rng = check_random_state(self.random_state)
        if self.user_factors is None:
            x0 = random_factors(rng, users, self.factors)
            self.user_factors = gpu.Matrix(x0.astype(self.dtype, copy=False))
        if self.item_factors is None:
            y0 = random_factors(rng, items, self.factors)
            self.item_factors = gpu.Matrix(y0.astype(self.dtype, copy=False))

    def fit(self, user_items, show_progress=True, callback=None):
        Cui = check_csr(user_items)
        if Cui.dtype != np.float32:
            Cui = Cui.astype(np.float32)
        if self.alpha != 1.0:
            Cui = self.alpha * Cui
        self._item_norms = self._user_norms = None
        self._item_norms_host = self._user_norms_host = None
        self._YtY = self._XtX = self._YtY0 = self._XtX0 = None
        if self.comm is not None and self.comm.nranks > 1:
            return self._fit_sharded(Cui, callback)  # `user_items` = THIS RANK's block of user rows

        # set-up: the transpose (threaded counting transpose in the library, host only, no device lock; scipy for
        # non-canonical input) runs on a worker thread while this one uploads the user-side matrix with its row schedule
        # and draws / uploads the initial factors; ctypes and numpy drop the GIL in all of them
        t0 = time.time()
        users, items = Cui.shape
        with ThreadPoolExecutor(max_workers=1) as worker:
            transposed = worker.submit(transpose_csr, Cui)
            Cui_dev = gpu.CSRMatrix(Cui)
            self._initial_factors(users, items)
            Ciu = transposed.result()
        log.debug("Transpose, user-side upload and initial factors in %.3fs", time.time() - t0)
        Ciu_dev = gpu.CSRMatrix(Ciu)
        X, Y = self.user_factors, self.item_factors
        gram = gpu.Matrix.zeros(self.factors, self.factors)
        loss = None
        progress = _progress(self.iterations, show_progress)
        # the four solver calls of an iteration are queued back to back (deferred mode) and waited for once: no host round
        # trip between a gramian and the sweep that uses it (-1 % per iteration at configs[2]; errors surface at that wait)
        gpu.set_deferred_sync(True)
        fixups_before = gpu.fixup_rows(reset=False)
        try:
            for iteration in range(self.iterations):
                t0 = time.time()
                self._half_sweep(Cui_dev, X, Y, gram)
                self._half_sweep(Ciu_dev, Y, X, gram)
                gpu.synchronize()
                if self.calculate_training_loss:
                    loss = self.solver.calculate_loss(Cui_dev, X, Y, self.regularization)
                    if not show_progress:
                        log.info("loss %.4f", loss)
                progress.update(loss)
                cb = callback or self.fit_callback
                if cb:
                    cb(iteration, time.time() - t0, loss)
        finally:
            gpu.set_deferred_sync(False)
            if self.factors not in (64, 128, 256):
                gpu.release_workspaces()  # the zero-padded factor copies: rows x F floats per side, not worth keeping
        progress.close()
        # rows > 512 nonzeros run fp16-split operands on the matrix cores; an operand that leaves the fp16 range (very large
        # confidences x alpha on un-normalised factors) sends its row to the one-wavefront fp32 kernel instead: correct, and
        # ~1 ms per 4096-nonzero row -- a slowdown nobody would otherwise see
        refits = gpu.fixup_rows(reset=False) - fixups_before
        if refits > max(64, 0.001 * (users + items) * self.iterations):
            log.warning("%d long-row solves of this fit were re-done by the fp32 fix-up kernel (operands beyond the fp16 range of "
                        "the matrix-core path): scale the confidences (alpha) or the factors down, or expect slow iterations", refits)
        if self.calculate_training_loss:
            log.info("Final training loss %s", loss)
        self._check_fit_errors()

    def _fit_sharded(self, Cui_rows, callback):
        """Multi-GPU fit: `Cui_rows` is this rank's contiguous block of user rows (all item columns, blocks in rank
        order; `sharded.take_rank_rows(full, comm)` cuts one out of a full matrix).  Nothing of full-matrix size is
        built on any rank: the item-side shard comes from a transpose of the block plus one personalised exchange."""
        from . import sharded

        if not self.use_cg:
            raise ValueError("the multi-GPU fit runs the CG solver (use_cg=True)")
        if self.dtype != np.float32:
            raise ValueError("the multi-GPU fit keeps float32 factor replicas")
        backend = sharded.GpuBackend(gpu, solver=self.solver, nranks=self.comm.nranks)
        try:  # whatever fails below, the device gets its launch shape and synchronous calls back (backend.close)
            sizes = np.zeros(self.comm.nranks, dtype=np.int64)
            sizes[self.comm.rank] = Cui_rows.shape[0]
            users = int(sharded.allreduce_ints(self.comm, backend, sizes).sum())
            self._initial_factors(users, Cui_rows.shape[1])
            # every rank must start from the same factors whatever its random_state: rank 0's win (the others contribute
            # zeros to a sum all-reduce)
            for m in (self.user_factors, self.item_factors):
                if self.comm.rank != 0:
                    m.copy_from_numpy(np.zeros(m.shape, dtype=np.float32))
                self.comm.allreduce_sum(m)
            u_off, _ = sharded.fit_sharded(self, Cui_rows, self.comm, callback or self.fit_callback, backend=backend,
                                           csr=gpu.CSRMatrix, users=users)
        finally:
            backend.close()
            if self.factors not in (64, 128, 256):
                gpu.release_workspaces()
        if self.calculate_training_loss:
            # the objective restricted to this rank's user rows (each rank logs its own; there is no global reduction)
            r = self.comm.rank
            loss = self.solver.calculate_loss(gpu.CSRMatrix(Cui_rows), self.user_factors[int(u_off[r]):int(u_off[r + 1])],
                                              self.item_factors, self.regularization)
            log.info("Final training loss over rank %d's user rows %s", r, loss)
        self._check_fit_errors()

    def _half_sweep(self, C, X, Y, gram):
        """One half iteration: X <- argmin given Y (gramian + per-row solves)."""
        if self.use_cg:
            self.solver.calculate_yty(Y, gram, self.regularization)
            self.solver.least_squares(C, X, gram, Y, self.cg_steps)
        else:
            self.solver.calculate_yty(Y, gram, 0.0)
            self.solver.least_squares_cholesky(C, X, gram, Y, self.regularization)

    # ---- fold-in -----------------------------------------------------------------------------------
    def _recalculate(self, ids, rows_csr, other_factors, gram_reg_fn, gram_unreg_fn):
        rows_csr = check_csr(rows_csr)
        count = 1 if np.isscalar(ids) else len(ids)
        if rows_csr.shape[0] != count:
            raise ValueError("expected one sparse row for every id to recalculate")
        if self.alpha != 1.0:
            rows_csr = self.alpha * rows_csr
        out = gpu.Matrix.zeros(count, self.factors).astype(self.dtype)
        C = gpu.CSRMatrix(rows_csr.astype(np.float32))
        # only the gramian the chosen solver needs is evaluated, and both are cached until the factors change
        if self.factors <= _CHOLESKY_MAX_FACTORS:
            self.solver.least_squares_cholesky(C, out, gram_unreg_fn(), other_factors, self.regularization)
        else:
            self.solver.least_squares(C, out, gram_reg_fn(), other_factors, self.factors)
        return out[0] if np.isscalar(ids) else out

    def recalculate_user(self, userid, user_items):
        return self._recalculate(userid, user_items, self.item_factors, lambda: self.YtY, lambda: self._YtY_unreg)

    def recalculate_item(self, itemid, item_users):
        return self._recalculate(itemid, item_users, self.user_factors, lambda: self.XtX, lambda: self._XtX_unreg)

    def partial_fit_users(self, userids, user_items):
        if len(userids) != user_items.shape[0]:
            raise ValueError("user_items must contain 1 row for every user in userids")
        new_rows = self.recalculate_user(userids, user_items)
        rows, factors = self.user_factors.shape
        if max(userids) >= rows:
            self.user_factors.resize(max(userids) + 1, factors)
        self.user_factors.assign_rows(userids, new_rows)  # same dtype on both sides (fp16 models: scattered on the device)
        self._user_norms = self._user_norms_host = None
        self._XtX = self._XtX0 = None

    def partial_fit_items(self, itemids, item_users):
        if len(itemids) != item_users.shape[0]:
            raise ValueError("item_users must contain 1 row for every user in itemids")
        new_rows = self.recalculate_item(itemids, item_users)
        rows, factors = self.item_factors.shape
        if max(itemids) >= rows:
            self.item_factors.resize(max(itemids) + 1, factors)
        self.item_factors.assign_rows(itemids, new_rows)
        self._item_norms = self._item_norms_host = None
        self._YtY = self._YtY0 = None

    # ---- cached gramians ------------------------------------------------------------------------------
    @property
    def solver(self):
        if self._solver is None:
            self._solver = gpu.LeastSquaresSolver()
        return self._solver

    def _gram(self, factors, reg):
        out = gpu.Matrix.zeros(self.factors, self.factors)
        self.solver.calculate_yty(factors, out, reg)
        return out

    @property
    def YtY(self):
        if self._YtY is None:
            self._YtY = self._gram(self.item_factors, self.regularization)
        return self._YtY

    @property
    def XtX(self):
        if self._XtX is None:
            self._XtX = self._gram(self.user_factors, self.regularization)
        return self._XtX

    @property
    def _YtY_unreg(self):
        if self._YtY0 is None:
            self._YtY0 = self._gram(self.item_factors, 0.0)
        return self._YtY0

    @property
    def _XtX_unreg(self):
        if self._XtX0 is None:
            self._XtX0 = self._gram(self.user_factors, 0.0)
        return self._XtX0

    def to_cpu(self):
        """implicit/gpu/als.py:300-313: the same model as the reference's CPU class.  This package does not ship a CPU
        model (the reference's is the parity oracle), so stock `implicit` has to be importable."""
        try:
            import implicit.cpu.als as cpu_als
        except ImportError as e:
            raise ImportError("to_cpu() builds implicit.cpu.als.AlternatingLeastSquares: install benfred/implicit for the "
                              "CPU model (implicit_amd ships the MI355X path only)") from e
        ret = cpu_als.AlternatingLeastSquares(factors=self.factors, regularization=self.regularization, alpha=self.alpha,
                                              dtype=np.float32, iterations=self.iterations, use_cg=self.use_cg,
                                              calculate_training_loss=self.calculate_training_loss,
                                              random_state=self.random_state)
        ret.user_factors = None if self.user_factors is None else self.user_factors.to_numpy().astype(np.float32)
        ret.item_factors = None if self.item_factors is None else self.item_factors.to_numpy().astype(np.float32)
        return ret

    # ---- persistence (same .npz keys as implicit/cpu/als.py:458-477 so stock implicit can load it) ----
    def save(self, fileobj_or_path):
        args = {
            "user_factors": None if self.user_factors is None else self.user_factors.to_numpy(),
            "item_factors": None if self.item_factors is None else self.item_factors.to_numpy(),
            "regularization": self.regularization,
            "factors": self.factors,
            "iterations": self.iterations,
            "use_cg": self.use_cg,
            "cg_steps": self.cg_steps,
            "calculate_training_loss": self.calculate_training_loss,
            "dtype": self.dtype.name,
            "random_state": self.random_state if isinstance(self.random_state, (int, np.integer)) else None,
            "alpha": self.alpha,
        }
        np.savez(fileobj_or_path, **{k: v for k, v in args.items() if v is not None})

    @classmethod
    def load(cls, fileobj_or_path):
        model = super().load(fileobj_or_path)
        for name in ("user_factors", "item_factors"):
            value = getattr(model, name, None)
            if isinstance(value, np.ndarray):
                setattr(model, name, gpu.Matrix(np.ascontiguousarray(value)))
        for stale in ("num_threads", "use_native"):  # keys written by the CPU model
            if hasattr(model, stale):
                delattr(model, stale)
        return model

    def __getstate__(self):
        state = super().__getstate__()
        state["_solver"] = None
        state["comm"] = None  # a communicator belongs to a process group, not to a model file
        state["_XtX0"] = state["_YtY0"] = None
        state["_XtX"] = self._XtX.to_numpy() if self._XtX is not None else None
        state["_YtY"] = self._YtY.to_numpy() if self._YtY is not None else None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        if self._XtX is not None:
            self._XtX = gpu.Matrix(self._XtX)
        if self._YtY is not None:
            self._YtY = gpu.Matrix(self._YtY)


def _as_f32(m):
    return m if m.itemsize == 4 else m.astype(np.float32)


class _progress:
    """tqdm when available and asked for, otherwise a no-op."""

    def __init__(self, total, show):
        self.bar = None
        if show:
            try:
                from tqdm.auto import tqdm

                self.bar = tqdm(total=total)
            except ImportError:
                pass

    def update(self, loss):
        if self.bar is not None:
            self.bar.update(1)
            if loss is not None:
                self.bar.set_postfix({"loss": loss})

    def close(self):
        if self.bar is not None:
            self.bar.close()


def calculate_loss(Cui, X, Y, regularization, solver=None):
    """Module-level helper of implicit/gpu/als.py:330-341."""
    if not isinstance(Cui, gpu.CSRMatrix):
        Cui = gpu.CSRMatrix(Cui)
    if not isinstance(X, gpu.Matrix):
        X = gpu.Matrix(X)
    if not isinstance(Y, gpu.Matrix):
        Y = gpu.Matrix(Y)
    if solver is None:
        solver = gpu.LeastSquaresSolver()
    return solver.calculate_loss(Cui, X, Y, regularization)
