"""N LOGICAL ranks on ONE device: the multi-GPU driver (`sharded.fit_sharded`, `shard_transpose`, the chunked half
sweeps) run as it runs on N GPUs -- same partition, same local chunk schedules, same oversubscribed launches, same deferred
iterations -- with the exchange replaced by device-to-device copies between the ranks' replicas.

SURVEY.md section 8(e) prescribes this harness for boxes that expose a single GPU (RCCL refuses two ranks on one device):
every logical rank is a Python thread with its OWN replicas of X and Y and its own CSR shards; the ranks share the device,
the library stream and its call lock, so their kernels interleave call by call (round-robin in effect).  What differs from
the real thing is only the transport:

  allreduce_sum       every rank deposits its matrix, all wait, each rank sums the N deposits in RANK ORDER (a fixed
                      association, as a ring all-reduce has) and overwrites its own
  allgather_rows*     after a barrier (every owner has QUEUED the solve of the rows it contributes -- the library stream
                      orders the copy behind it) each rank copies the other owners' row ranges out of their replicas
                      (`imp_matrix_copy_rows`); `occupy=(workgroups, microseconds)` additionally parks a resident foreign
                      kernel on another stream per exchange (`imp_debug_occupy`), standing in for RCCL's send / recv kernels
                      beside the next chunk's solve
  alltoall_rows       the same with the personalised row ranges of the set-up exchange

`run(n, fn)` starts the N threads and returns their results; an exception in any rank aborts the barriers of the others.
Nothing here is a fallback of the product: `Comm` (RCCL) is what N processes use.
"""
import threading

import numpy as np


class _World:
    def __init__(self, nranks, gpu, occupy):
        self.nranks, self.gpu, self.occupy = nranks, gpu, occupy
        self.barrier = threading.Barrier(nranks)
        self.slots = [None] * nranks
        self.stats = {"allreduce": 0, "allgather_rows_copied": 0, "alltoall_rows_copied": 0, "occupied": 0}
        self.lock = threading.Lock()


class LocalComm:
    """The `comm` interface of implicit_amd.gpu.sharded for one logical rank (see module docstring)."""

    def __init__(self, world, rank):
        self._w, self.rank, self.nranks = world, int(rank), world.nranks

    # -- collectives -----------------------------------------------------------------------------------------------------
    def _exchange(self, item):
        """Deposit `item`, wait for everybody, return the list of all deposits (valid until the closing barrier)."""
        w = self._w
        w.slots[self.rank] = item
        w.barrier.wait()
        return list(w.slots)

    def _close(self):
        self._w.barrier.wait()

    def allreduce_sum(self, m):
        parts = self._exchange(m)
        total = parts[0].to_numpy().astype(np.float32, copy=True)
        for other in parts[1:]:
            total += other.to_numpy()
        self._close()  # everybody has read every deposit: now they may be overwritten
        m.copy_from_numpy(total)
        if self.rank == 0:
            self._w.stats["allreduce"] += 1

    def _gather(self, full, lo, hi, key):
        parts = self._exchange(full)
        copied = 0
        for q in range(self.nranks):
            n = int(hi[q]) - int(lo[q])
            if q != self.rank and n > 0:
                full.copy_rows_from(int(lo[q]), parts[q], int(lo[q]), n)
                copied += n
        w = self._w
        if w.occupy and self.rank == 0:
            w.gpu.debug_occupy(int(w.occupy[0]), int(w.occupy[1]))
            w.stats["occupied"] += 1
        with w.lock:
            w.stats[key] += copied
        self._close()

    def allgather_rows(self, full, row_offsets):
        self._gather(full, row_offsets[:-1], row_offsets[1:], "allgather_rows_copied")

    def allgather_rows_begin(self, full, row_lo, row_hi):
        self._gather(full, row_lo, row_hi, "allgather_rows_copied")

    def allgather_rows_end(self):
        pass  # the copies were queued on the library stream itself: already ordered before the next kernels

    def alltoall_rows(self, send, send_lo, send_hi, recv, recv_lo, recv_hi):
        parts = self._exchange((send, [int(v) for v in send_lo], [int(v) for v in send_hi]))
        copied = 0
        for q in range(self.nranks):
            s, lo, hi = parts[q]
            n = hi[self.rank] - lo[self.rank]
            if n != int(recv_hi[q]) - int(recv_lo[q]):
                raise ValueError("alltoall_rows: send and receive ranges do not match")
            if n > 0:
                recv.copy_rows_from(int(recv_lo[q]), s, lo[self.rank], n)
                copied += n
        self._w.gpu.synchronize()  # blocking, like the RCCL form: the senders may release their buffers afterwards
        with self._w.lock:
            self._w.stats["alltoall_rows_copied"] += copied
        self._close()

    def barrier(self):
        self._w.barrier.wait()

    @property
    def stats(self):
        return dict(self._w.stats)


def run(nranks, fn, gpu=None, occupy=None, oversubscribe=4):
    """Run `fn(comm)` on `nranks` logical ranks (threads) of the current device and return [fn's result per rank].

    `oversubscribe`: launch factor of the persistent row kernels for the duration (what GpuBackend sets with nranks > 1; set
    here once so that N backends opening and closing in any order cannot leave it changed).  `occupy`: see module docstring."""
    if gpu is None:
        import implicit_amd.gpu as gpu
    world = _World(nranks, gpu, occupy)
    results, errors = [None] * nranks, [None] * nranks
    before = gpu.get_oversubscribe()
    if oversubscribe:
        gpu.set_oversubscribe(int(oversubscribe))

    device = gpu.get_device() if hasattr(gpu, "get_device") else None

    def body(rank):
        try:
            if device is not None:
                gpu.set_device(device)  # the current device is per thread
            results[rank] = fn(LocalComm(world, rank))
        except BaseException as e:  # noqa: BLE001 -- reported below; the other ranks must not wait for this one
            errors[rank] = e
            world.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,), name=f"logical-rank-{r}") for r in range(nranks)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        gpu.set_deferred_sync(False)
        gpu.set_oversubscribe(before)
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real:
        raise real[0]
    if any(e is not None for e in errors):
        raise errors[[e is not None for e in errors].index(True)]
    return results
