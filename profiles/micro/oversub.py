"""What a resident foreign kernel does to the persistent row kernels, and what oversubscription buys back.

Rank 0's eighth of BASELINE configs[3] (the per-rank work of the 8-GPU run), CG iteration on one GPU:
  clean            : nothing else on the device
  occupied         : 32 stand-in workgroups (256 threads, 32 KB LDS each: roughly what RCCL's send/recv kernels hold
                     during an exchange) parked on another stream for the duration (imp_debug_occupy)
each with the row kernels launched 1x (one workgroup per slot, fixed shares) and 4x oversubscribed
(imp_set_oversubscribe, what the multi-GPU driver selects).  Prints ms per iteration.
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import implicit_amd.gpu as gpu  # noqa: E402
from implicit_amd.synthetic import SHAPES, grid_shards  # noqa: E402

users, items, nnz, gamma = SHAPES["c4"]
Cui, Ciu, u_off, i_off = grid_shards(0, 8, users, items, nnz, 8, gamma=gamma, seed=42)
f = 128
Cu, Ci = gpu.CSRMatrix(Cui), gpu.CSRMatrix(Ciu)
X = gpu.RandomState(7).uniform(users, f, 0.0, 0.01)
Y = gpu.RandomState(8).uniform(items, f, 0.0, 0.01)
Xs, Ys = X[int(u_off[0]):int(u_off[1])], Y[int(i_off[0]):int(i_off[1])]
solver = gpu.LeastSquaresSolver()
gram = gpu.Matrix.zeros(f, f)


def iteration():
    solver.calculate_yty(Y, gram, 0.01)
    solver.least_squares(Cu, Xs, gram, Y, 3)
    solver.calculate_yty(X, gram, 0.01)
    solver.least_squares(Ci, Ys, gram, X, 3)


def timed(n=3):   # every solver call returns after its stream has drained; no device-wide synchronise (the stand-in kernel
    t0 = time.perf_counter()   # on its own stream is still running)
    for _ in range(n):
        iteration()
    return (time.perf_counter() - t0) / n * 1e3


iteration()
for blockers in (0, 32, 64):
    for over in (1, 4):
        gpu.set_oversubscribe(over)
        iteration()
        if blockers:
            gpu.debug_occupy(blockers, 400_000)  # 0.4 s: resident before and throughout the timed iterations
            time.sleep(0.01)
        ms = timed()
        time.sleep(0.5 if blockers else 0.0)   # let the stand-in kernel run out
        print(f"occupying workgroups {blockers:3d}  oversubscription {over}x : {ms:7.2f} ms / iteration", flush=True)
