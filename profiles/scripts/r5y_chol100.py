"""Round 5: the f = 100 Cholesky half sweep zero-padded onto the f = 128 normal-matrix path (als_cholesky.hip; IMP_CHOL_PAD=0 keeps the
workgroup kernel): per-row parity against the oracle on every row class, then a configs[2]-shaped iteration's time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import named
from oracle import oracle
from test_gpu_nm import _long_row_matrix
oracle.build()
f = 100
lengths = [1, 2, 3, 5, 16, 17, 33, 64, 65, 129, 300, 512, 513, 700, 1500, 2049, 5000, 9000, 0, 40]
C = _long_row_matrix(lengths, 12000, seed=3)
rng = np.random.default_rng(1)
Y = ((rng.random((12000, f), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
X = np.zeros((len(lengths), f), np.float32)
want = X.copy(); oracle.least_squares(C, want, Y, 0.01)
solver = gpu.LeastSquaresSolver()
Xd, Yd, gram = gpu.Matrix(X), gpu.Matrix(Y), gpu.Matrix.zeros(f, f)
solver.calculate_yty(Yd, gram, 0.0)
solver.least_squares_cholesky(gpu.CSRMatrix(C), Xd, gram, Yd, 0.01)
got = Xd.to_numpy()
err = np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
print("per-row rel", np.array2string(err, precision=2), "max", err.max(), "fixups", gpu.fixup_rows(), flush=True)
C3 = named("lastfm360k"); Ct = C3.T.tocsr()
rng = np.random.default_rng(2)
X0 = (rng.random((C3.shape[0], f), dtype=np.float32) * 0.01); Y0 = (rng.random((C3.shape[1], f), dtype=np.float32) * 0.01)
Xd, Yd, gram = gpu.Matrix(X0), gpu.Matrix(Y0), gpu.Matrix.zeros(f, f)
Cd, Ctd = gpu.CSRMatrix(C3), gpu.CSRMatrix(Ct)
def it():
    solver.calculate_yty(Yd, gram, 0.0); solver.least_squares_cholesky(Cd, Xd, gram, Yd, 0.01)
    solver.calculate_yty(Xd, gram, 0.0); solver.least_squares_cholesky(Ctd, Yd, gram, Xd, 0.01)
it(); gpu.synchronize(); t0 = time.perf_counter()
for _ in range(3): it()
gpu.synchronize(); print("f=100 cholesky ms per iteration %.2f" % ((time.perf_counter() - t0) / 3 * 1e3), "finite", bool(np.isfinite(Xd.to_numpy()).all()))
