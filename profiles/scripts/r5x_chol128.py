"""Round 5: the f = 128 Cholesky half sweep through the rows' normal matrices (als_cg_nm.hip nm_chol): per-row parity against the
oracle on every row class, then the configs[2]-shaped timing of bench.py's cholesky_c3_f128 object.  IMP_CHOL_NM=0 for the A/B."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import named, synthetic_csr
from oracle import oracle
from test_gpu_nm import _long_row_matrix
oracle.build()
f = 128
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
if "time" in sys.argv:
    C3 = named("lastfm360k")
    out = bench.extra_cholesky_f128(gpu, C3, C3.T.tocsr())
    for key in out:
        print(key, json.dumps({k: v for k, v in out[key].items() if k in ("ms_per_iter", "tflops", "kernels_ms_per_iter")}))
