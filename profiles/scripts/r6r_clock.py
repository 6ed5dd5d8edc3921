# shader clock WHILE the kernels run (probe on a side stream beside a queued step): CG step at configs[2], top-k batches
import sys, time, numpy as np
sys.path.insert(0, '.')
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import synthetic_csr
U, I, f = 358868, 292385, 128
C = synthetic_csr(U, I, 17_300_000, seed=42)
Ct = C.T.tocsr()
rng = np.random.default_rng(7)
X = gpu.Matrix((rng.standard_normal((U, f)) * 0.1).astype(np.float32))
Y = gpu.Matrix((rng.standard_normal((I, f)) * 0.1).astype(np.float32))
Cu, Ci = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
solver = gpu.LeastSquaresSolver()
gram = gpu.Matrix.zeros(f, f)
def step():
    solver.calculate_yty(Y, gram, 0.01); solver.least_squares(Cu, X, gram, Y, 3)
    solver.calculate_yty(X, gram, 0.01); solver.least_squares(Ci, Y, gram, X, 3)
for _ in range(5): step()
gpu.synchronize()
print("idle / behind:", round(gpu.core_clock_mhz(200), 1))
gpu.set_deferred_sync(True)
for us in (500, 1500, 3000):
    for rep in range(3):
        for _ in range(6): step()          # ~24 ms of queued work
        time.sleep(0.004)                  # let the first steps start
        mhz = gpu.core_clock_mhz(-us)
        gpu.synchronize()
        print(f"beside CG steps, {us} us probe: {mhz:.1f} MHz")
gpu.set_deferred_sync(False)
knn = gpu.KnnQuery()
q = [X[s:s + 1000] for s in range(0, 20000, 1000)]
for v in q: knn.topk(Y, v, 10)
gpu.synchronize()
print("behind top-k:", round(gpu.core_clock_mhz(100), 1))
