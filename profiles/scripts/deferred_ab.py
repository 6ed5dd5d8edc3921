import sys, time, warnings, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
warnings.simplefilter("ignore")
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import named
C = named("lastfm360k"); Ct = C.T.tocsr()
f = 128
rng = np.random.default_rng(7)
X = gpu.Matrix(rng.random((C.shape[0], f), dtype=np.float32) * 0.01)
Y = gpu.Matrix(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
gram = gpu.Matrix.zeros(f, f)
s = gpu.LeastSquaresSolver()
Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
def step():
    s.calculate_yty(Y, gram, 0.01); s.least_squares(Cd, X, gram, Y, 3)
    s.calculate_yty(X, gram, 0.01); s.least_squares(Ctd, Y, gram, X, 3)
for mode in (False, True, False, True):
    for _ in range(3): step()
    gpu.synchronize()
    gpu.set_deferred_sync(mode)
    t0 = time.perf_counter()
    for _ in range(20):
        step()
        if mode: gpu.synchronize()   # one host wait per iteration
    gpu.synchronize()
    dt = (time.perf_counter() - t0) / 20
    gpu.set_deferred_sync(False)
    print("deferred" if mode else "synchronous", "%.3f ms / iteration" % (1e3 * dt), flush=True)
