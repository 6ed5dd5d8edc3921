"""Per-kernel times of one recommend()-shaped top-k batch (1000 queries x 292 385 items x f = 128, k = 10) on random factors:
the quick driver for A/B and knock-out libraries (IMP_LIB_PATH), without bench.py's matrix generation."""
import sys
import numpy as np
sys.path.insert(0, ".")
import implicit_amd.gpu as gpu

rng = np.random.default_rng(0)
Y = gpu.Matrix((rng.random((292385, 128), dtype=np.float32) - 0.5) * 0.2)
Q = gpu.Matrix((rng.random((1000, 128), dtype=np.float32) - 0.5) * 0.2)
knn = gpu.KnnQuery()
knn.topk(Y, Q, 10)
gpu.synchronize()
gpu.Profiler.reset()
gpu.Profiler.enable(True)
n = 5
for _ in range(n):
    knn.topk(Y, Q, 10)
gpu.synchronize()
gpu.Profiler.enable(False)
print({name: round(gpu.Profiler.get(name)[0] / n, 4) for name in gpu.Profiler.names()})
