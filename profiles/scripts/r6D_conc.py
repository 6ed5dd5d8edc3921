# concentrated scores (all-positive nearly parallel factors): what a recommend-style batch costs when the screen fails
import sys, time, numpy as np
sys.path.insert(0, '.')
import implicit_amd.gpu as gpu
rng = np.random.default_rng(2)
ni, f, nq, k = 292_385, 128, 1000, 10
items = (rng.random((ni, f), dtype=np.float32) * 0.01 + 0.02).astype(np.float32)
queries = (rng.random((nq, f), dtype=np.float32) * 0.01 + 0.02).astype(np.float32)
knn, I, Q = gpu.KnnQuery(), gpu.Matrix(items), gpu.Matrix(queries)
knn.topk(I, Q, k); gpu.synchronize()
t0 = time.perf_counter()
for _ in range(10): knn.topk(I, Q, k)
gpu.synchronize()
print("ms per 1000-row batch:", (time.perf_counter() - t0) / 10 * 1e3)
gpu.Profiler.reset(); gpu.Profiler.enable(True)
for _ in range(5): knn.topk(I, Q, k)
gpu.synchronize(); gpu.Profiler.enable(False)
print({n: round(gpu.Profiler.get(n)[0] / 5, 4) for n in gpu.Profiler.names()})
