import sys, numpy as np
sys.path.insert(0,'.')
import implicit_amd.gpu as gpu
from oracle import oracle
rng = np.random.default_rng(12)
ni, f, nq, k = 40_000, 128, 256, 10
items = (rng.standard_normal((ni, f)) * 0.05).astype(np.float32)
items[::997] *= 40.0
items[1::2000] = items[3]
queries = (rng.standard_normal((nq, f)) * 0.1).astype(np.float32)
queries[::5] *= 1e-4
queries[1::5] *= 300.0
queries[7] = 1e-30
queries[9] = items[3] * 2
queries[11] = 0.0
want_ids, want_d = oracle.topk(items, queries, k + 1)
knn = gpu.KnnQuery()
ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(queries), k)
bad = np.nonzero((ids != want_ids[:, :k]).any(axis=1))[0]
print("rows differing", bad)
for r in bad[:4]:
    print(r, "gpu", ids[r], d[r]); print("   want", want_ids[r], want_d[r])
