# where model.recommend()'s host time goes (per 1000-user batch, configs[2] shape, random factors)
import sys, time, numpy as np
sys.path.insert(0, '.')
import os
import implicit_amd._libpath as _lp
if os.environ.get('IMP_LIB_PATH'): _lp.OVERRIDE = os.environ['IMP_LIB_PATH']
import implicit_amd.gpu as gpu
from implicit_amd.als import AlternatingLeastSquares
from implicit_amd.synthetic import synthetic_csr
U, I, f, B, Q = 358868, 292385, 128, 1000, 20000
C = synthetic_csr(U, I, 17_300_000, seed=42)
rng = np.random.default_rng(7)
model = AlternatingLeastSquares(factors=f, use_gpu=True)
model.user_factors = gpu.Matrix((rng.standard_normal((U, f)) * 0.1).astype(np.float32))
model.item_factors = gpu.Matrix((rng.standard_normal((I, f)) * 0.1).astype(np.float32))
ids = np.arange(Q)
slices = [C[s:s + B] for s in range(0, Q, B)]
def timeit(fn, n=3):
    fn(); gpu.synchronize()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); fn(); gpu.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / len(slices) * 1e3
def full():
    for j, s in enumerate(range(0, Q, B)): model.recommend(ids[s:s + B], slices[j], N=10)
def coo_only():
    for sl in slices: gpu.COOMatrix.from_csr_pattern(sl)
def repeat_only():
    for sl in slices: np.repeat(np.arange(sl.shape[0], dtype=np.int32), np.diff(np.asarray(sl.indptr)))
views = [model.user_factors[s:s + B] for s in range(0, Q, B)]
coos = [gpu.COOMatrix.from_csr_pattern(sl) for sl in slices]
def knn_only():
    for v, c in zip(views, coos): model.knn.topk(model.item_factors, v, 10, query_filter=c)
def view_only():
    for s in range(0, Q, B): model.user_factors[s:s + B]
def check_only():
    for s in range(0, Q, B):
        u = ids[s:s + B]; (np.diff(u) == 1).all()
def slice_only():
    for s in range(0, Q, B): C[s:s + B]
for name, fn in [("recommend (presliced)", full), ("from_csr_pattern", coo_only), ("np.repeat", repeat_only), ("knn.topk", knn_only),
                 ("row-range view", view_only), ("consecutive check", check_only), ("caller's slice", slice_only)]:
    print(f"{name:24s} {timeit(fn):.4f} ms per batch")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); full(); gpu.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
