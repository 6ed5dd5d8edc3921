"""Where does a KnnQuery.topk call spend its time besides the kernels?  Times 1000-query calls (a) as bench.py does
(host outputs, profiler on), (b) profiler off, (c) outputs left on the device (C-ABI called with device pointers)."""
import ctypes
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import implicit_amd.gpu as gpu
from implicit_amd.gpu._hip import lib

items, f, nq, k = 292_385, 128, 1000, 10
rng = np.random.default_rng(0)
Y = gpu.Matrix(rng.standard_normal((items, f), dtype=np.float32))
Q = gpu.Matrix(rng.standard_normal((nq, f), dtype=np.float32))
knn = gpu.KnnQuery()


def timeit(fn, n=20):
    fn()
    gpu.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    gpu.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


gpu.Profiler.reset()
gpu.Profiler.enable(True)
a = timeit(lambda: knn.topk(Y, Q, k))
gpu.Profiler.enable(False)
kern = {n: gpu.Profiler.get(n)[0] / 21 for n in gpu.Profiler.names()}
b = timeit(lambda: knn.topk(Y, Q, k))
ids = gpu.Matrix.zeros(nq, k)  # float32 storage reused as the int32 output buffer
dist = gpu.Matrix.zeros(nq, k)
pi, pd = ctypes.c_void_p(), ctypes.c_void_p()
lib().imp_matrix_device_ptr(ids._h, ctypes.byref(pi))
lib().imp_matrix_device_ptr(dist._h, ctypes.byref(pd))
c = timeit(lambda: lib().imp_knn_topk(knn._h, Y._h, Q._h, k, pi, pd, None, None, None))
print(f"host outputs + profiler {a:.3f} ms | host outputs {b:.3f} ms | device outputs {c:.3f} ms | kernels {sum(kern.values()):.3f} ms {kern}")
