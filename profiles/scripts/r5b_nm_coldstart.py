"""Round 5: where the normal-matrix kernel's distance from float64 comes from on ALL-POSITIVE cold-start factors (the first
sweep of every fit, implicit/gpu/als.py:98-101): operand scale on / off (IMP_NM_SCALE), old long-row kernels (IMP_NM=0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import implicit_amd.gpu as gpu
from oracle import oracle
from test_gpu_nm import _long_row_matrix, _solve, _row_errors
oracle.build()
f, items = 128, 9000
C = _long_row_matrix([600, 1300, 4000, 9000], items, seed=21)
for name, gen in (("uniform(0,s)", lambda rng, shape, s: rng.random(shape) * s), ("uniform(-s/2,s/2)", lambda rng, shape, s: (rng.random(shape) - 0.5) * s)):
    for scale in (0.01, 1e-3):
        rng = np.random.default_rng(4)
        Y = gen(rng, (items, f), scale).astype(np.float32)
        X = gen(rng, (4, f), scale).astype(np.float32)
        want = X.copy(); oracle.least_squares_cg(C, want, Y, 0.01)
        exact = oracle.least_squares_cg_f64(C, X, Y, 0.01)
        got = _solve(gpu, C, X.copy(), Y, 0.01, 3)
        print(os.environ.get("TAG"), name, scale, "gpu-fp64", _row_errors(got, exact), "oracle-fp64", _row_errors(want, exact), "fixups", gpu.fixup_rows(), flush=True)
