"""Round 5: the f = 256 CG objects of bench.py alone (cg_c3_f192 / f256 on the configs[2] matrix, cg_c5 on the ml-20m shape),
per-kernel HIP-event times.  IMP_F256_OLD=1 for the A/B."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import implicit_amd.gpu as gpu
from implicit_amd.synthetic import SHAPES, named

def run(C, f, tag):
    Ct = C.T.tocsr()
    rng = np.random.default_rng(7)
    X = gpu.Matrix(rng.random((C.shape[0], f), dtype=np.float32) * 0.01)
    Y = gpu.Matrix(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
    Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
    gram = gpu.Matrix.zeros(f, f)
    solver = gpu.LeastSquaresSolver()
    def cg():
        solver.calculate_yty(Y, gram, bench.REG); solver.least_squares(Cd, X, gram, Y, bench.CG_STEPS)
        solver.calculate_yty(X, gram, bench.REG); solver.least_squares(Ctd, Y, gram, X, bench.CG_STEPS)
    t, k = bench._time_iterations(gpu, cg)
    gb = bench._iteration_bytes(C, Ct, f) / 1e9
    print(json.dumps({"tag": tag, "old": os.environ.get("IMP_F256_OLD"), "f": f, "ms_per_iter": 1e3 * t, "frac": gb / t / 8000.0,
                      "kernels": {a: round(b, 3) for a, b in k.items()}}), flush=True)

which = sys.argv[1:] or ["c3_256", "c5"]
if "c3_256" in which or "c3_192" in which:
    C3 = named("lastfm360k")
    if "c3_256" in which: run(C3, 256, "cg_c3_f256")
    if "c3_192" in which: run(C3, 192, "cg_c3_f192")
if "c5" in which:
    run(named("ml20m"), 256, "cg_c5")
