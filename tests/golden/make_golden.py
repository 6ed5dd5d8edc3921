#!/usr/bin/env python3
"""Generates tests/golden/als_golden.npz from the REFERENCE ITSELF (the compiled Cython modules in
oracle/_ref, built from /root/reference/implicit/cpu/{_als.pyx,topk.pyx,select.h} by
oracle/build_ref.py).  Run in the build container only (the reference tree is not on the GPU box):

    python oracle/build_ref.py && python tests/golden/make_golden.py

Every array in the file is either a seeded input or the reference's output on it; the parity tests
compare both the plain-C oracle (CPU suite) and the HIP kernels (GPU suite) with these outputs.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from implicit_amd.synthetic import synthetic_csr  # noqa: E402
from oracle import ref  # noqa: E402

als, topk = ref.load()
assert als is not None, "build the reference first: python oracle/build_ref.py"
from threadpoolctl import threadpool_limits  # noqa: E402

out = {}


def put_csr(prefix, m):
    out[prefix + "_indptr"] = m.indptr.astype(np.int32)
    out[prefix + "_indices"] = m.indices.astype(np.int32)
    out[prefix + "_data"] = m.data.astype(np.float32)
    out[prefix + "_shape"] = np.array(m.shape, dtype=np.int64)


with threadpool_limits(1, "blas"):
    # ---- ALS solver cases: (name, users, items, nnz, factors) -----------------------------------
    cases = [("f6", 60, 40, 600, 6), ("f16", 120, 90, 2400, 16), ("f50", 150, 110, 3500, 50),
             ("f64", 200, 150, 6000, 64), ("f128", 160, 140, 5000, 128)]
    out["als_cases"] = np.array([c[0] for c in cases])
    for name, users, items, nnz, f in cases:
        C = synthetic_csr(users, items, nnz, seed=100 + f, neg_frac=0.08, empty_frac=0.04)
        C.data[::53] = 0.0  # explicit zeros (confidence-0 branch)
        Ct = C.T.tocsr()
        rng = np.random.default_rng(f)
        X0 = (rng.random((users, f), dtype=np.float32) - 0.5).astype(np.float32) * 0.2
        Y0 = (rng.random((items, f), dtype=np.float32) - 0.5).astype(np.float32) * 0.2
        put_csr(name + "_C", C)
        out[name + "_X0"], out[name + "_Y0"] = X0, Y0
        for steps in (1, 3):
            X = X0.copy()
            als.least_squares_cg(C, X, Y0, 0.05, num_threads=1, cg_steps=steps)
            out[f"{name}_cg{steps}_X"] = X
        Y = Y0.copy()
        als.least_squares_cg(Ct, Y, out[f"{name}_cg3_X"], 0.05, num_threads=1, cg_steps=3)
        out[name + "_cg3_Y"] = Y
        X64 = X0.astype(np.float64)
        als.least_squares_cg(C, X64, Y0.astype(np.float64), 0.05, num_threads=1, cg_steps=3)
        out[name + "_cg3_X_f64"] = X64
        X = X0.copy()
        als.least_squares(C, X, Y0, 0.05, num_threads=1)
        out[name + "_chol_X"] = X
        out[name + "_loss"] = np.array([als.calculate_loss(C, out[f"{name}_cg3_X"], Y0, r, num_threads=1)
                                        for r in (0.0, 0.05, 10.0)])

    # ---- the reference's own known-answer fixtures -----------------------------------------------
    # tests/als_test.py:142-186 (7x6 matrix reconstructed to 1e-3 with regularization=0, alpha=2)
    counts = sp.csr_matrix(np.array([[1, 1, 0, 1, 0, 0], [0, 1, 1, 1, 0, 0], [1, 0, 1, 0, 0, 0], [1, 1, 0, 0, 0, 0],
                                     [0, 0, 1, 1, 0, 1], [0, 1, 0, 0, 0, 1], [0, 0, 0, 0, 1, 1]], dtype=np.float32))
    put_csr("factorize_counts", counts)
    Cui = (2.0 * counts).astype(np.float32).tocsr()
    Ciu = Cui.T.tocsr()
    for solver_name in ("cg", "chol"):
        rng = np.random.default_rng(42)
        X = rng.random((7, 6), dtype=np.float32) * 0.01
        Y = rng.random((6, 6), dtype=np.float32) * 0.01
        out["factorize_X0"], out["factorize_Y0"] = X.copy(), Y.copy()
        for _ in range(15):
            if solver_name == "cg":
                als.least_squares_cg(Cui, X, Y, 0.0, num_threads=1, cg_steps=3)
                als.least_squares_cg(Ciu, Y, X, 0.0, num_threads=1, cg_steps=3)
            else:
                als.least_squares(Cui, X, Y, 0.0, num_threads=1)
                als.least_squares(Ciu, Y, X, 0.0, num_threads=1)
        out[f"factorize_{solver_name}_X"], out[f"factorize_{solver_name}_Y"] = X, Y

    # ---- top-k / select -----------------------------------------------------------------------------
    rng = np.random.default_rng(5)
    items = (rng.standard_normal((500, 24)) * 0.3).astype(np.float32)
    query = (rng.standard_normal((37, 24)) * 0.3).astype(np.float32)
    liked = sp.random(37, 500, density=0.02, format="csr", dtype=np.float32, random_state=1)
    filt = np.array([3, 77, 499], dtype=np.int32)
    norms = np.linalg.norm(items, axis=1).astype(np.float32)
    out["topk_items"], out["topk_query"], out["topk_filter_items"], out["topk_norms"] = items, query, filt, norms
    put_csr("topk_liked", liked)
    for tag, kw in (("plain", {}), ("norms", {"item_norms": norms}),
                    ("filters", {"filter_query_items": liked, "filter_items": filt}),
                    ("all", {"item_norms": norms, "filter_query_items": liked, "filter_items": filt})):
        for k in (1, 10, 64):
            ids, dist = topk.topk(items, query, k, num_threads=1, **kw)
            out[f"topk_{tag}_k{k}_ids"], out[f"topk_{tag}_k{k}_dist"] = ids, dist
    # tie semantics of select.h: exact ties inside / at the boundary, k > cols
    tie_rows = [np.array(r, dtype=np.float32) for r in ([5, 5, 5, 9], [9, 5, 5, 5], [5, 5, 9, 5], [1, 1, 1, 1, 1],
                                                        [3, 1, 2], [2, 7, 7, 7, 1, 7, 3])]
    for i, row in enumerate(tie_rows):
        q = np.ones((1, 1), dtype=np.float32)
        it = row.reshape(-1, 1)
        out[f"tie{i}_row"] = row
        for k in (2, 3, 5):
            ids, dist = topk.topk(it, q, k, num_threads=1)
            out[f"tie{i}_k{k}_ids"], out[f"tie{i}_k{k}_dist"] = ids, dist
    out["n_ties"] = np.array(len(tie_rows))

path = os.path.join(HERE, "als_golden.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
