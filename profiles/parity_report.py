#!/usr/bin/env python3
"""Generates PARITY.md on the GPU box: per BASELINE configuration and side, the distance of the HIP path from

  * the CPU oracle (oracle/als_oracle.c),
  * the COMPILED REFERENCE itself (oracle/_ref: implicit/cpu/_als.pyx, topk.pyx built from /root/reference), and
  * the same row solve in float64 (what both fp32 paths approximate),

on sampled rows of a full-size half sweep (a row's solve depends only on its nonzeros, the other side's factors and the
gramian), plus the near-tie swap rate of top-k against the compiled reference's `topk`.  Everything the parity tests bound
with an inequality is recorded here as a number.

    python profiles/parity_report.py [out_dir]          # writes <out_dir>/PARITY.md and parity.json (default gpurun_out/parity)

Test infrastructure: imports oracle/ (allowed for tests/, smoke() and bench.py's cpu_baseline only -- this script is part of
the test side; the product never does).
"""
import json
import os
import sys
import time
import warnings

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from test_gpu_fullsize import _cg_fp64, _reference_cg, rel  # noqa: E402


def sample_rows(C, n_uniform, n_longest):
    lens = np.diff(C.indptr)
    rows = np.unique(np.concatenate([np.arange(0, C.shape[0], max(1, C.shape[0] // n_uniform)), np.argsort(lens)[-n_longest:]]))
    return rows, lens


def cg_side(gpu, oracle, name, side, C, X0, Y0, reg, cg_steps=3, n_uniform=2000, n_longest=64, n_fp64=400, Yd=None):
    """One CG half sweep on the GPU at full size; distances on sampled rows.  Returns (record, solved X as numpy)."""
    f = X0.shape[1]
    solver = gpu.LeastSquaresSolver()
    Xd = gpu.Matrix(X0)
    Yd = Yd if Yd is not None else gpu.Matrix(Y0)
    gram = gpu.Matrix.zeros(f, f)
    solver.calculate_yty(Yd, gram, reg)
    solver.least_squares(gpu.CSRMatrix(C), Xd, gram, Yd, cg_steps)
    got = Xd.to_numpy()
    gram_h = gram.to_numpy()
    rows, lens = sample_rows(C, n_uniform, n_longest)
    sub = C[rows]
    want = np.ascontiguousarray(X0[rows])
    oracle.least_squares_cg(sub, want, Y0, reg, cg_steps=cg_steps, YtY=gram_h)
    ref_rows = _reference_cg(sub, np.ascontiguousarray(X0[rows]), Y0, reg, cg_steps=cg_steps)
    # float64 on a subsample (python loop per row): the uniform part thinned + the longest rows
    pick = np.unique(np.concatenate([np.arange(0, len(rows), max(1, len(rows) // n_fp64)), np.arange(len(rows) - 16, len(rows))]))
    exact = np.stack([_cg_fp64(sub[int(i)], X0[rows[int(i)]], Y0, gram_h, cg_steps) for i in pick])
    rec = {"config": name, "side": side, "rows": int(C.shape[0]), "nnz": int(C.nnz), "factors": int(f), "sampled_rows": int(len(rows)),
           "max_row_nnz": int(lens.max()), "fp64_rows": int(len(pick)),
           "gpu_vs_oracle": rel(got[rows], want),
           "gpu_vs_reference": rel(got[rows], ref_rows) if ref_rows is not None else None,
           "oracle_vs_reference": rel(want, ref_rows) if ref_rows is not None else None,
           "gpu_vs_fp64": rel(got[rows][pick], exact), "oracle_vs_fp64": rel(want[pick], exact),
           "reference_vs_fp64": rel(ref_rows[pick], exact) if ref_rows is not None else None}
    return rec, got


def cholesky_side(gpu, oracle, name, side, C, Y0, reg, n_uniform=2000, n_longest=64):
    f = Y0.shape[1]
    solver = gpu.LeastSquaresSolver()
    Xd, Yd = gpu.Matrix.zeros(C.shape[0], f), gpu.Matrix(Y0)
    gram = gpu.Matrix.zeros(f, f)
    solver.calculate_yty(Yd, gram, 0.0)
    solver.least_squares_cholesky(gpu.CSRMatrix(C), Xd, gram, Yd, reg)
    got = Xd.to_numpy()
    rows, lens = sample_rows(C, n_uniform, n_longest)
    want = np.zeros((len(rows), f), dtype=np.float32)
    oracle.least_squares(C[rows], want, Y0, reg)
    from oracle import ref

    als_ref, _ = ref.load()
    ref_rows = None
    if als_ref is not None:
        from threadpoolctl import threadpool_limits

        ref_rows = np.zeros((len(rows), f), dtype=np.float32)
        with threadpool_limits(1, "blas"):
            als_ref.least_squares(C[rows], ref_rows, Y0, reg, num_threads=16)
    # float64: the normal equations of each sampled row solved by numpy
    Y64 = Y0.astype(np.float64)
    G = Y64.T @ Y64 + reg * np.eye(f)
    pick = np.arange(0, len(rows), max(1, len(rows) // 300))
    exact = []
    for i in pick:
        row = C[int(rows[i])]
        Yu, c = Y64[row.indices], row.data.astype(np.float64)
        A = G + (Yu.T * (np.abs(c) - 1.0)) @ Yu
        b = Yu.T @ np.where(c > 0, c, 0.0)
        exact.append(np.linalg.solve(A, b) if row.nnz else np.zeros(f))
    exact = np.stack(exact)
    return {"config": name, "side": side + " (Cholesky)", "rows": int(C.shape[0]), "nnz": int(C.nnz), "factors": int(f),
            "sampled_rows": int(len(rows)), "max_row_nnz": int(lens.max()), "fp64_rows": int(len(pick)),
            "gpu_vs_oracle": rel(got[rows], want), "gpu_vs_reference": rel(got[rows], ref_rows) if ref_rows is not None else None,
            "oracle_vs_reference": rel(want, ref_rows) if ref_rows is not None else None,
            "gpu_vs_fp64": rel(got[rows][pick], exact), "oracle_vs_fp64": rel(want[pick], exact),
            "reference_vs_fp64": rel(ref_rows[pick], exact) if ref_rows is not None else None}


def topk_record(gpu, oracle, name, items, queries, k, norms=None):
    """ids / scores of KnnQuery.topk against the compiled reference's topk (oracle/_ref) and the oracle; every id mismatch is
    classified in float64: a near-tie (the two candidates' exact scores closer than the fp32 noise of an f-term dot product)
    or a real disagreement."""
    from oracle import ref

    f = items.shape[1]
    item_d = gpu.Matrix(items)
    norms_d = gpu.calculate_norms(item_d) if norms else None
    ids, d = gpu.KnnQuery().topk(item_d, gpu.Matrix(queries), k, item_norms=norms_d)
    nh = norms_d.to_numpy().reshape(-1) if norms else None
    o_ids, o_d = oracle.topk(items, queries, k, item_norms=nh)
    _, topk_ref = ref.load()
    r_ids = None
    if topk_ref is not None:
        from threadpoolctl import threadpool_limits

        with threadpool_limits(1, "blas"):
            r_ids, _ = topk_ref.topk(items, queries, k, item_norms=nh, num_threads=16)
    rec = {"config": name, "items": int(items.shape[0]), "queries": int(queries.shape[0]), "k": int(k), "factors": int(f),
           "norms": bool(norms), "score_rel_max": float(np.max(np.abs(d - o_d) / np.maximum(np.abs(o_d), 1e-30)))}
    tol = 4 * f * np.finfo(np.float32).eps
    I64, Q64 = items.astype(np.float64), queries.astype(np.float64)
    for label, other in (("oracle", o_ids), ("reference", r_ids)):
        if other is None:
            rec[f"vs_{label}"] = None
            continue
        differ = ids != other
        near, real = 0, 0
        for r, j in zip(*np.nonzero(differ)):
            a, b = int(ids[r, j]), int(other[r, j])
            sa, sb = I64[a] @ Q64[r], I64[b] @ Q64[r]
            if nh is not None:
                sa, sb = sa / nh[a], sb / nh[b]
            if abs(sa - sb) <= tol * max(abs(sa), abs(sb), 1e-30):
                near += 1
            else:
                real += 1
        rec[f"vs_{label}"] = {"id_positions_differing": int(differ.sum()), "of": int(differ.size),
                              "fraction": float(differ.mean()), "rows_with_a_difference": int(differ.any(axis=1).sum()),
                              "near_ties_fp64": near, "real_disagreements": real}
    return rec


def fmt(v):
    return "—" if v is None else f"{v:.1e}"


TOPK_FORMS = (("default: one-product screening pass + fp32 re-scoring of the candidates (emit path, no norms); fp16 x 2, three products elsewhere", {}),
              ("fp16 x 2, three products everywhere (IMP_TOPK_SCREEN=0)", {"IMP_TOPK_SCREEN": "0"}),
              ("bf16 x 3, six products (IMP_TOPK_RESIDENT=0)", {"IMP_TOPK_RESIDENT": "0"}),
              ("exact fp32 MFMA (IMP_TOPK_FP32_MFMA=1)", {"IMP_TOPK_FP32_MFMA": "1", "IMP_TOPK_RESIDENT": "0"}))


def topk_forms(out_dir, cases):
    """The same top-k workloads under each form of the scoring GEMM: the form is chosen once per process, so every form runs in
    a child process (`--topk-only`) on factors saved by the parent; returns {form label: [records]}."""
    import subprocess

    path = os.path.join(out_dir, "topk_cases.npz")
    np.savez(path, **{f"{i}_{k}": v for i, c in enumerate(cases) for k, v in c.items() if isinstance(v, np.ndarray)},
             meta=json.dumps([{k: v for k, v in c.items() if not isinstance(v, np.ndarray)} for c in cases]))
    out = {}
    for label, env in TOPK_FORMS:
        res = os.path.join(out_dir, "topk_form.json")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--topk-only", path, res], env={**os.environ, **env},
                           capture_output=True, text=True, timeout=1800)
        if p.returncode != 0:
            out[label] = {"error": p.stderr[-500:]}
            continue
        out[label] = json.load(open(res))
    os.remove(path)
    return out


def topk_only(case_path, res_path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import implicit_amd.gpu as gpu
    from oracle import oracle

    oracle.build()
    z = np.load(case_path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    recs = []
    for i, m in enumerate(meta):
        recs.append(topk_record(gpu, oracle, m["name"], z[f"{i}_items"], z[f"{i}_queries"], m["k"], norms=m["norms"]))
    json.dump(recs, open(res_path, "w"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--topk-only":
        return topk_only(sys.argv[2], sys.argv[3])
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(out_dir, exist_ok=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import implicit_amd.gpu as gpu
    if not gpu.HAS_CUDA:
        sys.exit("parity_report: no HIP device")
    from implicit_amd.synthetic import SHAPES, grid_shards, named
    from oracle import oracle, ref

    oracle.build()
    have_ref = ref.load()[0] is not None
    t_start = time.time()
    cg, topk = [], []

    # ---- configs[2]: last.fm-360K shape, f = 128 -- cold-ish random state and a trained state -----------------------------
    C = named("lastfm360k")
    Ct = C.T.tocsr()
    f, reg = 128, 0.01
    rng = np.random.default_rng(7)
    X0 = rng.random((C.shape[0], f), dtype=np.float32) * 0.2 - 0.1
    Y0 = rng.random((C.shape[1], f), dtype=np.float32) * 0.2 - 0.1
    r, X1 = cg_side(gpu, oracle, "configs[2] (random state +-0.1)", "user rows", C, X0, Y0, reg)
    cg.append(r)
    r, _ = cg_side(gpu, oracle, "configs[2] (random state +-0.1)", "item rows", Ct, Y0, X1, reg)
    cg.append(r)
    # trained state: 5 ALS iterations from the default cold start on the GPU
    solver, gram = gpu.LeastSquaresSolver(), gpu.Matrix.zeros(f, f)
    Xd = gpu.Matrix(rng.random((C.shape[0], f), dtype=np.float32) * 0.01)
    Yd = gpu.Matrix(rng.random((C.shape[1], f), dtype=np.float32) * 0.01)
    Cd, Ctd = gpu.CSRMatrix(C), gpu.CSRMatrix(Ct)
    for _ in range(5):
        solver.calculate_yty(Yd, gram, reg)
        solver.least_squares(Cd, Xd, gram, Yd, 3)
        solver.calculate_yty(Xd, gram, reg)
        solver.least_squares(Ctd, Yd, gram, Xd, 3)
    Xt, Yt = Xd.to_numpy(), Yd.to_numpy()
    del Cd, Ctd
    r, X2 = cg_side(gpu, oracle, "configs[2] (after 5 iterations)", "user rows", C, Xt, Yt, reg)
    cg.append(r)
    r, Y2 = cg_side(gpu, oracle, "configs[2] (after 5 iterations)", "item rows", Ct, Yt, X2, reg)
    cg.append(r)
    # top-k k = 10 over the trained item factors, 2000 user queries
    q = np.arange(0, X2.shape[0], X2.shape[0] // 2000)[:2000]
    topk_cases = [{"name": "configs[2] recommend k=10 (trained factors)", "items": Y2, "queries": np.ascontiguousarray(X2[q]), "k": 10, "norms": False}]
    del C, Ct

    # ---- configs[1]: 1M x 100K, f = 64 -- Cholesky (user side) and CG both sides -------------------------------------------
    C = named("c2")
    f = 64
    rng = np.random.default_rng(3)
    X0 = rng.random((C.shape[0], f), dtype=np.float32) * 0.2 - 0.1
    Y0 = rng.random((C.shape[1], f), dtype=np.float32) * 0.2 - 0.1
    cg.append(cholesky_side(gpu, oracle, "configs[1]", "user rows", C, Y0, reg))
    r, X1 = cg_side(gpu, oracle, "configs[1]", "user rows", C, X0, Y0, reg)
    cg.append(r)
    Ct = C.T.tocsr()
    del C
    r, _ = cg_side(gpu, oracle, "configs[1]", "item rows", Ct, Y0, X1, reg, n_uniform=1000)
    cg.append(r)
    del Ct

    # ---- configs[4]: ml-20m shape, f = 256 + similar_items k = 100 ---------------------------------------------------------
    C = named("ml20m")
    f = 256
    rng = np.random.default_rng(9)
    X0 = rng.random((C.shape[0], f), dtype=np.float32) * 0.2 - 0.1
    Y0 = rng.random((C.shape[1], f), dtype=np.float32) * 0.2 - 0.1
    r, X1 = cg_side(gpu, oracle, "configs[4]", "user rows", C, X0, Y0, reg, n_uniform=1500)
    cg.append(r)
    r, Y1 = cg_side(gpu, oracle, "configs[4]", "item rows", C.T.tocsr(), Y0, X1, reg, n_uniform=1000)
    cg.append(r)
    qi = np.arange(0, Y1.shape[0], Y1.shape[0] // 500)[:500]
    topk_cases.append({"name": "configs[4] similar_items k=100", "items": Y1, "queries": np.ascontiguousarray(Y1[qi]), "k": 100, "norms": True})
    del C

    # ---- configs[0]: MovieLens-100K shape, f = 16 (the reference's own CPU-runnable case): every row, CG and Cholesky, and the
    # compiled reference timed on ONE thread ------------------------------------------------------------------------------------
    C = named("ml100k")
    Ct = C.T.tocsr()
    f = 16
    rng = np.random.default_rng(5)
    X0 = rng.random((C.shape[0], f), dtype=np.float32) * 0.2 - 0.1
    Y0 = rng.random((C.shape[1], f), dtype=np.float32) * 0.2 - 0.1
    r, X1 = cg_side(gpu, oracle, "configs[0]", "user rows", C, X0, Y0, reg, n_uniform=C.shape[0], n_longest=1, n_fp64=C.shape[0])
    cg.append(r)
    r, _ = cg_side(gpu, oracle, "configs[0]", "item rows", Ct, Y0, X1, reg, n_uniform=Ct.shape[0], n_longest=1, n_fp64=Ct.shape[0])
    cg.append(r)
    cg.append(cholesky_side(gpu, oracle, "configs[0]", "user rows", C, Y0, reg, n_uniform=C.shape[0], n_longest=1))
    config0 = None
    als_ref, _ = ref.load()
    if als_ref is not None:
        from threadpoolctl import threadpool_limits

        with threadpool_limits(1, "blas"):
            Xh, Yh = X0.copy(), Y0.copy()
            t0, reps = time.perf_counter(), 0
            while reps < 5 or time.perf_counter() - t0 < 1.0:
                als_ref.least_squares_cg(C, Xh, Yh, reg, num_threads=1, cg_steps=3)
                als_ref.least_squares_cg(Ct, Yh, Xh, reg, num_threads=1, cg_steps=3)
                reps += 1
            t_cg = (time.perf_counter() - t0) / reps
            t0, reps = time.perf_counter(), 0
            while reps < 5 or time.perf_counter() - t0 < 1.0:
                als_ref.least_squares(C, Xh, Yh, reg, num_threads=1)
                als_ref.least_squares(Ct, Yh, Xh, reg, num_threads=1)
                reps += 1
            t_ch = (time.perf_counter() - t0) / reps
        config0 = {"users": int(C.shape[0]), "items": int(C.shape[1]), "nnz": int(C.nnz), "factors": f,
                   "reference_1thread_ms_per_iteration": {"cg_3": 1e3 * t_cg, "cholesky": 1e3 * t_ch},
                   "reference_1thread_updates_per_s": {"cg_3": (C.shape[0] + C.shape[1]) / t_cg, "cholesky": (C.shape[0] + C.shape[1]) / t_ch}}
    del C, Ct

    # ---- configs[3] from a TRAINED state: three ALS iterations over the whole matrix on this one GPU, then rank 0's rows -------
    trained3 = os.environ.get("IMP_PARITY_SKIP_C4_TRAINED") is None
    if trained3:
        users, items, nnz, gamma = SHAPES["c4"]
        Cui_all, Ciu_all, _, _ = grid_shards(0, 1, users, items, nnz, 8, gamma=gamma, seed=42)
        f = 128
        Xd = gpu.RandomState(7).uniform(users, f, 0.0, 0.01)
        Yd = gpu.RandomState(8).uniform(items, f, 0.0, 0.01)
        solver, gram = gpu.LeastSquaresSolver(), gpu.Matrix.zeros(f, f)
        Cd, Ctd = gpu.CSRMatrix(Cui_all), gpu.CSRMatrix(Ciu_all)
        for _ in range(3):
            solver.calculate_yty(Yd, gram, reg)
            solver.least_squares(Cd, Xd, gram, Yd, 3)
            solver.calculate_yty(Xd, gram, reg)
            solver.least_squares(Ctd, Yd, gram, Xd, 3)
        del Cd, Ctd
        Xh, Yh = Xd.to_numpy(), Yd.to_numpy()
        nu, ni = users // 8, items // 8
        r, _ = cg_side(gpu, oracle, "configs[3] (rank 0 of 8, after 3 iterations)", "user rows", Cui_all[:nu], np.ascontiguousarray(Xh[:nu]), Yh, reg,
                       n_uniform=500, n_longest=16, n_fp64=200, Yd=Yd)
        cg.append(r)
        r, _ = cg_side(gpu, oracle, "configs[3] (rank 0 of 8, after 3 iterations)", "item rows", Ciu_all[:ni], np.ascontiguousarray(Yh[:ni]), Xh, reg,
                       n_uniform=500, n_longest=16, n_fp64=200, Yd=Xd)
        cg.append(r)
        del Cui_all, Ciu_all, Xd, Yd, Xh, Yh

    # ---- configs[3]: rank 0's eighth of 10M x 1M x 500M ---------------------------------------------------------------------
    users, items, nnz, gamma = SHAPES["c4"]
    Cui, Ciu, u_off, i_off = grid_shards(0, 8, users, items, nnz, 8, gamma=gamma, seed=42)
    f = 128
    X = gpu.RandomState(7).uniform(users, f, -0.1, 0.1)
    Y = gpu.RandomState(8).uniform(items, f, -0.1, 0.1)
    Xh, Yh = X.to_numpy(), Y.to_numpy()
    r, _ = cg_side(gpu, oracle, "configs[3] (rank 0 of 8)", "user rows", Cui, Xh[int(u_off[0]):int(u_off[1])], Yh, reg, n_uniform=500,
                   n_longest=16, n_fp64=200, Yd=Y)
    cg.append(r)
    r, _ = cg_side(gpu, oracle, "configs[3] (rank 0 of 8)", "item rows", Ciu, Yh[int(i_off[0]):int(i_off[1])], Xh, reg, n_uniform=500,
                   n_longest=16, n_fp64=200, Yd=X)
    cg.append(r)

    forms = topk_forms(out_dir, topk_cases)
    topk = forms.get(TOPK_FORMS[0][0]) if isinstance(forms.get(TOPK_FORMS[0][0]), list) else []
    result = {"solver_rows": cg, "topk": topk, "topk_by_gemm_form": forms, "configs0": config0, "compiled_reference_present": have_ref,
              "seconds": time.time() - t_start}
    json.dump(result, open(os.path.join(out_dir, "parity.json"), "w"), indent=1)
    lines = ["# PARITY — measured distances of the HIP path (generated by `profiles/parity_report.py` on an MI355X box)", "",
             "Relative Frobenius distance over the sampled rows of ONE half sweep from identical inputs (uniform row sample + the longest "
             "rows; fp64 = the same row solve in float64 on a sub-sample).  `reference` = the compiled `implicit/cpu/_als.pyx` / `topk.pyx` "
             "(`oracle/_ref`), `oracle` = `oracle/als_oracle.c`.  north_star's bar: 1e-4 against the reference's Cython path; where the "
             "fp32 reference itself is further than that from the float64 answer, its own distance is the yardstick "
             "(`tests/test_gpu_fullsize.py`).", "",
             "| config | side | rows | nnz | f | max row | gpu–oracle | gpu–reference | oracle–reference | gpu–fp64 | oracle–fp64 | reference–fp64 |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in cg:
        lines.append(f"| {r['config']} | {r['side']} | {r['rows']:,} | {r['nnz']:,} | {r['factors']} | {r['max_row_nnz']:,} | "
                     f"{fmt(r['gpu_vs_oracle'])} | {fmt(r['gpu_vs_reference'])} | {fmt(r['oracle_vs_reference'])} | {fmt(r['gpu_vs_fp64'])} | "
                     f"{fmt(r['oracle_vs_fp64'])} | {fmt(r['reference_vs_fp64'])} |")
    over = [r for r in cg if r["gpu_vs_reference"] is not None and r["gpu_vs_reference"] > 1e-4]
    lines += ["", "Rows of the table above where gpu–reference exceeds 1e-4: " +
              ("none." if not over else "; ".join(f"{r['config']} {r['side']}: {r['gpu_vs_reference']:.1e} (reference–fp64 "
                                                  f"{fmt(r['reference_vs_fp64'])}, gpu–fp64 {fmt(r['gpu_vs_fp64'])})" for r in over) + "."), "",
              "## Top-k ids", "",
              "An id position differs when the GPU and the other side put different items at the same rank of a query's list.  Every "
              "difference is re-scored in float64: a *near-tie* has the two items' exact scores within 4·f·2⁻²³ relative (the noise of an "
              "f-term fp32 dot product under a different summation order); anything else is a real disagreement.", "",
              "| workload | items | queries | k | vs | positions differing | rows affected | near-ties (fp64) | real disagreements |",
              "|---|---|---|---|---|---|---|---|---|"]
    for r in topk:
        for label in ("oracle", "reference"):
            v = r.get(f"vs_{label}")
            if v is None:
                continue
            lines.append(f"| {r['config']} | {r['items']:,} | {r['queries']:,} | {r['k']} | {label} | {v['id_positions_differing']} of {v['of']} "
                         f"({100 * v['fraction']:.3f} %) | {v['rows_with_a_difference']} | {v['near_ties_fp64']} | {v['real_disagreements']} |")
    lines += ["", "### By form of the scoring GEMM (same factors, same queries; differences against the COMPILED REFERENCE)", "",
              "| workload | GEMM form | positions differing | near-ties (fp64) | real disagreements | max score rel. diff vs oracle |",
              "|---|---|---|---|---|---|"]
    for label, recs in forms.items():
        if not isinstance(recs, list):
            lines.append(f"| — | {label} | failed: {recs.get('error', '')[:80]} | | | |")
            continue
        for r in recs:
            v = r.get("vs_reference") or r.get("vs_oracle")
            lines.append(f"| {r['config']} | {label} | {v['id_positions_differing']} of {v['of']} | {v['near_ties_fp64']} | "
                         f"{v['real_disagreements']} | {r['score_rel_max']:.1e} |")
    if config0:
        c0 = config0
        lines += ["", "## configs[0] (MovieLens-100K shape, f = 16): the reference on ONE CPU thread", "",
                  f"{c0['users']} x {c0['items']}, {c0['nnz']:,} nnz.  Compiled reference, `num_threads=1`, BLAS threads = 1: CG (3 steps) "
                  f"{c0['reference_1thread_ms_per_iteration']['cg_3']:.2f} ms per iteration = "
                  f"{c0['reference_1thread_updates_per_s']['cg_3']:,.0f} updates/s; Cholesky "
                  f"{c0['reference_1thread_ms_per_iteration']['cholesky']:.2f} ms = {c0['reference_1thread_updates_per_s']['cholesky']:,.0f} "
                  "updates/s.  GPU parity for every row of both sides is in the table above (rows `configs[0]`); its GPU timing is "
                  "`bench.py`'s `c1_cg` / `c1_cholesky` (launch-latency bound: a 1 ms problem)."]
    lines += ["", f"Generated in {result['seconds']:.0f} s; compiled reference present: {have_ref}."]
    open(os.path.join(out_dir, "PARITY.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
