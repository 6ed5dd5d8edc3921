# randomised shapes through KnnQuery.topk (emit / materialising / general paths, filters, norms, fp16, repeated calls on one handle)
# judged in float64: the returned ids' exact scores must be the best k (up to fp32 near-ties) and the returned scores must match them
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
import os
import implicit_amd._libpath as _lp
if os.environ.get("IMP_LIB_PATH"): _lp.OVERRIDE = os.environ["IMP_LIB_PATH"]
import implicit_amd.gpu as gpu
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
knn = gpu.KnnQuery()
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    f = int(rng.choice([8, 16, 24, 32, 50, 64, 100, 128, 192, 256, 320]))
    ni = int(rng.choice([300, 5000, 20000, 70000, 150000]))
    nq = int(rng.choice([1, 7, 64, 300, 1100]))
    k = int(rng.choice([1, 5, 10, 37, 100, 300]))
    k = min(k, ni)
    kind = rng.choice(["normal", "positive", "lognorm", "lowrank"])
    if kind == "normal":
        items = rng.standard_normal((ni, f)) * 0.1; q = rng.standard_normal((nq, f)) * 0.1
    elif kind == "positive":
        items = rng.random((ni, f)) * 0.01 + 0.005; q = rng.random((nq, f)) * 0.01 + 0.005
    elif kind == "lognorm":
        items = rng.standard_normal((ni, f)) * 0.05 * rng.lognormal(0, 1.0, (ni, 1)); q = rng.standard_normal((nq, f)) * 0.1
    else:
        r = max(1, f // 8); items = rng.standard_normal((ni, r)) @ rng.standard_normal((r, f)) * 0.05; q = rng.standard_normal((nq, r)) @ rng.standard_normal((r, f)) * 0.05
    dt = np.float16 if rng.random() < 0.2 else np.float32
    items = items.astype(dt); q = q.astype(dt)
    use_norms = rng.random() < 0.3
    use_coo = rng.random() < 0.5 and nq > 1
    use_items = rng.random() < 0.3
    I64, Q64 = items.astype(np.float64), q.astype(np.float64)
    S = Q64 @ I64.T
    norms = None
    if use_norms:
        norms = np.linalg.norm(items.astype(np.float32), axis=1).astype(np.float32); norms[norms == 0] = 1e-10
        S = S / norms[None, :].astype(np.float64)
    kw = {}
    if use_norms: kw["item_norms"] = gpu.Matrix(norms.reshape(1, -1))
    if use_coo:
        liked = sp.random(nq, ni, density=min(0.5, 20.0 / ni), format="csr", dtype=np.float32, random_state=int(rng.integers(1 << 30)))
        kw["query_filter"] = gpu.COOMatrix.from_csr_pattern(liked) if rng.random() < 0.5 else gpu.COOMatrix(liked.tocoo())
        S[liked.nonzero()] = -np.inf
    if use_items:
        filt = np.unique(rng.integers(0, ni, size=max(1, ni // 50))).astype(np.int32)
        kw["item_filter"] = gpu.IntVector(filt); S[:, filt] = -np.inf
    handle = knn if rng.random() < 0.7 else gpu.KnnQuery()
    ids, d = handle.topk(gpu.Matrix(items), gpu.Matrix(q), k, **kw)
    avail = np.isfinite(S).sum(axis=1)
    ok = True
    for r in range(nq):
        kk = int(min(k, avail[r]))
        best = -np.sort(-S[r])[:kk]
        got = S[r, ids[r, :kk].astype(np.int64)]
        scale = np.abs(Q64[r]) @ np.abs(I64).max(axis=0) + 1e-300
        tol = 8 * f * np.finfo(np.float32).eps * scale / (norms.min() if use_norms else 1.0) if False else 4e-6 * (np.abs(best) + 1e-30) + 16 * f * 6e-8 * np.abs(Q64[r]) @ np.abs(I64).mean(axis=0) / (np.median(norms) if use_norms else 1.0)
        if not (np.isfinite(got).all() and (np.abs(np.sort(got)[::-1] - best) <= tol).all() and len(set(ids[r, :kk])) == kk):
            ok = False
            dev = np.abs(np.sort(got)[::-1] - best); w = int(np.argmax(dev - tol))
            print("  row", r, "rank", w, "got", np.sort(got)[::-1][max(0, w - 1):w + 2], "best", best[max(0, w - 1):w + 2], "tol", (tol if np.isscalar(tol) else tol[w]),
                  "finite", np.isfinite(got).all(), "distinct", len(set(ids[r, :kk])) == kk, "next best", -np.sort(-S[r])[kk:kk + 2],
                  "missing ids", sorted(set(np.argsort(-S[r])[:kk].tolist()) - set(ids[r, :kk].tolist()))[:5], "of ni", ni)
            break
        rt = 2e-3 if dt == np.float16 else 4e-5
        if not np.allclose(d[r, :kk], got, rtol=rt, atol=tol.max() if hasattr(tol, "max") else tol):
            ok = False; print("  row", r, "scores", d[r, :4], "exact", got[:4]); break
    print(f"trial {trial}: ni={ni} f={f} nq={nq} k={k} {kind} {dt.__name__} norms={use_norms} coo={use_coo} items={use_items} ->", "ok" if ok else "MISMATCH")
    bad += not ok
print("mismatches:", bad)
