"""Randomised shapes through KnnQuery.topk, judged in float64 (the check that found the early-flush bug of round 6): whatever
path a shape takes -- screened or three-product emit pass, materialising path, general path, row-by-row exact path -- the ids
returned must be the best k of the float64 scores up to fp32 near-ties, distinct, and the scores returned must be theirs."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _factors(rng, kind, ni, nq, f):
    if kind == "normal":
        return rng.standard_normal((ni, f)) * 0.1, rng.standard_normal((nq, f)) * 0.1
    if kind == "positive":       # the same items best for every query
        return rng.random((ni, f)) * 0.01 + 0.005, rng.random((nq, f)) * 0.01 + 0.005
    if kind == "lognorm":        # heavy-tailed item norms, as trained factors have
        return rng.standard_normal((ni, f)) * 0.05 * rng.lognormal(0, 1.0, (ni, 1)), rng.standard_normal((nq, f)) * 0.1
    r = max(1, f // 8)           # low rank: concentrated scores
    return (rng.standard_normal((ni, r)) @ rng.standard_normal((r, f)) * 0.05,
            rng.standard_normal((nq, r)) @ rng.standard_normal((r, f)) * 0.05)


def _check(gpu, knn, rng, ni, f, nq, k, kind, dt, use_norms, use_coo, use_items):
    items, q = _factors(rng, kind, ni, nq, f)
    items, q = items.astype(dt), q.astype(dt)
    I64, Q64 = items.astype(np.float64), q.astype(np.float64)
    S = Q64 @ I64.T
    kw, norms = {}, None
    if use_norms:
        norms = np.linalg.norm(items.astype(np.float32), axis=1).astype(np.float32)
        norms[norms == 0] = 1e-10
        S = S / norms[None, :].astype(np.float64)
        kw["item_norms"] = gpu.Matrix(norms.reshape(1, -1))
    if use_coo:
        liked = sp.random(nq, ni, density=min(0.5, 20.0 / ni), format="csr", dtype=np.float32, random_state=int(rng.integers(1 << 30)))
        kw["query_filter"] = gpu.COOMatrix.from_csr_pattern(liked) if rng.random() < 0.5 else gpu.COOMatrix(liked.tocoo())
        S[liked.nonzero()] = -np.inf
    if use_items:
        filt = np.unique(rng.integers(0, ni, size=max(1, ni // 50))).astype(np.int32)
        kw["item_filter"] = gpu.IntVector(filt)
        S[:, filt] = -np.inf
    ids, d = knn.topk(gpu.Matrix(items), gpu.Matrix(q), k, **kw)
    avail = np.isfinite(S).sum(axis=1)
    noise = 16 * f * 6e-8 * (np.abs(Q64) @ np.abs(I64).mean(axis=0)) / (np.median(norms) if use_norms else 1.0)
    for r in range(nq):
        kk = int(min(k, avail[r]))
        best = -np.sort(-S[r])[:kk]
        got = S[r, ids[r, :kk].astype(np.int64)]
        tol = 4e-6 * (np.abs(best) + 1e-30) + noise[r]
        assert np.isfinite(got).all() and len(set(ids[r, :kk].tolist())) == kk, (r, ids[r, :8])
        assert (np.abs(np.sort(got)[::-1] - best) <= tol).all(), (r, np.sort(got)[::-1][:4], best[:4])
        assert np.allclose(d[r, :kk], got, rtol=2e-3 if dt == np.float16 else 4e-5, atol=float(tol.max()))


def test_a_step_that_outruns_the_staging_headroom_loses_nothing(gpu):
    """The shape the fuzz run caught: k = 100 over heavy-tailed fp16 factors -- an item that passes for every row of a query
    block stages 256 entries in one step; the early flush of the staging buffer once forgot the entries such a step had to drop."""
    for seed in (1234, 2, 77):
        _check(gpu, gpu.KnnQuery(), np.random.default_rng(seed), 150_000, 32, 1100, 100, "lognorm", np.float16, False, True, False)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_shapes_against_float64(gpu, seed):
    rng = np.random.default_rng(seed)
    knn = gpu.KnnQuery()
    for _ in range(10):
        f = int(rng.choice([8, 24, 32, 64, 100, 128, 256, 320]))
        ni = int(rng.choice([300, 5000, 20000, 70000, 150000]))
        nq = int(rng.choice([1, 7, 64, 300]))
        k = min(int(rng.choice([1, 5, 10, 37, 100, 300])), ni)
        kind = str(rng.choice(["normal", "positive", "lognorm", "lowrank"]))
        dt = np.float16 if rng.random() < 0.2 else np.float32
        handle = knn if rng.random() < 0.7 else gpu.KnnQuery()   # (a shared handle: cached planes, the adaptive pre-pass stride)
        _check(gpu, handle, rng, ni, f, nq, k, kind, dt, rng.random() < 0.3, rng.random() < 0.5 and nq > 1, rng.random() < 0.3)
