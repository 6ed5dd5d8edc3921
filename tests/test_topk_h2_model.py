"""CPU model of the arithmetic of the emit GEMM's fp16 form (implicit_amd/csrc/topk.hip, H2: split8_f16, split_query_rows_f16_kernel,
topk_absmax_kernel, h2_scale_exp), so that its precision and range claims are checked where no GPU is.  Every operand value x is
scaled by a power of two taken from the call's largest magnitude (queries: all rows; items: every 16th row) to [2^11, 2^12), split
into two fp16 halves h + l (round to nearest), and a product keeps  l h + h l + h h  with fp32 accumulation; the scales are taken
out of the accumulators again.  The GPU parity of the kernel itself is tests/test_gpu_topk.py."""
import numpy as np


def scale_exp(max_abs):
    """h2_scale_exp: 11 - exponent(largest magnitude), clamped to +-60 (0 for an all-zero operand is whatever: nothing to scale)."""
    bits = np.float32(max_abs).view(np.uint32)
    return int(max(-60, min(60, 11 - (int(bits >> np.uint32(23)) - 127))))


def split(x):
    x = np.asarray(x, np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        h = x.astype(np.float16).astype(np.float32)
        l = (x - h).astype(np.float16).astype(np.float32)
    return h, l


def h2_scores(items, queries, sample=16):
    kq = scale_exp(np.abs(queries).max())
    ki = scale_exp(np.abs(items[::sample]).max())
    qh, ql = split(queries * np.float32(2.0 ** kq))
    ih, il = split(items * np.float32(2.0 ** ki))
    with np.errstate(over="ignore", invalid="ignore"):
        acc = (ql.astype(np.float64) @ ih.T.astype(np.float64) + qh.astype(np.float64) @ il.T.astype(np.float64)
               + qh.astype(np.float64) @ ih.T.astype(np.float64))
    return (acc * 2.0 ** -(kq + ki)).astype(np.float32)   # (the kernel accumulates in fp32: modelled by the final rounding only)


def test_the_scale_brings_the_largest_magnitude_to_2_11():
    for m in (1e-7, 3e-4, 0.1, 1.0, 7.9, 300.0, 6.0e4, 1e9):
        k = scale_exp(m)
        assert 2.0 ** 11 <= m * 2.0 ** k < 2.0 ** 12
    assert scale_exp(1e30) == -60 and scale_exp(1e-30) == 60   # clamped: what comes out then is what the NaN guard catches


def test_scores_keep_fp32_accuracy_over_the_magnitudes_als_produces():
    rng = np.random.default_rng(2)
    f = 128
    for item_scale, query_scale in ((1.0, 1.0), (1e-4, 30.0), (2e3, 1e-3), (1e-6, 1e-6)):
        items = (rng.standard_normal((4000, f)) * 0.1 * item_scale).astype(np.float32)
        items[::7] *= 1e-3                                   # mixed row sizes inside one matrix
        q = (rng.standard_normal((50, f)) * 0.1 * query_scale).astype(np.float32)
        got = h2_scores(items, q).astype(np.float64)
        exact = q.astype(np.float64) @ items.T.astype(np.float64)
        terms = np.abs(q).astype(np.float64) @ np.abs(items).T.astype(np.float64)   # sum of |q_k y_k|: what rounding is relative to
        err = np.abs(got - exact) / terms
        # operands to 2^-22 each, the dropped l l term 2^-22: a few 1e-7 of the terms' magnitude at worst, far less on average
        assert err.max() < 6e-7 and err.mean() < 1.5e-7, (item_scale, query_scale, err.max(), err.mean())
        top = np.sort(exact, axis=1)[:, -10:]                # the scores a top-10 is decided among: relative to themselves
        rel = np.abs(np.sort(got, axis=1)[:, -10:] - top) / np.abs(top)
        assert rel.max() < 3e-5                              # the bar tests/test_gpu_topk.py puts on the returned distances


def test_an_item_row_far_above_the_sample_turns_every_score_it_touches_into_nan():
    rng = np.random.default_rng(3)
    f = 64
    items = (rng.standard_normal((320, f)) * 0.1).astype(np.float32)
    items[7] *= 1e4                                          # row 7 is not sampled (rows 0, 16, 32 ... are)
    q = (rng.standard_normal((20, f)) * 0.1).astype(np.float32)
    q[3] = 0.0                                               # 0 x inf is a NaN too
    q[5, ::2] = 0.0
    s = h2_scores(items, q)
    assert np.isnan(s[:, 7]).all()                           # never a finite wrong number
    others = np.delete(s, 7, axis=1)
    assert np.isfinite(others).all()


def test_values_inside_the_headroom_do_not_overflow():
    """[2^11, 2^12) for the sampled maximum leaves a factor of 16 .. 32 to 65504 for rows the sample did not see."""
    rng = np.random.default_rng(4)
    items = (rng.standard_normal((640, 32)) * 0.1).astype(np.float32)
    big = np.abs(items[::16]).max()
    items[5, 0] = 15.9 * big
    q = (rng.standard_normal((8, 32)) * 0.1).astype(np.float32)
    assert np.isfinite(h2_scores(items, q)).all()


def test_ranking_matches_float64_outside_near_ties():
    rng = np.random.default_rng(5)
    f, k = 128, 10
    items = (rng.standard_normal((20000, f)) * 0.1).astype(np.float32)
    q = (rng.standard_normal((40, f)) * 0.1).astype(np.float32)
    got = h2_scores(items, q)
    exact = q.astype(np.float64) @ items.T.astype(np.float64)
    ids = np.argsort(-got, axis=1, kind="stable")[:, :k]
    want = np.argsort(-exact, axis=1, kind="stable")[:, :k + 1]
    top = np.take_along_axis(exact, want, axis=1)
    near_tie = (np.abs(np.diff(top, axis=1)) < 4 * np.finfo(np.float32).eps * f * np.abs(top[:, :-1])).any(axis=1)
    assert (~near_tie).mean() > 0.9
    assert np.array_equal(ids[~near_tie], want[~near_tie, :k])
