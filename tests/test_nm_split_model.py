"""CPU model of the operand arithmetic of the normal-matrix kernels (implicit_amd/csrc/als_cg_nm.hip), so that its precision and
range claims are checked where no GPU is: the weight w = |c| - 1 is dealt to the two matrix-core operands as w 2^-e and 2^e
(e = floor(exponent(|w|) / 2), clamped to +-12), each operand is split into two fp16 halves (round to nearest), and a product
keeps the three terms  u_h y_h + u_l y_h + u_h y_l  with fp32 accumulation.  The numpy code below restates nm_build's produce
phase bit for bit (same integer expression for the exponent, same conversions); the GPU parity of the whole kernel is
tests/test_gpu_nm.py.
"""
import numpy as np


def deal_weight(w):
    """(wa, sb) with wa * sb == w exactly: sb = 2^e, e = floor((exponent of |w|) / 2) clamped to [-12, 12]."""
    w = np.asarray(w, np.float32)
    bits = w.view(np.uint32)
    hb = ((((bits & np.uint32(0x7F800000)).astype(np.uint64) + (127 << 23)) >> 1) & 0x7F800000).astype(np.uint32)
    hb = np.minimum(np.maximum(hb, np.uint32((127 - 12) << 23)), np.uint32((127 + 12) << 23))
    sb = hb.view(np.float32)
    wa = w * (np.uint32(254 << 23) - hb).view(np.float32)
    return wa, sb


def split(x):
    """x = h + l with fp16 halves (both returned as float32 values)."""
    x = np.asarray(x, np.float32)
    h = x.astype(np.float16).astype(np.float32)
    l = (x - h).astype(np.float16).astype(np.float32)
    return h, l


def three_products(u, y):
    uh, ul = split(u)
    yh, yl = split(y)
    return uh.astype(np.float64) * yh + ul.astype(np.float64) * yh + uh.astype(np.float64) * yl


def test_the_weight_is_dealt_exactly_and_balanced():
    rng = np.random.default_rng(0)
    w = np.concatenate([rng.random(1000) * 10.0 ** rng.integers(-6, 8, 1000), -rng.random(100), [0.0, 1.0, 2.0 ** 20, 3e7]]).astype(np.float32)
    wa, sb = deal_weight(w)
    assert np.array_equal(wa * sb, w)                       # exact: a power-of-two scaling
    assert (np.log2(sb) == np.round(np.log2(sb))).all()
    inside = (np.abs(w) >= 2.0 ** -24) & (np.abs(w) < 2.0 ** 24)
    assert (np.abs(wa[inside]) < 2.0 * np.sqrt(2.0) * np.sqrt(np.abs(w[inside]))).all()   # both operands ~ sqrt|w|
    assert (sb[inside] <= np.sqrt(np.abs(w[inside])) * 1.0001).all()
    assert deal_weight(np.float32(0.0))[0] == 0.0           # weight 0 (confidence 1, padding): a zero operand, no NaN


def test_precision_of_the_three_products():
    """An operand x is held as h + l with |x - (h + l)| <= max(2^-23 |x|, 2^-25): the low half is an fp16 SUBNORMAL (spacing 2^-24)
    below |x| = 2^-3.  ALS factors of 0.01 .. 0.1 therefore carry 18 .. 21 significant bits per operand; the dropped l * l term
    is 2^-22 of the product.  In absolute terms every product is good to 2^-24 (|u| + |v|)."""
    rng = np.random.default_rng(1)
    y_i = (rng.standard_normal(200_000) * 0.1).astype(np.float32)
    y_j = (rng.standard_normal(200_000) * 0.1).astype(np.float32)
    w = (rng.random(200_000) * 40).astype(np.float32)
    wa, sb = deal_weight(w)
    u, v = wa * y_i, sb * y_j
    got = three_products(u, v)
    exact = w.astype(np.float64) * y_i * y_j
    err = np.abs(got - exact)
    assert (err <= 2.0 ** -24 * (np.abs(u) + np.abs(v)) + 2.0 ** -21 * np.abs(exact)).all()
    normal = (np.abs(u) >= 0.125) & (np.abs(v) >= 0.125)    # both low halves in the fp16 normal range
    rel = err / np.maximum(np.abs(exact), 1e-300)
    print("operands >= 1/8: max rel %.2e (2^-20 = %.2e); all: mean rel %.2e" % (rel[normal].max(), 2.0 ** -20, rel.mean()))
    assert rel[normal].max() < 2.0 ** -20 and rel.mean() < 2.0 ** -19
    # a row's sum of 200 K terms: the errors are unbiased
    assert abs(got.sum() - exact.sum()) / np.abs(exact).sum() < 2e-7


def test_confidences_to_1e7_stay_in_the_fp16_range():
    for conf in (1e3, 1e5, 1e7):
        w = np.float32(conf - 1.0)
        wa, sb = deal_weight(w)
        for y in (1e-4, 0.3, 8.0, 13.0):
            u, v = wa * np.float32(y), sb * np.float32(y)
            assert np.isfinite(np.float16(u)) and np.isfinite(np.float16(v)), (conf, y)
            got = three_products(np.float32(u), np.float32(v))
            exact = float(w) * y * y
            assert abs(got - exact) <= 2.0 ** -24 * (abs(float(u)) + abs(float(v))) + 2.0 ** -21 * exact


def test_tiny_factors_lose_absolute_not_relative_accuracy():
    """|2^e y| below the fp16 normal range: the halves are fp16 subnormals, spacing 2^-24 -- an ABSOLUTE error of at most 2^-25
    per operand, i.e. nothing beside the other terms of a row whose weights are not themselves tiny."""
    y = np.float32(3e-6)
    h, l = split(y)
    assert abs(float(h) + float(l) - float(y)) <= 2.0 ** -25
