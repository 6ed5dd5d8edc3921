"""CPU model of the operand arithmetic of the normal-matrix kernels (implicit_amd/csrc/als_cg_nm.hip), so that its precision and
range claims are checked where no GPU is.  fp32 factors (round 5): the weight w = |c| - 1 enters both operands as sqrt|w| (one
split per gathered value, the sign of w as a bit flip of one operand) -- `root_products` below.  fp16 factor storage (and round 4
for both): the weight w = |c| - 1 is dealt to the two matrix-core operands as w 2^-e and 2^e
(e = floor(exponent(|w|) / 2), clamped to [-12, 24]; w carries the launch's operand scale 4^k), each operand is split into two fp16 halves (round to nearest), and a product
keeps the three terms  u_h y_h + u_l y_h + u_h y_l  with fp32 accumulation.  The numpy code below restates nm_build's produce
phase bit for bit (same integer expression for the exponent, same conversions); the GPU parity of the whole kernel is
tests/test_gpu_nm.py.
"""
import numpy as np


def operand_scale(max_diag, y_rows):
    """k of nm_gram_image_kernel: brings the rms of the largest factor column (gramian diagonal / rows of Y) to about 8; 0 .. 16."""
    rms = np.sqrt(np.float32(max_diag) / np.float32(max(y_rows, 1)))
    if not (rms > 0 and rms < 8):
        return 0
    return int(min(np.floor(np.log2(np.float32(8.0) / rms)), 16))


def deal_weight(w, k=0):
    """(wa, sb) with wa * sb == 4^k w exactly: sb = 2^e, e = floor((exponent of |4^k w|) / 2) clamped to [-12, 24]."""
    w = np.asarray(w, np.float32) * np.float32(4.0 ** k)
    bits = w.view(np.uint32)
    hb = ((((bits & np.uint32(0x7F800000)).astype(np.uint64) + (127 << 23)) >> 1) & 0x7F800000).astype(np.uint32)
    hb = np.minimum(np.maximum(hb, np.uint32((127 - 12) << 23)), np.uint32((127 + 24) << 23))
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


def test_operand_scale_restores_the_bits_of_small_factors():
    """Cold-start factors (0 .. 0.01, implicit/gpu/als.py:98-101) and trained factors of 1e-3: without the launch's scale the low
    halves are fp16 subnormals and an operand keeps 14 .. 17 bits; with it (k from the gramian's diagonal) the products are good to
    2^-20 of themselves again.  The scale is exact: the image is multiplied by 4^-k where it is written."""
    rng = np.random.default_rng(3)
    for scale in (0.01, 1e-3, 1e-4):
        items = 50_000
        y_i = (rng.random(items) * scale).astype(np.float32)
        y_j = (rng.random(items) * scale).astype(np.float32)
        k = operand_scale(float((y_i.astype(np.float64) ** 2).sum()) + 0.01, items)   # diagonal incl. the regularisation
        assert 1 <= k <= 16
        w = (rng.random(items) * 4).astype(np.float32)
        exact = w.astype(np.float64) * y_i * y_j
        rel = {}
        for kk in (0, k):
            wa, sb = deal_weight(w, kk)
            u, v = wa * y_i, sb * y_j
            assert np.isfinite(u.astype(np.float16)).all() and np.isfinite(v.astype(np.float16)).all()
            got = three_products(u, v) * 4.0 ** -kk
            big = np.abs(exact) > 1e-3 * np.abs(exact).max()
            rel[kk] = (np.abs(got - exact)[big] / np.abs(exact)[big]).max()
        print("scale %g: k = %d, worst product rel %.1e -> %.1e" % (scale, k, rel[0], rel[k]))
        # (at 1e-4 the regularisation dominates the diagonal the scale is taken from -- and the matrix the products are added to)
        assert rel[k] < (2.0 ** -17 if scale >= 1e-3 else 2.0 ** -13) and rel[k] < 1e-2 * rel[0]


def test_operands_beyond_the_fp16_range_become_infinite_not_wrong():
    """What the kernel's finiteness check relies on: an operand past 65504 converts to an fp16 infinity (never to a finite wrong
    number), and its low half to the opposite infinity or NaN -- every product it takes part in is then non-finite."""
    u = np.float32(70000.0)
    with np.errstate(over="ignore", invalid="ignore"):
        h, l = split(u)
        assert np.isinf(h) and not np.isfinite(l)
        assert not np.isfinite(three_products(u, np.float32(0.5)))
        assert not np.isfinite(three_products(u, np.float32(0.0)))


def root_products(w, y_i, y_j, k=0):
    """nm_build's fp32-storage form: z = sqrt(|w| 4^k) y split once, u = sign(w) z; three products, scaled back by 4^-k."""
    w = np.asarray(w, np.float32) * np.float32(4.0 ** k)
    sq = np.sqrt(np.abs(w)).astype(np.float32)            # v_sqrt_f32: 1 ulp
    zi, zj = sq * np.asarray(y_i, np.float32), sq * np.asarray(y_j, np.float32)
    return np.sign(w).astype(np.float64) * three_products(zi, zj) * 4.0 ** -k


def test_root_form_matches_the_dealt_form():
    """One split per value instead of two: the product carries w to 2^-23 (the square root's ulp twice) on top of the halves' 2^-22,
    for positive, negative (confidences below one: -1 < w < 0) and zero weights."""
    rng = np.random.default_rng(5)
    n = 200_000
    y_i = (rng.standard_normal(n) * 0.1).astype(np.float32)
    y_j = (rng.standard_normal(n) * 0.1).astype(np.float32)
    w = np.concatenate([rng.random(n - 1000) * 40, -rng.random(999), [0.0]]).astype(np.float32)
    k = operand_scale(float((y_i.astype(np.float64) ** 2).sum() / 100), n // 100)
    got = root_products(w, y_i, y_j, k)
    exact = w.astype(np.float64) * y_i * y_j
    big = np.abs(exact) > 1e-4 * np.abs(exact).max()
    rel = np.abs(got - exact)[big] / np.abs(exact)[big]
    print("root form: k = %d, worst rel %.2e, mean %.2e" % (k, rel.max(), rel.mean()))
    assert rel.max() < 2.0 ** -17 and rel.mean() < 2.0 ** -20
    assert got[-1] == 0.0 and (np.sign(got[:-1]) == np.sign(exact[:-1]))[np.abs(exact[:-1]) > 0].all()
    assert abs(got.sum() - exact.sum()) / np.abs(exact).sum() < 2e-7   # unbiased over a row


def chol_block_of(bid, nb=32):
    """nm_chol's ownership map (als_cg_nm.hip): block id -> (block row, block column) of the lower triangle's 4 x 4 blocks, numbered
    by block COLUMN from the right, top to bottom inside a column -- the same float expression and fix-up loops as the kernel."""
    t = int((np.sqrt(np.float32(8.0) * np.float32(bid) + np.float32(1.0)) - np.float32(1.0)) * np.float32(0.5))
    while t * (t + 1) // 2 > bid:
        t -= 1
    while (t + 1) * (t + 2) // 2 <= bid:
        t += 1
    bn = nb - 1 - t
    return bn + (bid - t * (t + 1) // 2), bn


def test_cholesky_block_numbering_covers_the_triangle_and_keeps_live_blocks_a_prefix():
    """f = 128: 528 blocks over 256 threads x 3 slots.  Every block of the lower triangle is owned exactly once, and the blocks
    still to be updated at turn kb (block columns > kb) are exactly the ids below (31 - kb)(32 - kb) / 2 -- the first slot of the
    first threads, which is what lets a wavefront run ceil(live / 256) passes of the trailing update per turn."""
    nb = 32
    n_blocks = nb * (nb + 1) // 2
    owned = [chol_block_of(b, nb) for b in range(n_blocks)]
    assert len(set(owned)) == n_blocks
    assert all(0 <= bn <= bm < nb for bm, bn in owned)
    for kb in range(nb):
        live = [b for b, (bm, bn) in enumerate(owned) if bn > kb]
        n_live = (nb - 1 - kb) * (nb - kb) // 2
        assert live == list(range(n_live))
    # inside a block column the blocks are consecutive ids, top (the diagonal block) to bottom
    for t in range(nb):
        ids = [b for b, (bm, bn) in enumerate(owned) if bn == nb - 1 - t]
        assert ids == list(range(t * (t + 1) // 2, (t + 1) * (t + 2) // 2))
        assert [owned[b][0] for b in ids] == list(range(nb - 1 - t, nb))
