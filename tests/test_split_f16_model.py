"""The arithmetic claim behind LCRC_ARITH_SPLIT_F16, checked on the CPU with numpy: an f32 product
evaluated as three exact f16 x f16 products (high x high, high x low, low x high) of (high, low) operand pairs is as
close to the exact result as the f32 product chain itself -- and the two cheaper splits one might try are not."""
import numpy as np

from phnrec_amd import modelgen


def _f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def _bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def _pairs(x, rnd):
    hi = rnd(x)
    return hi, rnd((np.asarray(x, np.float32) - hi).astype(np.float32))


def _split_product(a, b, rnd, terms):
    """sum of the chosen (piece of a) x (piece of b) products, each exact, summed in f64 (the MFMA's f32 accumulation
    is common to every variant and to the f32 kernels)"""
    pa, pb = _pairs(a, rnd), _pairs(b, rnd)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float64)
    for i, j in terms:
        acc += pa[i].astype(np.float64) @ pb[j].astype(np.float64)
    return acc


def test_three_f16_products_match_f32_accuracy():
    rng = np.random.default_rng(5)
    net = modelgen.random_net(rng, 165, 400, 138, "band")
    x = (rng.standard_normal((64, 165)) * 2.0).astype(np.float32)
    w1 = net["w1"].T.copy()
    exact = x.astype(np.float64) @ w1.astype(np.float64)
    scale = np.abs(exact).max()
    err_f32 = np.abs(np.float32(x) @ np.float32(w1) - exact).max() / scale
    three = [(0, 0), (0, 1), (1, 0)]
    err_split = np.abs(_split_product(x, w1, _f16, three) - exact).max() / scale
    err_hi_only = np.abs(_split_product(x, w1, _f16, [(0, 0)]) - exact).max() / scale
    err_bf16 = np.abs(_split_product(x, w1, _bf16, three) - exact).max() / scale
    assert err_split < 4e-7                       # 2^-22 operand tails, random signs over 165 terms
    assert err_split < 2.0 * err_f32 + 1e-7       # the size of the f32 chain's own rounding
    assert err_hi_only > 50 * err_split           # one f16 product is not enough
    assert err_bf16 > 10 * err_split              # nor are (high, low) bf16 pairs with three products


def test_pairs_hold_22_bits_and_subnormal_tails():
    rng = np.random.default_rng(6)
    v = (rng.standard_normal(20000) * np.exp(rng.uniform(-8, 8, 20000))).astype(np.float32)
    v = v[np.abs(v) < 60000]
    hi, lo = _pairs(v, _f16)
    big = np.abs(v) > 2.0 ** -3                   # low part normal: relative error 2^-22
    assert (np.abs((hi + lo).astype(np.float64) - v)[big] / np.abs(v[big])).max() <= 2.0 ** -22
    # below that the low part is an f16 subnormal (spacing 2^-24): absolute error <= 2^-25
    assert np.abs((hi + lo).astype(np.float64) - v)[~big].max() <= 2.0 ** -25
    # a product of two f16 values is exact in f32
    a, b = _f16(rng.standard_normal(1000)), _f16(rng.standard_normal(1000))
    assert np.array_equal((a * b).astype(np.float32).astype(np.float64), a.astype(np.float64) * b.astype(np.float64))


def test_power_of_two_scaling_restores_the_split_for_small_operands():
    """What lcrc_set_arithmetic's packer does about the subnormal tails (pack_net_h2, mlp_dev.h "Operand scaling"): a
    1500-term product with weights of ~0.005 and activations in (0, 1).  Split as they stand, the low halves are f16
    subnormals and the error floor is ~1e-6 however small the weights; scaled by powers of two first (largest weight into
    (2^13, 2^14], activations by 2^14) and scaled back exactly, the split is again as good as the f32 chain."""
    rng = np.random.default_rng(7)
    k = 1500
    w = (rng.standard_normal((k, 64)) * 0.005).astype(np.float32)
    s = rng.uniform(0.0, 1.0, (32, k)).astype(np.float32)
    exact = s.astype(np.float64) @ w.astype(np.float64)
    three = [(0, 0), (0, 1), (1, 0)]
    err_f32 = np.abs(np.float32(s) @ np.float32(w) - exact).max()
    err_raw = np.abs(_split_product(s, w, _f16, three) - exact).max()
    e_w = 14 - int(np.floor(np.log2(np.abs(w).max())) + 1)            # max|w| * 2^e in [2^13, 2^14)
    ws, ss = w * np.float32(2.0 ** e_w), s * np.float32(2.0 ** 14)
    assert 2.0 ** 13 <= np.abs(ws).max() < 2.0 ** 14
    err_scaled = np.abs(_split_product(ss, ws, _f16, three) * 2.0 ** -(e_w + 14) - exact).max()
    assert err_raw > 5e-7                       # the floor the unscaled split runs into
    assert err_scaled < err_raw / 20.0          # gone
    assert err_scaled < 2.0 * err_f32 + 1e-9    # back at the f32 chain's own rounding
