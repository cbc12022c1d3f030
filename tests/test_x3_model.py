"""The arithmetic model behind the split-fp16 ("f16x3") kernels, checked on the CPU with numpy's IEEE fp16: what
mscnn_amd/csrc/x3_device.h (pow2_scale, split16) and the three-MFMA product promise, independent of any GPU.  The kernels
themselves are held to the fp32 gates in tests/test_gpu_ops.py / test_gpu_net.py."""
import numpy as np
import pytest


def pow2_scale(bound):
    """x3_device.h::pow2_scale: s = 2^(14 - floor(log2(bound))), exponent clamped to [-100, 110]; bound 0 -> 1."""
    if bound == 0:
        return 1.0
    e = int(np.floor(np.log2(np.float32(bound))))
    e = max(-100, min(110, e))
    return float(2.0 ** (14 - e))


def split16(v):
    v = np.asarray(v, np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


@pytest.mark.parametrize("amax", [1e-30, 3e-5, 0.7, 1.0, 255.0, 6.5e4, 3.1e7, 1e30])
def test_scale_keeps_the_tensor_inside_fp16_and_is_exact(amax):
    s = pow2_scale(amax)
    assert 2.0 ** 14 <= np.float32(amax) * np.float32(s) < 2.0 ** 15          # bound * s in [2^14, 2^15): never overflows 65504
    rng = np.random.default_rng(1)
    x = (rng.uniform(-1, 1, 4096) * amax).astype(np.float32)
    xs = x * np.float32(s)
    assert np.array_equal(xs * np.float32(1.0 / s), x)                         # power of two: scaling and un-scaling are exact
    hi, lo = split16(xs)
    assert np.isfinite(hi.astype(np.float32)).all()


def test_split_is_exact_to_22_bits():
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-1, 1, 200000) * 2.0 ** 15, rng.uniform(-1, 1, 200000) * 2.0 ** rng.integers(-30, 15, 200000)]).astype(np.float32)
    hi, lo = split16(x)
    err = np.abs(x.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
    bound = np.maximum(2.0 ** -22 * np.abs(x.astype(np.float64)), 2.0 ** -25)
    assert (err <= bound).all(), float((err / bound).max())


@pytest.mark.parametrize("K", [128, 512, 4608])
def test_three_products_are_fp32_grade(K):
    """hi*hi + hi*lo + lo*hi with fp32 accumulation against float64: the same order of error as an fp32 GEMM, far below fp16."""
    rng = np.random.default_rng(K)
    x = np.maximum(rng.standard_normal((K, 256)), 0).astype(np.float32) * 3
    w = (rng.standard_normal((64, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    truth = w.astype(np.float64) @ x.astype(np.float64)
    sx, sw = pow2_scale(np.abs(x).max()), pow2_scale(np.abs(w).max())
    xh, xl = split16(x * np.float32(sx))
    wh, wl = split16(w * np.float32(sw))
    f = lambda a: a.astype(np.float32)                                           # noqa: E731  (fp16 x fp16 is exact in fp32)
    y3 = (f(wl) @ f(xh) + f(wh) @ f(xl) + f(wh) @ f(xh)) * np.float32(1.0 / (sx * sw))
    y1 = (f(wh) @ f(xh)) * np.float32(1.0 / (sx * sw))
    metric = lambda y: float((np.abs(y - truth) / np.maximum(1, np.abs(truth))).max())   # noqa: E731
    e3, e32, e16 = metric(y3), metric(w @ x), metric(y1)
    assert e3 < 1e-5 and e3 <= 4 * e32 + 1e-6, (e3, e32)
    assert e16 > 50 * e3                                                          # what a single fp16 product would give
    assert np.array_equal((f(wh) @ f(np.zeros_like(xh))), np.zeros((64, 256), np.float32))
