"""The oracle's restatements of the MATLAB stages (final detection stage, imresize) against the independent second witnesses
of tests/witness.py, on random inputs and on adversarial ties.  Neither side is MATLAB: agreement of two separately written
routes is what this pins ('parity unpinned' rows)."""
import numpy as np
import pytest

from tests import witness


def _random_outputs(rng, R, ncls=5, img=(576, 1920)):
    x1 = rng.uniform(0, img[1] - 80, R); y1 = rng.uniform(0, img[0] - 60, R)
    w = rng.uniform(8, 300, R); h = rng.uniform(8, 200, R)
    props = np.stack([np.zeros(R), x1, y1, x1 + w, y1 + h, rng.normal(0, 4, R)], 1).astype(np.float32)
    return (rng.normal(0, 0.5, (R, 4 * ncls)).astype(np.float32), rng.normal(0, 2, (R, ncls)).astype(np.float32), props)


@pytest.mark.parametrize("seed,R,cls_id", [(0, 1, 2), (1, 57, 2), (2, 400, 2), (3, 400, 5), (4, 1500, 3)])
def test_final_stage_witness_random(orc, seed, R, cls_id):
    rng = np.random.default_rng(seed)
    bbox, cls, props = _random_outputs(rng, R)
    kw = dict(ratios=(576 / 375.0, 1920 / 1242.0), org_hw=(375, 1242))
    d0, i0 = orc.detections(bbox, cls, props, cls_id, **kw)
    d1, i1 = witness.final_stage(bbox, cls, props, cls_id, **kw)
    assert np.array_equal(i0, i1)                                  # the same boxes survive, in the same order
    assert np.allclose(d0, d1, rtol=2e-6, atol=3e-4)               # single-precision exp may differ in the last place: one ulp of
                                                                   # a ~1000-pixel coordinate is 1.2e-4 (and survives tx - tw/2)


def test_final_stage_witness_adversarial_ties(orc):
    """Equal scores (stable-sort order decides), duplicate boxes, IoU exactly at the 0.5 threshold (strict >), touching boxes
    (iw == 0 is skipped), zero-width / zero-height proposals, scores exactly at proposal_thr, boxes clipped at the border."""
    R = 16
    bbox = np.zeros((R, 20), np.float32)                           # zero deltas: boxes = proposals (exactly representable)
    cls = np.zeros((R, 5), np.float32)                             # all probabilities 0.2: every score ties
    P = np.zeros((R, 6), np.float32)
    P[:, 5] = 1.0
    P[0, 1:5] = [100, 100, 200, 200]
    P[1, 1:5] = [100, 100, 200, 200]                               # duplicate of 0
    P[2, 1:5] = [100, 150, 200, 250]                               # IoU with 0 = 1/3
    P[3, 1:5] = [100, 100, 200, 150]                               # inside 0: IoU exactly 0.5 -> kept (strict >)
    P[4, 1:5] = [200, 100, 300, 200]                               # touches 0: iw == 0
    P[5, 1:5] = [50, 50, 50, 120]                                  # zero width: dropped before anything else
    P[6, 1:5] = [50, 50, 120, 50]                                  # zero height
    P[7, 1:5] = [0, 0, 64, 64]; P[7, 5] = -10.0                    # score == proposal_thr: kept (>=)
    P[8, 1:5] = [0, 0, 64, 64]; P[8, 5] = np.nextafter(np.float32(-10.0), np.float32(-11.0))   # just below: dropped
    P[9, 1:5] = [1200, 300, 1300, 400]                             # beyond the original image after /1: clipped to 1242 x 375
    for k in range(10, R):
        P[k, 1:5] = [400 + 8 * k, 40, 500 + 8 * k, 140]            # a chain of heavily overlapping equal-score boxes
    cls[12, 1] = 0.5                                               # one box of the chain scores higher
    kw = dict(ratios=(1.0, 1.0), org_hw=(375, 1242))
    d0, i0 = orc.detections(bbox, cls, P, 2, **kw)
    d1, i1 = witness.final_stage(bbox, cls, P, 2, **kw)
    assert np.array_equal(i0, i1), (i0, i1)
    assert np.allclose(d0, d1, rtol=2e-6, atol=3e-4)
    kept = set(i0.tolist())
    assert 0 in kept and 1 not in kept and 3 in kept and 4 in kept and 5 not in kept and 6 not in kept
    assert 7 in kept and 8 not in kept and 12 in kept
    row9 = d0[list(i0).index(9)]
    assert row9[0] + row9[2] <= 1242 + 1e-9 and row9[1] + row9[3] <= 375 + 1e-9


def test_nms_witness_matches_oracle_on_many_random_sets(orc):
    """bbNms alone on dense random boxes (thousands of IoU comparisons near every threshold)."""
    rng = np.random.default_rng(11)
    for n in (2, 33, 257, 900):
        x = rng.uniform(0, 200, n); y = rng.uniform(0, 100, n); w = rng.uniform(5, 80, n); h = rng.uniform(5, 60, n)
        score = np.round(rng.uniform(0, 1, n), 2)                  # 2 decimals: many exact ties
        P = np.stack([np.zeros(n), x, y, x + w, y + h, np.ones(n)], 1).astype(np.float32)
        cls = np.zeros((n, 2), np.float32)
        cls[:, 1] = np.log(score + 1e-3)
        d0, i0 = orc.detections(np.zeros((n, 8), np.float32), cls, P, 2, ratios=(1.0, 1.0), org_hw=(1000, 1000))
        d1, i1 = witness.final_stage(np.zeros((n, 8), np.float32), cls, P, 2, ratios=(1.0, 1.0), org_hw=(1000, 1000))
        assert np.array_equal(i0, i1), n


@pytest.mark.parametrize("in_hw,out_hw", [((7, 9), (11, 14)), ((12, 16), (5, 7)), ((9, 5), (9, 13)), ((6, 10), (15, 4)),
                                          ((15, 50), (23, 77))])
def test_imresize_witness_exact_arithmetic(orc, in_hw, out_hw):
    """Up- and down-scaling (antialiased), mixed, identity along one axis; the exact-arithmetic witness and the oracle's float64
    restatement must give the same uint8 image wherever the true value is not within 1e-9 of a .5 rounding boundary."""
    rng = np.random.default_rng(in_hw[0] * 100 + out_hw[1])
    img = rng.integers(0, 256, in_hw + (3,), dtype=np.uint8)
    img[0, 0] = 255; img[-1, -1] = 0; img[0, -1] = [255, 0, 255]                 # saturation at the mirrored borders
    got = orc.imresize_u8(img, out_hw[0], out_hw[1])
    exact, margin = witness.imresize_exact(img, out_hw[0], out_hw[1])
    assert got.shape == exact.shape
    if margin > 1e-9:
        assert np.array_equal(got, exact)
    else:
        assert np.abs(got.astype(int) - exact.astype(int)).max() <= 1


def test_imresize_witness_properties(orc):
    """Properties any correct imresize has: constants are preserved, a horizontal ramp stays a ramp in the interior (bicubic
    reproduces affine functions), left-right mirrored input gives mirrored output."""
    const = np.full((10, 14, 3), 137, np.uint8)
    assert np.all(orc.imresize_u8(const, 23, 31) == 137) and np.all(orc.imresize_u8(const, 4, 6) == 137)
    ramp = np.tile((np.arange(40) * 5)[None, :, None], (8, 1, 3)).astype(np.uint8)
    up = orc.imresize_u8(ramp, 8, 80).astype(int)
    assert np.abs(np.diff(up[4, 6:-6, 0], 2)).max() <= 1           # constant slope up to uint8 rounding
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (9, 12, 3), dtype=np.uint8)
    a = orc.imresize_u8(img, 14, 19)
    b = orc.imresize_u8(img[:, ::-1].copy(), 14, 19)[:, ::-1]
    assert np.array_equal(a, b)
