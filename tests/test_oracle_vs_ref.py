"""Pins the C restatement (oracle/mscnn_oracle.c) to the reference's OWN sources compiled by oracle/ref.mk
(oracle/_ref/libmscnn_ref.so).  CPU only; skipped where _ref has not been built (it needs the reference checkout)."""
import numpy as np
import pytest

from oracle import pyref

pytestmark = pytest.mark.skipif(not pyref.available(), reason="oracle/_ref not built (needs /root/reference at build time)")


def _rois(rng, R, H, W, batch=1):
    x1 = rng.uniform(-60, W, R); y1 = rng.uniform(-60, H, R)
    return np.stack([rng.integers(0, batch, R), x1, y1, x1 + rng.uniform(0, 600, R), y1 + rng.uniform(0, 400, R)], 1).astype(np.float32)


@pytest.mark.parametrize("cfg", [
    (1, 16, 20, 30, 8, (3, 3), (1, 1), (1, 1), 1), (2, 6, 9, 7, 4, (3, 3), (0, 0), (2, 2), 2),
    (1, 32, 18, 60, 9, (5, 5), (2, 2), (1, 1), 1), (1, 8, 12, 20, 7, (7, 5), (3, 2), (1, 1), 1), (3, 16, 7, 7, 12, (3, 3), (0, 0), (1, 1), 1),
])
def test_conv(orc, cfg):
    N, Cin, H, W, Cout, k, pad, stride, g = cfg
    rng = np.random.default_rng(1)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin // g, *k)) * 0.1).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    a, r = orc.conv2d(x, w, b, pad, stride, g), pyref.conv2d(x, w, b, pad, stride, g)
    assert a.shape == r.shape
    assert np.abs(a - r).max() <= 1e-4 * max(1.0, np.abs(r).max())      # BLAS summation order is unspecified


def test_pool_relu_softmax_ip_deconv(orc):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 5, 15, 21)).astype(np.float32)
    for k, p, s, m in [((2, 2), (0, 0), (2, 2), "MAX"), ((3, 3), (1, 1), (2, 2), "MAX"), ((2, 2), (0, 0), (1, 1), "AVE"), ((3, 3), (1, 1), (2, 2), "AVE")]:
        a, r = orc.pool2d(x, k, p, s, m), pyref.pool2d(x, k, p, s, m)
        assert a.shape == r.shape and (np.array_equal(a, r) if m == "MAX" else np.allclose(a, r, atol=1e-6))
    assert np.array_equal(orc.relu(x), pyref.relu(x))
    assert np.allclose(orc.relu(x, 0.1), pyref.relu(x, 0.1), atol=1e-7)
    xi = rng.standard_normal((6, 5)).astype(np.float32)
    assert np.allclose(orc.softmax(xi), pyref.softmax(xi), atol=1e-7)
    w = rng.standard_normal((11, 5 * 15 * 21)).astype(np.float32) * 0.05; b = rng.standard_normal(11).astype(np.float32)
    assert np.allclose(orc.inner_product(x, w, b), pyref.inner_product(x, w, b), atol=1e-4)
    xd = rng.standard_normal((1, 6, 7, 9)).astype(np.float32)
    wd = rng.standard_normal((6, 1, 4, 4)).astype(np.float32)
    assert np.allclose(orc.deconv2d(xd, wd, None, (1, 1), (2, 2), 6), pyref.deconv2d(xd, wd, None, (1, 1), (2, 2), 6), atol=1e-6)
    assert np.array_equal(orc.deconv2d(xd, orc.bilinear_filler((6, 1, 4, 4)), None, (1, 1), (2, 2), 6),
                          pyref.deconv2d(xd, None, None, (1, 1), (2, 2), 6, kernel=(4, 4), num_output=6, bilinear=True))
    wg = rng.standard_normal((6, 2, 3, 3)).astype(np.float32)     # generic (non-depthwise) transposed conv, groups
    assert np.allclose(orc.deconv2d(xd, wg, None, (0, 0), (2, 2), 2), pyref.deconv2d(xd, wg, None, (0, 0), (2, 2), 2), atol=1e-5)


@pytest.mark.parametrize("ph,pw,scale,pad", [(7, 7, 0.125, 0.0), (7, 7, 0.125, 0.25), (7, 5, 0.25, 0.25), (8, 4, 0.125, 0.0), (3, 3, 1.0, 0.5)])
def test_roipool_bitexact(orc, ph, pw, scale, pad):
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((2, 7, 36, 120)).astype(np.float32)
    rois = _rois(rng, 200, 36 / scale, 120 / scale, 2)
    assert np.array_equal(orc.roipool(feat, rois, ph, pw, scale, pad), pyref.roipool(feat, rois, ph, pw, scale, pad))


@pytest.mark.parametrize("ph,pw,scale,pad", [(7, 7, 0.125, 0.0), (7, 7, 0.125, 0.25), (4, 4, 0.25, 0.5)])
def test_roialign_bitexact(orc, ph, pw, scale, pad):
    rng = np.random.default_rng(13)
    feat = rng.standard_normal((2, 5, 36, 120)).astype(np.float32)
    rois = _rois(rng, 150, 36 / scale, 120 / scale, 2)
    rois[:10, 3] = rois[:10, 1] - 5                     # malformed (negative width) -> zeros
    assert np.array_equal(orc.roialign(feat, rois, ph, pw, scale, pad), pyref.roialign(feat, rois, ph, pw, scale, pad))


def test_eltwise(orc):
    rng = np.random.default_rng(14)
    xs = [rng.standard_normal((7, 5)).astype(np.float32) for _ in range(3)]
    for op in ("PROD", "MAX"):
        assert np.array_equal(orc.eltwise(xs, op), pyref.eltwise(xs, op))
    assert np.array_equal(orc.eltwise(xs[:2], "MAX"), pyref.eltwise(xs[:2], "MAX"))
    assert np.allclose(orc.eltwise(xs, "SUM", [0.33333333] * 3), pyref.eltwise(xs, "SUM", [0.33333333] * 3), atol=1e-6)   # saxpy may fuse
    assert np.array_equal(orc.eltwise(xs, "SUM"), pyref.eltwise(xs, "SUM"))


def test_decode_bbox_bitexact(orc):
    rng = np.random.default_rng(4)
    prior = _rois(rng, 300, 576, 1920)
    bbox = (rng.standard_normal((300, 8)) * 2).astype(np.float32)
    for mean, std in [((0, 0, 0, 0), (1, 1, 1, 1)), ((0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2)), ((0.1, -0.1, 0.05, 0), (0.2, 0.2, 0.3, 0.3))]:
        assert np.array_equal(orc.decode_bbox(bbox, prior, mean, std), pyref.decode_bbox(bbox, prior, mean, std))


def test_box_iou_bitexact(orc):
    rng = np.random.default_rng(5)
    for _ in range(2000):
        a = rng.uniform(0, 100, 4).astype(np.float32); b = (a + rng.normal(0, 10, 4)).astype(np.float32)
        for mode in ("IOU", "IOMU", "IOFU"):
            assert orc.box_iou(a, b, mode) == pyref.box_iou(a, b, mode)


SHAPES = [(18, 60), (18, 60), (9, 30), (9, 30), (5, 15), (5, 15), (3, 8)]
FIELD = [60, 84, 120, 168, 240, 336, 480]
DS = [8, 8, 16, 16, 32, 32, 64]


@pytest.mark.parametrize("bg,kw", [
    (-8.0, dict()), (4.0, dict()), (-8.0, dict(max_nms_num=100)), (-8.0, dict(max_post_nms_num=37)), (60.0, dict()),
    (-8.0, dict(nms_type="IOMU", iou_thr=0.5)), (-8.0, dict(nms_type="IOFU", iou_thr=0.7)),
    (-8.0, dict(bbox_mean=[0, 0, 0, 0], bbox_std=[0.1, 0.1, 0.2, 0.2], min_size=5.0)), (-2.0, dict(fg_thr=-7.0, field_whr=3.0, field_xyr=4.0)),
])
def test_boxoutput_bitexact(orc, bg, kw):
    rng = np.random.default_rng(6)
    heads = []
    for (h, w) in SHAPES:
        t = rng.standard_normal((2, 9, h, w)).astype(np.float32)
        t[:, :5] *= 2; t[:, 0] += bg; t[:, 5:] *= 0.5
        t[:, 1:3, :2, :] = 1.25                      # score plateaus: exact ties
        heads.append(t)
    a = orc.boxoutput(heads, FIELD, FIELD, DS, **kw)
    r = pyref.boxoutput(heads, FIELD, FIELD, DS, **kw)
    assert a[0].shape == r[0].shape
    assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1])


def test_small_net_end_to_end(orc):
    """The oracle net executor on both back-ends: same graph, restatement vs the reference's own layer classes."""
    from mscnn_amd import net as mnet, synth, zoo
    from oracle import pynet
    n = mnet.Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576", height=128, width=256, max_nms_num=60), device=-1)
    layers = [(n.layer_names[i], n.layer_types[i], n.layer_bottoms(i), n.layer_tops(i), n.layer_param_text(i)) for i in range(len(n.layer_names))]
    ws = synth.weights(n.layer_names, n.layer_types, [n.param_shapes(i) for i in range(len(n.layer_names))], "mid")
    x = synth.frame(128, 256)
    a = pynet.forward(layers, ws, {"data": x})
    r = pynet.forward(layers, ws, {"data": x}, backend=pyref)
    for b in ("conv4_3", "conv6_1", "LFCN_1_7x7", "LFCN_4_5x5"):
        assert np.abs(a[b] - r[b]).max() <= 1e-4 * max(1.0, np.abs(r[b]).max()), b
    # same graph, same heads -> same proposals (feed the reference's heads to the restatement to avoid BLAS-order flips)
    bo = [l for l in layers if l[1] == "BoxOutput"][0]
    a2 = pynet.forward([bo], ws, {k: r[k] for k in bo[2]})
    assert np.array_equal(a2["proposals"], r["proposals"]) and np.array_equal(a2["proposals_score"], r["proposals_score"])
