"""Pins the CPU oracle (oracle/mscnn_oracle.c) against the reference's own known-answer
tests and against hand-computable cases for the MS-CNN layers (which the reference
does not test at all).  CPU only."""
import numpy as np
import pytest


# ---- pooling KATs: src/caffe/test/test_pooling_layer.cpp:49-119 and :478-522 ----
def test_pool_square_kat(orc):
    plane = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)
    x = np.tile(plane, (2, 2, 1, 1))
    y, mask = orc.pool2d(x, kernel=(2, 2), stride=(1, 1), with_mask=True)
    assert y.shape == (2, 2, 2, 4)
    exp = np.array([[9, 5, 5, 8], [9, 5, 5, 8]], np.float32)
    expm = np.array([[5, 2, 2, 9], [5, 12, 12, 9]], np.int32)
    for n in range(2):
        for c in range(2):
            assert np.array_equal(y[n, c], exp)
            assert np.array_equal(mask[n, c], expm)


def test_pool_max_padded_kat(orc):
    x = np.array([[1, 2, 4], [2, 3, 2], [4, 2, 1]], np.float32).reshape(1, 1, 3, 3)
    y = orc.pool2d(x, kernel=(3, 3), pad=(2, 2), stride=(2, 2))
    assert y.shape == (1, 1, 3, 3)
    assert np.array_equal(y[0, 0], np.array([[1, 4, 4], [4, 4, 4], [4, 4, 1]], np.float32))


def test_pool_ceil_mode_shapes(orc):
    # pooling_layer.cpp:90-93 ceil mode: 15 -> 8 (caltech pool6 480x640: 15x20 -> 8x10)
    assert orc.pool_out_dim(15, 2, 0, 2) == 8
    assert orc.pool_out_dim(576, 2, 0, 2) == 288
    assert orc.pool_out_dim(9, 2, 0, 2) == 5


def test_pool_ave_kat(orc):
    # test_pooling_layer.cpp TestForwardAve: 3x3 constant 2, k3 s1 p1 -> [8/9 4/3 8/9; 4/3 2 4/3; ...]
    x = np.full((1, 1, 3, 3), 2, np.float32)
    y = orc.pool2d(x, kernel=(3, 3), pad=(1, 1), stride=(1, 1), method="AVE")
    exp = np.array([[8 / 9, 4 / 3, 8 / 9], [4 / 3, 2, 4 / 3], [8 / 9, 4 / 3, 8 / 9]], np.float32)
    assert np.allclose(y[0, 0], exp, atol=1e-5)


# ---- convolution: naive comparator (test_convolution_layer.cpp:19-139) vs im2col+GEMM ----
@pytest.mark.parametrize("cfg", [
    dict(N=2, Cin=3, H=6, W=4, Cout=4, k=(3, 3), pad=(0, 0), stride=(2, 2), group=1),  # TestSimpleConvolution
    dict(N=2, Cin=3, H=6, W=4, Cout=4, k=(1, 1), pad=(0, 0), stride=(1, 1), group=1),  # Test1x1Convolution
    dict(N=2, Cin=6, H=6, W=4, Cout=3, k=(3, 3), pad=(0, 0), stride=(2, 2), group=3),  # TestSimpleConvolutionGroup
    dict(N=1, Cin=5, H=9, W=11, Cout=7, k=(5, 7), pad=(2, 3), stride=(1, 1), group=1),
    dict(N=1, Cin=4, H=7, W=7, Cout=6, k=(3, 3), pad=(1, 1), stride=(1, 1), group=1),
])
def test_conv_gemm_equals_naive_bitexact(orc, cfg):
    rng = np.random.default_rng(1701)
    x = rng.standard_normal((cfg["N"], cfg["Cin"], cfg["H"], cfg["W"])).astype(np.float32)
    w = rng.standard_normal((cfg["Cout"], cfg["Cin"] // cfg["group"], *cfg["k"])).astype(np.float32)
    b = rng.standard_normal(cfg["Cout"]).astype(np.float32)
    a = orc.conv2d(x, w, b, cfg["pad"], cfg["stride"], cfg["group"], naive=True)
    g = orc.conv2d(x, w, b, cfg["pad"], cfg["stride"], cfg["group"], naive=False)
    assert a.shape == g.shape
    assert np.array_equal(a, g)
    # independent float64 definition, tolerance 1e-4 as test_convolution_layer.cpp:256,263
    N, Cin, H, W = x.shape
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (cfg["pad"][0],) * 2, (cfg["pad"][1],) * 2))
    cig, cog = Cin // cfg["group"], cfg["Cout"] // cfg["group"]
    ref = np.zeros(a.shape)
    for o in range(cfg["Cout"]):
        gi = o // cog
        for oy in range(a.shape[2]):
            for ox in range(a.shape[3]):
                patch = xp[:, gi * cig:(gi + 1) * cig, oy * cfg["stride"][0]:oy * cfg["stride"][0] + cfg["k"][0],
                           ox * cfg["stride"][1]:ox * cfg["stride"][1] + cfg["k"][1]]
                ref[:, o, oy, ox] = (patch * w[o].astype(np.float64)).sum(axis=(1, 2, 3)) + b[o]
    assert np.allclose(a, ref, atol=1e-4)


def test_conv_sobel_separable(orc):
    # test_convolution_layer.cpp:498-589: 3x3 Sobel == (3x1 [1 2 1]^T) o (1x3 [-1 0 1])
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 3, 6, 4)).astype(np.float32)
    sob = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float32)
    w = np.tile(sob, (1, 3, 1, 1))
    full = orc.conv2d(x, w, None, (0, 0), (2, 2))
    col = np.tile(np.array([[1], [2], [1]], np.float32), (1, 3, 1, 1))
    t = orc.conv2d(x, col, None, (0, 0), (2, 1))
    row = np.array([[-1, 0, 1]], np.float32).reshape(1, 1, 1, 3)
    sep = orc.conv2d(t, row, None, (0, 0), (1, 2))
    assert np.allclose(full, sep, atol=1e-4)


# ---- deconvolution: test_deconvolution_layer.cpp:91-137 overlap counts ----
def test_deconv_overlap_counts(orc):
    x = np.ones((2, 3, 6, 4), np.float32)
    w = np.ones((3, 4, 3, 3), np.float32)
    b = np.full(4, 0.1, np.float32)
    y = orc.deconv2d(x, w, b, (0, 0), (2, 2))
    assert y.shape == (2, 4, 13, 9)
    for h in range(13):
        for ww in range(9):
            expected = 3.1
            h_overlap = h % 2 == 0 and 0 < h < 12
            w_overlap = ww % 2 == 0 and 0 < ww < 8
            if h_overlap and w_overlap:
                expected += 9
            elif h_overlap or w_overlap:
                expected += 3
            assert np.allclose(y[:, :, h, ww], expected, atol=1e-4)


def test_deconv_bilinear_depthwise_upsamples(orc):
    # kitti_car/mscnn-7s-576-2x deploy :452-467: group=C 4x4 s2 p1 bilinear -> 2x upsample
    C, H, W = 4, 5, 6
    x = np.random.default_rng(3).standard_normal((1, C, H, W)).astype(np.float32)
    w = orc.bilinear_filler((C, 1, 4, 4))
    assert np.allclose(w[0, 0, 0], [0.0625, 0.1875, 0.1875, 0.0625])
    y = orc.deconv2d(x, w, None, (1, 1), (2, 2), group=C)
    assert y.shape == (1, C, 2 * H, 2 * W)
    # interior output (2i+1, 2j+1) = 9/16 x[i,j] + 3/16 x[i+1,j] + 3/16 x[i,j+1] + 1/16 x[i+1,j+1]
    i, j = 1, 2
    exp = (9 * x[0, :, i, j] + 3 * x[0, :, i + 1, j] + 3 * x[0, :, i, j + 1] + x[0, :, i + 1, j + 1]) / 16
    assert np.allclose(y[0, :, 2 * i + 1, 2 * j + 1], exp, atol=1e-5)


# ---- ReLU / InnerProduct / Softmax ----
def test_relu(orc):
    x = np.array([-1.5, 0.0, 2.0, -0.0], np.float32)
    assert np.array_equal(orc.relu(x), np.array([0, 0, 2, 0], np.float32))
    assert np.allclose(orc.relu(x, 0.01), [-0.015, 0, 2, 0])


def test_inner_product(orc):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 2, 3, 4)).astype(np.float32)
    w = rng.standard_normal((10, 24)).astype(np.float32)
    b = rng.standard_normal(10).astype(np.float32)
    y = orc.inner_product(x, w, b)
    ref = x.reshape(3, -1).astype(np.float64) @ w.T.astype(np.float64) + b
    assert np.allclose(y, ref, atol=1e-4)


def test_softmax(orc):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 10, 2, 3)).astype(np.float32)
    y = orc.softmax(x, axis=1)
    assert np.allclose(y.sum(axis=1), 1, atol=1e-5)
    e = np.exp(x - x.max(axis=1, keepdims=True))
    assert np.allclose(y, e / e.sum(axis=1, keepdims=True), atol=1e-6)


# ---- BoxIOU (math_functions.cpp:12-35) ----
def test_box_iou_modes(orc):
    a = (0, 0, 10, 10); b = (5, 5, 10, 10)
    assert orc.box_iou(a, b) == pytest.approx(25 / 175)
    assert orc.box_iou(a, b, "IOMU") == pytest.approx(0.25)
    assert orc.box_iou((0, 0, 10, 10), (5, 5, 20, 20), "IOFU") == pytest.approx(0.25)
    assert orc.box_iou(a, (10, 0, 5, 5)) == 0.0            # touching edges: tlx >= brx
    assert orc.box_iou(a, (0, 0, 0, 5)) == 0.0             # degenerate width


def test_nms_threshold_is_strict(orc):
    # IoU exactly equal to the threshold is kept (box_output_layer.cpp:54 `o>overlap`)
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 5], [100, 100, 5, 5]], np.float32)  # IoU(0,1) = 0.5
    assert orc.nms_greedy(boxes, 0.5).tolist() == [True, True, True]
    assert orc.nms_greedy(boxes, 0.4999).tolist() == [True, False, True]
    # greedy: a suppressed box does not suppress others
    chain = np.array([[0, 0, 10, 10], [4, 0, 10, 10], [8, 0, 10, 10]], np.float32)  # 0-1: .4286, 1-2: .4286, 0-2: .111
    assert orc.nms_greedy(chain, 0.4).tolist() == [True, False, True]


# ---- BoxOutput hand cases (box_output_layer.cpp:66-234) ----
def _head(h, w, cls=5):
    return np.zeros((1, cls + 4, h, w), np.float32)


def test_boxoutput_empty_dummy(orc):
    hd = _head(2, 3); hd[0, 0] = 100.0                      # background dominates -> fg < fg_thr
    rois, props, cidx, nreal = orc.boxoutput([hd], [60], [60], [8])
    assert nreal == 0
    assert rois.tolist() == [[0, 1, 1, 10, 10]]
    assert props.tolist() == [[0, 0, 0, 0, 0, 0]]


def test_boxoutput_single_anchor_decode(orc):
    hd = _head(4, 6); hd[0, 0] = 100.0
    hd[0, 0, 2, 3] = 0.0; hd[0, 2, 2, 3] = 3.0              # fg = 3 at (h=2,w=3)
    hd[0, 5, 2, 3] = 0.25; hd[0, 6, 2, 3] = -0.9           # dx, dy (dy clamps to -0.5)
    hd[0, 7, 2, 3] = np.log(1.5); hd[0, 8, 2, 3] = 5.0     # dw, dh (dh clamps to ln 2)
    fw = fh = 16.0; s = 8.0
    rois, props, cidx, nreal = orc.boxoutput([hd], [fw], [fh], [s], min_size=1.0)
    assert nreal == 1 and cidx.tolist() == [0]
    cx = 0.25 * fw + 3.5 * s; cy = -0.5 * fh + 2.5 * s
    bw = fw * 1.5; bh = fh * 2.0
    x = max(cx - bw / 2, 0); y = max(cy - bh / 2, 0)
    bw = min(bw, 6 * s - x); bh = min(bh, 4 * s - y)
    assert np.allclose(rois[0], [0, x, y, x + bw, y + bh], rtol=1e-6)
    assert props[0, 5] == pytest.approx(3.0)


def test_boxoutput_min_size_boundary_and_tie_order(orc):
    # two identical-score anchors far apart: ties sort by LARGER candidate index first (:168)
    hd = _head(4, 40); hd[0, 0] = 100.0
    for w in (2, 30):
        hd[0, 0, 1, w] = 0.0; hd[0, 1, 1, w] = 1.0
    rois, props, cidx, nreal = orc.boxoutput([hd], [15.0], [15.0], [8.0], min_size=15.0)
    assert nreal == 2
    assert cidx.tolist() == [1, 0]                          # larger insertion index first
    assert rois[0, 1] > rois[1, 1]
    # exactly min_size passes (>=); just below fails
    rois2, _, _, nreal2 = orc.boxoutput([hd], [14.999], [15.0], [8.0], min_size=15.0)
    assert nreal2 == 0


def test_boxoutput_topk_and_nms(orc):
    rng = np.random.default_rng(1701)
    hd = rng.standard_normal((1, 9, 12, 20)).astype(np.float32)
    rois, props, cidx, nreal = orc.boxoutput([hd], [40.0], [40.0], [8.0], max_nms_num=50, min_size=5.0)
    assert 1 <= nreal <= 50
    assert np.all(np.diff(props[:, 5]) <= 0)                # emitted in score order
    # re-run NMS on the output: nothing more is suppressed (idempotence)
    xywh = np.stack([rois[:, 1], rois[:, 2], rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]], 1)
    assert orc.nms_greedy(xywh, 0.65).all()


# ---- ROIPooling hand cases (roi_pooling_layer.cpp:48-139) ----
def test_roipool_whole_map_and_first_max(orc):
    feat = np.arange(2 * 4 * 6, dtype=np.float32).reshape(1, 2, 4, 6)
    rois = np.array([[0, 0, 0, 5, 3]], np.float32)          # scale 1: covers whole map
    out, am = orc.roipool(feat, rois, 2, 2, 1.0, 0.0, with_argmax=True)
    assert out[0, 0].tolist() == [[8, 11], [20, 23]]
    assert am[0, 0].tolist() == [[8, 11], [20, 23]]
    flat = np.zeros((1, 1, 4, 6), np.float32)               # all equal: first (top-left) index wins
    out, am = orc.roipool(flat, rois, 1, 1, 1.0, 0.0, with_argmax=True)
    assert out[0, 0, 0, 0] == 0 and am[0, 0, 0, 0] == 0


def test_roipool_unclipped_and_empty_bins(orc):
    feat = np.full((1, 1, 4, 4), -3.0, np.float32)
    rois = np.array([[0, -40, -40, -20, -20],               # fully outside: every bin empty -> 0
                     [0, -8, -8, 40, 40]], np.float32)      # oversize, partially outside
    out, am = orc.roipool(feat, rois, 2, 2, 0.125, 0.0, with_argmax=True)
    assert np.all(out[0] == 0) and np.all(am[0] == -1)
    assert np.all(out[1] == -3.0)                           # negative features are NOT clamped to 0


def test_roipool_pad_ratio_and_rounding(orc):
    feat = np.arange(8 * 8, dtype=np.float32).reshape(1, 1, 8, 8)
    # w = 16+1 = 17 -> pad 4.25; start = round((16-4.25)*.125) = round(1.46875) = 1
    # end = round((32+4.25)*.125) = round(4.53) = 5
    rois = np.array([[0, 16, 16, 32, 32]], np.float32)
    out = orc.roipool(feat, rois, 1, 1, 0.125, 0.25)
    assert out[0, 0, 0, 0] == 5 * 8 + 5
    out0 = orc.roipool(feat, rois, 1, 1, 0.125, 0.0)        # start 2, end 4
    assert out0[0, 0, 0, 0] == 4 * 8 + 4
    # half away from zero: 20*.125 = 2.5 -> 3 ; -20*.125 = -2.5 -> -3
    r2 = np.array([[0, 20, 20, 20, 20]], np.float32)
    assert orc.roipool(feat, r2, 1, 1, 0.125, 0.0)[0, 0, 0, 0] == 3 * 8 + 3


# ---- DecodeBBox (decode_bbox_layer.cpp:54-123, math_functions.cpp:46-75) ----
def test_decode_bbox_identity_and_shift(orc):
    prior = np.array([[0, 10, 20, 49, 79]], np.float32)      # w=40, h=60, c=(29.5, 49.5)
    bbox = np.zeros((1, 8), np.float32)
    out = orc.decode_bbox(bbox, prior)
    assert np.allclose(out[0], [0, 10, 20, 49, 79])
    bbox[0, 4:] = [1.0, -1.0, np.log(2.0), 0.0]
    out = orc.decode_bbox(bbox, prior, std=(0.1, 0.1, 0.2, 0.2))
    tw = 40 * np.exp(0.2 * np.log(2.0)); tx = 0.1 * 40 + 29.5 - (tw - 1) / 2
    assert np.allclose(out[0], [0, tx, -0.1 * 60 + 49.5 - 59 / 2, tx + tw - 1, -0.1 * 60 + 49.5 - 59 / 2 + 59], rtol=1e-5)


# ---- final detection stage (run_mscnn_detection.m:75-120, bbNms.m:112-126) ----
def test_detections_stage(orc):
    props = np.array([[0, 100, 100, 200, 180, 2.0],
                      [0, 104, 100, 204, 180, 1.0],       # overlaps #0 heavily
                      [0, 500, 50, 560, 120, -11.0],      # below proposal_thr
                      [0, 700, 60, 700, 100, 3.0],        # zero width -> dropped
                      [0, 900, 200, 980, 260, 0.5]], np.float32)
    R, ncls = props.shape[0], 5
    bbox_pred = np.zeros((R, 4 * ncls), np.float32)
    cls_pred = np.zeros((R, ncls), np.float32)
    cls_pred[:, 1] = [4.0, 3.0, 9.0, 9.0, 1.0]
    dets, ids = orc.detections(bbox_pred, cls_pred, props, cls_id=2, ratios=(576 / 375, 1920 / 1242), org_hw=(375, 1242))
    assert ids.tolist() == [0, 4]
    p0 = np.exp(4.0) / (np.exp(4.0) + 4)
    assert dets[0, 4] == pytest.approx(p0, rel=1e-6)
    assert dets[0, 0] == pytest.approx(100 / (1920 / 1242), rel=1e-6)
    assert dets[0, 2] == pytest.approx(100 / (1920 / 1242), rel=1e-6)
    # stable sort: equal prob keeps the lower input row first
    cls_pred[:, 1] = 1.0
    far = props.copy(); far[1, 1:5] = [1200, 300, 1300, 380]
    dets, ids = orc.detections(bbox_pred, cls_pred, far, cls_id=2, ratios=(1.0, 1.0), org_hw=(576, 1920))
    assert ids.tolist() == [0, 1, 4]


# ---------------------------------------------------------------------------------------------- pre-processing (8f rank 3)
def test_imresize_contributions_hand_computed(orc):
    """MATLAB imresize `contributions` for a 2 -> 4 bicubic upsample: u = x/2 + 0.25, taps floor(u - 2) + 0..5, cubic kernel
    a = -0.5 evaluated by hand (all values are dyadic, exact in binary), symmetric border mirroring aux = [1 2 2 1]."""
    w, idx = orc.imresize_contributions(2, 4)
    c175, c075, c025, c125 = -0.0234375, 0.2265625, 0.8671875, -0.0703125
    assert np.array_equal(w[0], [0.0, c175, c075, c025, c125, 0.0])          # u = 0.75
    assert np.array_equal(w[1], [0.0, c125, c025, c075, c175, 0.0])          # u = 1.25
    assert np.array_equal(idx[0], [1, 1, 0, 0, 1, 1])                        # 1-based -2..3 mirrored into {1, 2}
    assert np.array_equal(idx[3], [0, 0, 1, 1, 0, 0])


def test_preprocess_properties(orc):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 124, 3), dtype=np.uint8)
    assert np.array_equal(orc.imresize_u8(img, 37, 124), img)                # scale 1: every weight row is [.. 0 1 0 ..]
    flat = np.full((20, 30, 3), 77, np.uint8)
    assert np.all(orc.imresize_u8(flat, 31, 47) == 77) and np.all(orc.imresize_u8(flat, 9, 11) == 77)   # weights sum to 1
    up = orc.imresize_u8(img, 57, 191)
    assert up.shape == (57, 191, 3) and up.dtype == np.uint8
    down = orc.imresize_u8(img, 12, 40)                                      # antialiased (kernel stretched by 1/scale)
    assert abs(float(down.mean()) - float(img.mean())) < 2.0
    x = orc.preprocess(img, 57, 191, mean_bgr=(104.0, 117.0, 123.0))
    assert x.shape == (1, 3, 57, 191) and x.dtype == np.float32
    assert np.array_equal(x[0, 0] + 104.0, up[:, :, 2].astype(np.float32))   # plane 0 = blue
    assert np.array_equal(x[0, 2] + 123.0, up[:, :, 0].astype(np.float32))   # plane 2 = red


def test_detections_cascade_stage(orc):
    """run_cascademscnn.m:84-117 by hand: rescale by the ratios, clip to the original image, w = x2 - x1 + 1, rows whose
    PROPOSAL is degenerate dropped, det_thr, NMS (union > 0.5)."""
    boxes = np.array([[0, 100, 100, 199, 179], [0, 104, 100, 203, 179], [0, -20, -5, 50, 40], [0, 1900, 560, 2000, 600],
                      [0, 300, 300, 340, 330]], np.float32)
    props = boxes.copy()
    props[4, 3] = props[4, 1] - 1                        # proposal width x2 - x1 + 1 == 0 -> row dropped
    prob = np.zeros((5, 3), np.float32)
    prob[:, 1] = [0.9, 0.8, 0.7, 0.6, 0.99]
    dets, ids = orc.detections_cascade(boxes, prob, props, cls_id=2, ratios=(2.0, 2.0), org_hw=(288, 960))
    assert ids.tolist() == [0, 2, 3]                     # row 1 suppressed by row 0, row 4 dropped
    assert dets[0].tolist() == [50.0, 50.0, 50.5, 40.5, np.float32(0.9)]
    assert dets[1].tolist() == [0.0, 0.0, 26.0, 21.0, np.float32(0.7)]          # clipped at 0: w = 25 - 0 + 1
    assert dets[2].tolist() == [950.0, 280.0, 11.0, 9.0, np.float32(0.6)]       # clipped at org_w / org_h
    d2, i2 = orc.detections_cascade(boxes, prob, props, cls_id=2, det_thr=0.65, ratios=(2.0, 2.0), org_hw=(288, 960))
    assert i2.tolist() == [0, 2]
