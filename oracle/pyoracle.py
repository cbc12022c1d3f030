"""ctypes/numpy front-end of the CPU oracle (oracle/mscnn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by anything under mscnn_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmscnn_oracle.so")
_lib = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)


def build(force=False):
    """Compile oracle/mscnn_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "mscnn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libmscnn_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_box_iou.restype = C.c_float
        _lib.orc_box_iou.argtypes = [C.c_float] * 8 + [C.c_int]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


class BoxOutputParams(C.Structure):
    _fields_ = [("fg_thr", C.c_float), ("iou_thr", C.c_float), ("nms_mode", C.c_int),
                ("field_whr", C.c_float), ("field_xyr", C.c_float),
                ("max_nms_num", C.c_int), ("max_post_nms_num", C.c_int),
                ("min_size", C.c_float), ("do_bbox_norm", C.c_int),
                ("bbox_mean", C.c_float * 4), ("bbox_std", C.c_float * 4)]


NMS_MODES = {"IOU": 0, "IOMU": 1, "IOFU": 2}


def conv_out_dim(i, k, p, s):
    return (i + 2 * p - k) // s + 1


def conv2d(x, w, b=None, pad=(0, 0), stride=(1, 1), group=1, naive=False):
    x, xp = _f(x); w, wp = _f(w)
    N, Cin, H, W = x.shape
    Cout, _, Kh, Kw = w.shape
    Ho, Wo = conv_out_dim(H, Kh, pad[0], stride[0]), conv_out_dim(W, Kw, pad[1], stride[1])
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    args = [xp, wp, bp, y.ctypes.data_as(f32p), N, Cin, H, W, Cout, Kh, Kw,
            pad[0], pad[1], stride[0], stride[1], group]
    rc = lib().orc_conv2d_naive(*args) if naive else lib().orc_conv2d(*args, None)
    assert rc == 0, rc
    return y


def relu(x, slope=0.0):
    x, xp = _f(x)
    y = np.empty_like(x)
    lib().orc_relu(xp, y.ctypes.data_as(f32p), C.c_long(x.size), C.c_float(slope))
    return y


def pool_out_dim(i, k, p, s):
    return lib().orc_pool_out_dim(i, k, p, s)


def pool2d(x, kernel=(2, 2), pad=(0, 0), stride=(2, 2), method="MAX", with_mask=False):
    x, xp = _f(x)
    N, Cc, H, W = x.shape
    Ho, Wo = pool_out_dim(H, kernel[0], pad[0], stride[0]), pool_out_dim(W, kernel[1], pad[1], stride[1])
    y = np.empty((N, Cc, Ho, Wo), np.float32)
    mask = np.empty((N, Cc, Ho, Wo), np.int32) if with_mask else None
    rc = lib().orc_pool2d(xp, y.ctypes.data_as(f32p), mask.ctypes.data_as(i32p) if with_mask else None,
                          N, Cc, H, W, kernel[0], kernel[1], pad[0], pad[1], stride[0], stride[1],
                          0 if method == "MAX" else 1)
    assert rc == 0
    return (y, mask) if with_mask else y


def inner_product(x, w, b=None):
    x, xp = _f(x); w, wp = _f(w)
    M = x.shape[0]; K = int(np.prod(x.shape[1:])); Nn = w.shape[0]
    assert w.size == Nn * K
    y = np.empty((M, Nn), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    lib().orc_inner_product(xp, wp, bp, y.ctypes.data_as(f32p), M, Nn, K)
    return y


def deconv_out_dim(i, k, p, s):
    return s * (i - 1) + k - 2 * p


def deconv2d(x, w, b=None, pad=(0, 0), stride=(1, 1), group=1):
    x, xp = _f(x); w, wp = _f(w)
    N, Cin, H, W = x.shape
    _, cog, Kh, Kw = w.shape
    Cout = cog * group
    Ho, Wo = deconv_out_dim(H, Kh, pad[0], stride[0]), deconv_out_dim(W, Kw, pad[1], stride[1])
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    rc = lib().orc_deconv2d(xp, wp, bp, y.ctypes.data_as(f32p), N, Cin, H, W, Cout, Kh, Kw,
                            pad[0], pad[1], stride[0], stride[1], group)
    assert rc == 0
    return y


def bilinear_filler(shape):
    w = np.empty(shape, np.float32)
    rc = lib().orc_bilinear_filler(w.ctypes.data_as(f32p), w.size, shape[2], shape[3])
    assert rc == 0
    return w


def softmax(x, axis=1):
    x, xp = _f(x)
    outer = int(np.prod(x.shape[:axis])); Cc = x.shape[axis]; inner = int(np.prod(x.shape[axis + 1:]))
    y = np.empty_like(x)
    lib().orc_softmax(xp, y.ctypes.data_as(f32p), outer, Cc, inner)
    return y


def box_iou(a, b, mode="IOU"):
    return float(lib().orc_box_iou(*[C.c_float(float(v)) for v in list(a) + list(b)], NMS_MODES[mode]))


def nms_greedy(boxes_xywh, thr, mode="IOU"):
    b, bp = _f(boxes_xywh)
    n = b.shape[0]
    keep = np.zeros(max(n, 1), np.uint8)
    lib().orc_nms_greedy(bp, n, C.c_float(thr), NMS_MODES[mode], keep.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return keep[:n].astype(bool)


def boxoutput(heads, field_w, field_h, downsample, fg_thr=-5.0, iou_thr=0.65, nms_type="IOU",
              field_whr=2.0, field_xyr=2.0, max_nms_num=2000, max_post_nms_num=0, min_size=15.0,
              bbox_mean=None, bbox_std=None, with_anchor_ids=False):
    """heads: list of (num, cls+4, h, w) arrays. Returns (rois[R,5], props[R,6], cand_idx[R], num_real)."""
    hs = [np.ascontiguousarray(h, np.float32) for h in heads]
    n = len(hs)
    num, channels = hs[0].shape[:2]
    ptrs = (f32p * n)(*[h.ctypes.data_as(f32p) for h in hs])
    hh = (C.c_int * n)(*[h.shape[2] for h in hs]); ww = (C.c_int * n)(*[h.shape[3] for h in hs])
    fw = (C.c_float * n)(*field_w); fh = (C.c_float * n)(*field_h); ds = (C.c_float * n)(*downsample)
    p = BoxOutputParams(fg_thr, iou_thr, NMS_MODES[nms_type], field_whr, field_xyr, max_nms_num,
                        max_post_nms_num, min_size, 0, (C.c_float * 4)(0, 0, 0, 0), (C.c_float * 4)(1, 1, 1, 1))
    if bbox_mean is not None and bbox_std is not None and len(bbox_mean) and len(bbox_std):
        p.do_bbox_norm = 1
        p.bbox_mean = (C.c_float * 4)(*bbox_mean); p.bbox_std = (C.c_float * 4)(*bbox_std)
    cap = max(1, sum(h.shape[2] * h.shape[3] for h in hs) * num)
    if max_nms_num > 0:
        cap = max(1, min(cap, max_nms_num * num))
    rois = np.zeros((cap, 5), np.float32); props = np.zeros((cap, 6), np.float32)
    cidx = np.zeros(cap, np.int32); aids = np.zeros(cap, np.int32); nreal = C.c_int(0)
    R = lib().orc_boxoutput(ptrs, hh, ww, n, num, channels, fw, fh, ds, C.byref(p),
                            rois.ctypes.data_as(f32p), props.ctypes.data_as(f32p),
                            cidx.ctypes.data_as(i32p), aids.ctypes.data_as(i32p), cap, C.byref(nreal))
    assert R >= 1, R
    if with_anchor_ids:
        return rois[:R].copy(), props[:R].copy(), cidx[:R].copy(), nreal.value, aids[:R].copy()
    return rois[:R].copy(), props[:R].copy(), cidx[:R].copy(), nreal.value


def roipool(feat, rois, pooled_h, pooled_w, spatial_scale, pad_ratio=0.0, with_argmax=False):
    feat, fp = _f(feat); rois, rp = _f(rois)
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, Cc, pooled_h, pooled_w), np.float32)
    am = np.empty((R, Cc, pooled_h, pooled_w), np.int32) if with_argmax else None
    rc = lib().orc_roipool(fp, rp, out.ctypes.data_as(f32p), am.ctypes.data_as(i32p) if with_argmax else None,
                           R, N, Cc, H, W, pooled_h, pooled_w, C.c_float(spatial_scale), C.c_float(pad_ratio))
    assert rc == 0, rc
    return (out, am) if with_argmax else out


def roialign(feat, rois, pooled_h, pooled_w, spatial_scale, pad_ratio=0.0):
    feat, fp = _f(feat); rois, rp = _f(rois)
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, Cc, pooled_h + 1, pooled_w + 1), np.float32)
    rc = lib().orc_roialign(fp, rp, out.ctypes.data_as(f32p), R, N, Cc, H, W, pooled_h, pooled_w, C.c_float(spatial_scale), C.c_float(pad_ratio))
    assert rc == 0, rc
    return out


ELTWISE_OPS = {"PROD": 0, "SUM": 1, "MAX": 2}


def eltwise(xs, op="SUM", coeffs=None):
    arrs = [np.ascontiguousarray(x, np.float32) for x in xs]
    n = len(arrs)
    ptrs = (f32p * n)(*[a.ctypes.data_as(f32p) for a in arrs])
    cf = (C.c_float * n)(*(coeffs if coeffs is not None and len(coeffs) else [1.0] * n))
    y = np.empty_like(arrs[0])
    rc = lib().orc_eltwise(ptrs, n, cf, y.ctypes.data_as(f32p), C.c_long(y.size), ELTWISE_OPS[op])
    assert rc == 0
    return y


def decode_bbox(bbox, prior, mean=(0, 0, 0, 0), std=(1, 1, 1, 1)):
    bbox, bp = _f(bbox); prior, pp = _f(prior)
    R = bbox.shape[0]
    out = np.empty((R, 5), np.float32)
    m = (C.c_float * 4)(*mean); s = (C.c_float * 4)(*std)
    rc = lib().orc_decode_bbox(bp, pp, out.ctypes.data_as(f32p), R, bbox.shape[1], m, s)
    assert rc == 0
    return out


def detections(bbox_pred, cls_pred, props, cls_id, bbox_mean=(0, 0, 0, 0), bbox_std=(0.1, 0.1, 0.2, 0.2),
               proposal_thr=-10.0, ratios=(1.0, 1.0), org_hw=(375, 1242), nms_overlap=0.5):
    """Final MATLAB stage. Returns (dets[D,5] float64 [x y w h prob], ids[D] rows of the inputs)."""
    bbox_pred, bp = _f(bbox_pred); cls_pred, cp = _f(cls_pred); props, pp = _f(props)
    R = props.shape[0]; ncls = cls_pred.shape[1]
    dets = np.zeros((max(R, 1), 5), np.float64); ids = np.zeros(max(R, 1), np.int32)
    m = (C.c_float * 4)(*bbox_mean); s = (C.c_float * 4)(*bbox_std)
    D = lib().orc_detections(bp, cp, pp, R, ncls, cls_id, m, s, C.c_float(proposal_thr),
                             C.c_double(ratios[0]), C.c_double(ratios[1]),
                             C.c_double(org_hw[0]), C.c_double(org_hw[1]), C.c_double(nms_overlap),
                             dets.ctypes.data_as(C.POINTER(C.c_double)), ids.ctypes.data_as(i32p))
    return dets[:D].copy(), ids[:D].copy()


def detections_cascade(boxes, cls_prob, props, cls_id, det_thr=0.0, ratios=(1.0, 1.0), org_hw=(375, 1242), nms_overlap=0.5):
    """Final stage of the cascade drivers (run_cascademscnn.m:84-117). Returns (dets[D,5] float64 [x y w h prob], ids[D])."""
    boxes, bp = _f(boxes); cls_prob, cp = _f(cls_prob); props, pp = _f(props)
    R = props.shape[0]; ncls = cls_prob.shape[1]
    dets = np.zeros((max(R, 1), 5), np.float64); ids = np.zeros(max(R, 1), np.int32)
    D = lib().orc_detections_cascade(bp, cp, pp, R, ncls, cls_id, C.c_float(det_thr), C.c_double(ratios[0]), C.c_double(ratios[1]),
                                     C.c_double(org_hw[0]), C.c_double(org_hw[1]), C.c_double(nms_overlap),
                                     dets.ctypes.data_as(C.POINTER(C.c_double)), ids.ctypes.data_as(i32p))
    return dets[:D].copy(), ids[:D].copy()


def concat_channels(xs):
    return np.concatenate([np.asarray(x, np.float32) for x in xs], axis=1)


# ---------------------------------------------------------------------------------------------- pre-processing
# The step in front of net.forward in the reference's MATLAB driver (examples/kitti_car/run_mscnn_detection.m:64-69):
#     test_image = imresize(test_image,[imgH imgW]);  single(test_image(:,:,[3 2 1]));  minus mu;  permute [2 1 3]
# PARITY UNPINNED: MATLAB is not available here; `imresize` is restated from its published algorithm (imresize.m,
# `contributions`): bicubic kernel (a = -0.5), kernel stretched by 1/scale when shrinking (antialiasing), output pixel
# centre u = x/scale + 0.5 (1 - 1/scale), P = ceil(kernel_width) + 2 taps starting at floor(u - kernel_width/2), weights
# normalised to sum 1, symmetric mirroring at the borders, the dimension with the smaller scale factor first, and -- for
# an integer-class image -- rounding and saturation to uint8 after EACH 1-D pass.
def _cubic(x):
    ax = np.abs(x)
    ax2 = ax * ax
    ax3 = ax2 * ax
    return np.where(ax <= 1.0, (1.5 * ax3 - 2.5 * ax2) + 1.0,
                    np.where(ax <= 2.0, ((-0.5 * ax3 + 2.5 * ax2) - 4.0 * ax) + 2.0, 0.0))


def imresize_contributions(in_len, out_len):
    """weights [out_len, P] (float64), indices [out_len, P] (0-based, mirrored) of MATLAB's 1-D bicubic resize."""
    scale = np.float64(out_len) / np.float64(in_len)
    kw = 4.0 / scale if scale < 1.0 else 4.0
    x = np.arange(1, out_len + 1, dtype=np.float64)
    u = x / scale + 0.5 * (1.0 - 1.0 / scale)
    left = np.floor(u - kw / 2.0)
    P = int(np.ceil(kw)) + 2
    idx = left[:, None] + np.arange(P, dtype=np.float64)[None, :]          # 1-based, may be outside [1, in_len]
    d = u[:, None] - idx
    w = scale * _cubic(scale * d) if scale < 1.0 else _cubic(d)
    s = np.zeros(out_len, np.float64)
    for k in range(P):                                                     # fixed summation order (the device does the same)
        s = s + w[:, k]
    w = w / s[:, None]
    period = 2 * in_len
    m = np.mod(idx.astype(np.int64) - 1, period)                           # aux = [1..n, n..1]
    idx0 = np.where(m < in_len, m, period - 1 - m)
    return w, idx0


def _resize_u8_along(img, axis, out_len):
    w, idx = imresize_contributions(img.shape[axis], out_len)
    a = np.moveaxis(img, axis, 0).astype(np.float64)
    acc = np.zeros((out_len,) + a.shape[1:], np.float64)
    for k in range(w.shape[1]):
        acc = acc + w[:, k].reshape((-1,) + (1,) * (a.ndim - 1)) * a[idx[:, k]]
    out = np.clip(np.floor(acc + 0.5), 0, 255).astype(np.uint8)            # MATLAB double -> uint8: round, saturate
    return np.moveaxis(out, 0, axis)


def imresize_u8(img, out_h, out_w):
    """MATLAB imresize(uint8 HxWxC, [out_h out_w]) restated (bicubic, antialiasing when shrinking)."""
    img = np.ascontiguousarray(img, np.uint8)
    sh, sw = out_h / img.shape[0], out_w / img.shape[1]
    order = (0, 1) if sh <= sw else (1, 0)                                 # [~, order] = sort(scale): smaller scale first
    for ax in order:
        img = _resize_u8_along(img, ax, out_h if ax == 0 else out_w)
    return img


def preprocess(img_rgb_u8, out_h, out_w, mean_bgr=(104.0, 117.0, 123.0)):
    """run_mscnn_detection.m:64-69: resize, RGB -> BGR, single, subtract the per-channel mean; returned as the net's
    input blob (1, 3, out_h, out_w) (the MATLAB permute [2 1 3] is matcaffe's column-major view of the same memory)."""
    r = imresize_u8(img_rgb_u8, out_h, out_w)
    bgr = r[:, :, ::-1].astype(np.float32) - np.asarray(mean_bgr, np.float32).reshape(1, 1, 3)
    return np.ascontiguousarray(bgr.transpose(2, 0, 1)[None])
