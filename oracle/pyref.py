"""ctypes front-end of oracle/_ref/libmscnn_ref.so: the REFERENCE's own C++ layer sources (compiled from the reference
checkout by oracle/ref.mk against oracle/shim) behind the same call signatures as oracle/pyoracle.py.
TEST INFRASTRUCTURE ONLY.  `available()` is False where the library has not been built."""
import ctypes as C
import os

import numpy as np

from .pyoracle import NMS_MODES, BoxOutputParams, conv_out_dim, deconv_out_dim, f32p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libmscnn_ref.so")
_lib = None


def available():
    if not os.path.exists(LIB_PATH):
        return False
    try:
        lib()
        return True
    except OSError:
        return False


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        _lib = C.CDLL(LIB_PATH)
        _lib.ref_last_error.restype = C.c_char_p
        _lib.ref_box_iou.restype = C.c_float
        _lib.ref_box_iou.argtypes = [C.c_float] * 8 + [C.c_int]
    return _lib


def set_threads(n):
    """BLAS threads of the reference's cblas_sgemm (MKL through libmkl_rt); everything else in the reference's CPU path is
    single-threaded anyway.  n = 1 gives the 1-core baseline row; returns the previous maximum."""
    lib()
    mkl = C.CDLL("libmkl_rt.so.1")
    prev = mkl.MKL_Get_Max_Threads()
    mkl.MKL_Set_Num_Threads(int(n))
    return prev


def _ck(rc):
    if rc < 0:
        raise RuntimeError("reference layer failed: " + lib().ref_last_error().decode())
    return rc


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def conv2d(x, w, b=None, pad=(0, 0), stride=(1, 1), group=1):
    x, xp = _f(x); w, wp = _f(w)
    N, Cin, H, W = x.shape
    Cout, _, Kh, Kw = w.shape
    y = np.empty((N, Cout, conv_out_dim(H, Kh, pad[0], stride[0]), conv_out_dim(W, Kw, pad[1], stride[1])), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    _ck(lib().ref_conv2d(xp, wp, bp, y.ctypes.data_as(f32p), N, Cin, H, W, Cout, Kh, Kw, pad[0], pad[1], stride[0], stride[1], group))
    return y


def deconv2d(x, w=None, b=None, pad=(0, 0), stride=(1, 1), group=1, kernel=None, num_output=None, bilinear=False):
    x, xp = _f(x)
    N, Cin, H, W = x.shape
    if w is not None:
        w, wp = _f(w)
        Kh, Kw = w.shape[2:]; Cout = w.shape[1] * group
    else:
        wp = None; Kh, Kw = kernel; Cout = num_output
    y = np.empty((N, Cout, deconv_out_dim(H, Kh, pad[0], stride[0]), deconv_out_dim(W, Kw, pad[1], stride[1])), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    _ck(lib().ref_deconv2d(xp, wp, bp, y.ctypes.data_as(f32p), N, Cin, H, W, Cout, Kh, Kw, pad[0], pad[1], stride[0], stride[1],
                           group, int(bilinear)))
    return y


def pool2d(x, kernel=(2, 2), pad=(0, 0), stride=(2, 2), method="MAX"):
    x, xp = _f(x)
    N, Cc, H, W = x.shape
    oh, ow = C.c_int(), C.c_int()
    args = [N, Cc, H, W, kernel[0], kernel[1], pad[0], pad[1], stride[0], stride[1], 0 if method == "MAX" else 1]
    _ck(lib().ref_pool2d(xp, None, *args, C.byref(oh), C.byref(ow)))
    y = np.empty((N, Cc, oh.value, ow.value), np.float32)
    _ck(lib().ref_pool2d(xp, y.ctypes.data_as(f32p), *args, None, None))
    return y


def relu(x, slope=0.0):
    x, xp = _f(x)
    y = np.empty_like(x)
    _ck(lib().ref_relu(xp, y.ctypes.data_as(f32p), x.size, C.c_float(slope)))
    return y


def inner_product(x, w, b=None):
    x, xp = _f(x); w, wp = _f(w)
    M = x.shape[0]; K = int(np.prod(x.shape[1:])); Nn = w.shape[0]
    y = np.empty((M, Nn), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    _ck(lib().ref_inner_product(xp, wp, bp, y.ctypes.data_as(f32p), M, Nn, K))
    return y


def softmax(x, axis=1):
    x, xp = _f(x)
    outer = int(np.prod(x.shape[:axis])); Cc = x.shape[axis]; inner = int(np.prod(x.shape[axis + 1:]))
    y = np.empty_like(x)
    _ck(lib().ref_softmax(xp, y.ctypes.data_as(f32p), outer, Cc, inner))
    return y


def roipool(feat, rois, pooled_h, pooled_w, spatial_scale, pad_ratio=0.0):
    feat, fp = _f(feat); rois, rp = _f(rois)
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, Cc, pooled_h, pooled_w), np.float32)
    _ck(lib().ref_roipool(fp, rp, out.ctypes.data_as(f32p), R, N, Cc, H, W, pooled_h, pooled_w, C.c_float(spatial_scale), C.c_float(pad_ratio)))
    return out


def roialign(feat, rois, pooled_h, pooled_w, spatial_scale, pad_ratio=0.0):
    feat, fp = _f(feat); rois, rp = _f(rois)
    N, Cc, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, Cc, pooled_h + 1, pooled_w + 1), np.float32)
    _ck(lib().ref_roialign(fp, rp, out.ctypes.data_as(f32p), R, N, Cc, H, W, pooled_h, pooled_w, C.c_float(spatial_scale), C.c_float(pad_ratio)))
    return out


def eltwise(xs, op="SUM", coeffs=None):
    arrs = [np.ascontiguousarray(x, np.float32) for x in xs]
    n = len(arrs)
    ptrs = (f32p * n)(*[a.ctypes.data_as(f32p) for a in arrs])
    cf = (C.c_float * n)(*coeffs) if coeffs is not None and len(coeffs) else None
    y = np.empty_like(arrs[0])
    _ck(lib().ref_eltwise(ptrs, n, cf, y.ctypes.data_as(f32p), y.size, {"PROD": 0, "SUM": 1, "MAX": 2}[op]))
    return y


def boxoutput(heads, field_w, field_h, downsample, fg_thr=-5.0, iou_thr=0.65, nms_type="IOU", field_whr=2.0, field_xyr=2.0,
              max_nms_num=2000, max_post_nms_num=0, min_size=15.0, bbox_mean=None, bbox_std=None):
    hs = [np.ascontiguousarray(h, np.float32) for h in heads]
    n = len(hs)
    num, channels = hs[0].shape[:2]
    ptrs = (f32p * n)(*[h.ctypes.data_as(f32p) for h in hs])
    hh = (C.c_int * n)(*[h.shape[2] for h in hs]); ww = (C.c_int * n)(*[h.shape[3] for h in hs])
    fw = (C.c_float * n)(*field_w); fh = (C.c_float * n)(*field_h); ds = (C.c_float * n)(*downsample)
    p = BoxOutputParams(fg_thr, iou_thr, NMS_MODES[nms_type], field_whr, field_xyr, max_nms_num, max_post_nms_num, min_size, 0,
                        (C.c_float * 4)(0, 0, 0, 0), (C.c_float * 4)(1, 1, 1, 1))
    if bbox_mean is not None and bbox_std is not None and len(bbox_mean) and len(bbox_std):
        p.do_bbox_norm = 1
        p.bbox_mean = (C.c_float * 4)(*bbox_mean); p.bbox_std = (C.c_float * 4)(*bbox_std)
    cap = max(1, sum(h.shape[2] * h.shape[3] for h in hs) * num)
    rois = np.zeros((cap, 5), np.float32); props = np.zeros((cap, 6), np.float32)
    R = _ck(lib().ref_boxoutput(ptrs, hh, ww, n, num, channels, fw, fh, ds, C.byref(p), rois.ctypes.data_as(f32p),
                                props.ctypes.data_as(f32p), cap))
    return rois[:R].copy(), props[:R].copy()


def decode_bbox(bbox, prior, mean=(0, 0, 0, 0), std=(1, 1, 1, 1)):
    bbox, bp = _f(bbox); prior, pp = _f(prior)
    R = bbox.shape[0]
    out = np.empty((R, 5), np.float32)
    _ck(lib().ref_decode_bbox(bp, pp, out.ctypes.data_as(f32p), R, bbox.shape[1], (C.c_float * 4)(*mean), (C.c_float * 4)(*std)))
    return out


def box_iou(a, b, mode="IOU"):
    return float(lib().ref_box_iou(*[C.c_float(float(v)) for v in list(a) + list(b)], NMS_MODES[mode]))
