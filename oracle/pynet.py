"""CPU oracle executor for a whole deploy net: walks the layer list of the (Split-expanded) graph and applies the
restated reference layers of oracle/mscnn_oracle.c.  TEST INFRASTRUCTURE ONLY (see oracle/pyoracle.py)."""
import re

import numpy as np

from . import pyoracle as orc


def parse_param_text(text):
    """Parses the DebugString form produced by the host runtime (name: value / name { ... }) into nested dicts of lists."""
    tokens = re.findall(r'"[^"]*"|[{}]|[^\s{}]+', text)
    pos = 0

    def block():
        nonlocal pos
        d = {}
        while pos < len(tokens) and tokens[pos] != "}":
            key = tokens[pos].rstrip(":"); pos += 1
            if tokens[pos] == "{":
                pos += 1
                val = block()
                pos += 1
            else:
                val = tokens[pos].strip('"'); pos += 1
            d.setdefault(key, []).append(val)
        return d
    return block()


def _num(d, key, default, idx=0):
    return float(d[key][idx]) if key in d and len(d[key]) > idx else default


def _conv_geom(p):
    if "kernel_h" in p:
        kh, kw = int(_num(p, "kernel_h", 0)), int(_num(p, "kernel_w", 0))
    else:
        ks = [int(float(v)) for v in p["kernel_size"]]
        kh, kw = ks[0], ks[-1]
    if "pad_h" in p or "pad_w" in p:
        ph, pw = int(_num(p, "pad_h", 0)), int(_num(p, "pad_w", 0))
    else:
        ps = [int(float(v)) for v in p.get("pad", [0])]
        ph, pw = ps[0], ps[-1]
    if "stride_h" in p:
        sh, sw = int(_num(p, "stride_h", 1)), int(_num(p, "stride_w", 1))
    else:
        ss = [int(float(v)) for v in p.get("stride", [1])]
        sh, sw = ss[0], ss[-1]
    return kh, kw, ph, pw, sh, sw, int(_num(p, "group", 1))


def forward(layers, weights, inputs, stop_after=None, backend=None, timings=None):
    """layers: list of (name, type, bottoms, tops, param_text).  weights: {layer: [w, b]}.  inputs: {blob: array}.
    Returns {blob name: array} (in-place layers overwrite their blob, like the net) plus '__anchor_ids__'.
    backend: oracle.pyoracle (default, the C restatement) or oracle.pyref (the reference's own sources, oracle/_ref).
    timings: optional list that receives (layer name, type, seconds) per layer (the `caffe time` loop, tools/caffe.cpp:380-400)."""
    global orc
    _saved = orc
    if backend is not None:
        orc = backend
    try:
        return _forward(layers, weights, inputs, stop_after, timings)
    finally:
        orc = _saved


def _forward(layers, weights, inputs, stop_after, timings=None):
    import time
    blobs = dict(inputs)
    for name, typ, bottoms, tops, ptext in layers:
        _t0 = time.perf_counter()
        P = parse_param_text(ptext)
        x = [blobs[b] for b in bottoms]
        if typ == "Input":
            pass
        elif typ == "Split":
            for t in tops:
                blobs[t] = x[0]
        elif typ == "Convolution":
            kh, kw, ph, pw, sh, sw, g = _conv_geom(P["convolution_param"][0])
            w = weights[name]
            blobs[tops[0]] = orc.conv2d(x[0], w[0], w[1] if len(w) > 1 else None, (ph, pw), (sh, sw), g)
        elif typ == "Deconvolution":
            kh, kw, ph, pw, sh, sw, g = _conv_geom(P["convolution_param"][0])
            from . import pyoracle as _po
            w = weights.get(name) or [_po.bilinear_filler((x[0].shape[1], 1, kh, kw))]
            blobs[tops[0]] = orc.deconv2d(x[0], w[0], w[1] if len(w) > 1 else None, (ph, pw), (sh, sw), g)
        elif typ == "ReLU":
            slope = _num(P.get("relu_param", [{}])[0], "negative_slope", 0.0)
            blobs[tops[0]] = orc.relu(x[0], slope)
        elif typ == "Pooling":
            p = P["pooling_param"][0]
            k = int(_num(p, "kernel_size", 0)); s = int(_num(p, "stride", 1)); pad = int(_num(p, "pad", 0))
            blobs[tops[0]] = orc.pool2d(x[0], (k, k), (pad, pad), (s, s), p.get("pool", ["MAX"])[0])
        elif typ == "InnerProduct":
            w = weights[name]
            blobs[tops[0]] = orc.inner_product(x[0], w[0].reshape(w[0].shape[0], -1), w[1] if len(w) > 1 else None)
        elif typ == "Dropout":
            blobs[tops[0]] = x[0]
        elif typ == "Concat":
            blobs[tops[0]] = np.concatenate([np.asarray(v, np.float32) for v in x], axis=1)
        elif typ == "Softmax":
            blobs[tops[0]] = orc.softmax(x[0], 1)
        elif typ == "ROIPooling":
            p = P["roi_pooling_param"][0]
            blobs[tops[0]] = orc.roipool(x[0], x[1].reshape(-1, 5), int(_num(p, "pooled_h", 0)), int(_num(p, "pooled_w", 0)),
                                         _num(p, "spatial_scale", 1.0), _num(p, "pad_ratio", 0.0))
        elif typ == "ROIAlign":
            p = P["roi_pooling_param"][0]
            blobs[tops[0]] = orc.roialign(x[0], x[1].reshape(-1, 5), int(_num(p, "pooled_h", 0)), int(_num(p, "pooled_w", 0)),
                                          _num(p, "spatial_scale", 1.0), _num(p, "pad_ratio", 0.0))
        elif typ == "Eltwise":
            p = P.get("eltwise_param", [{}])[0]
            blobs[tops[0]] = orc.eltwise(x, p.get("operation", ["SUM"])[0], [float(v) for v in p.get("coeff", [])])
        elif typ == "BoxOutput":
            p = P["box_output_param"][0]
            br = P.get("bbox_reg_param", [{}])[0]
            kw = dict(fg_thr=_num(p, "fg_thr", 0.0), iou_thr=_num(p, "iou_thr", 0.5), nms_type=p.get("nms_type", ["IOU"])[0],
                      field_whr=_num(p, "field_whr", 2.0), field_xyr=_num(p, "field_xyr", 2.0), max_nms_num=int(_num(p, "max_nms_num", 0)),
                      max_post_nms_num=int(_num(p, "max_post_nms_num", 0)), min_size=_num(p, "min_size", 15.0),
                      bbox_mean=[float(v) for v in br.get("bbox_mean", [])], bbox_std=[float(v) for v in br.get("bbox_std", [])])
            geom = ([float(v) for v in p["field_w"]], [float(v) for v in p["field_h"]], [float(v) for v in p["downsample_rate"]])
            if orc.__name__.endswith("pyref"):
                rois, props = orc.boxoutput(x, *geom, **kw)
                aids = None
            else:
                rois, props, cidx, nreal, aids = orc.boxoutput(x, *geom, with_anchor_ids=True, **kw)
            blobs[tops[0]] = rois.reshape(-1, 5, 1, 1)
            if len(tops) > 1:
                blobs[tops[1]] = props.reshape(-1, 6, 1, 1)
            blobs["__anchor_ids__"] = aids
        elif typ == "DecodeBBox":
            br = P.get("bbox_reg_param", [{}])[0]
            mean = [float(v) for v in br.get("bbox_mean", [0, 0, 0, 0])]; std = [float(v) for v in br.get("bbox_std", [1, 1, 1, 1])]
            blobs[tops[0]] = orc.decode_bbox(x[0].reshape(x[0].shape[0], -1), x[1].reshape(-1, 5), mean, std).reshape(-1, 5, 1, 1)
        else:
            raise NotImplementedError(typ)
        if timings is not None:
            timings.append((name, typ, time.perf_counter() - _t0))
        if stop_after == name:
            break
    return blobs
