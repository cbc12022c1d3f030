#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own layer sources (oracle/_ref/libmscnn_ref.so, built from the
reference checkout by `make -C oracle ref`).  Run in the build container only; the fixtures are committed so that the
GPU box (no reference checkout) can check both the oracle and the HIP path against real reference outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref  # noqa: E402

assert pyref.available(), "build oracle/_ref first: make -C oracle ref"
rng = np.random.default_rng(20160925)


def rois(R, H, W, batch=1):
    x1 = rng.uniform(-60, W, R); y1 = rng.uniform(-60, H, R)
    return np.stack([rng.integers(0, batch, R), x1, y1, x1 + rng.uniform(0, 600, R), y1 + rng.uniform(0, 400, R)], 1).astype(np.float32)


out = {}
# ---- BoxOutput: kitti-car geometry at 1/4 scale, dense / sparse / empty / truncated / bbox-norm
SHAPES = [(18, 60), (18, 60), (9, 30), (9, 30), (5, 15), (5, 15), (3, 8)]
FIELD = [60, 84, 120, 168, 240, 336, 480]; DS = [8, 8, 16, 16, 32, 32, 64]
cases = {"dense": (-8.0, {}), "sparse": (5.0, {}), "empty": (60.0, {}), "trunc": (-8.0, dict(max_nms_num=150)),
         "norm": (-6.0, dict(bbox_mean=[0, 0, 0, 0], bbox_std=[0.1, 0.1, 0.2, 0.2], min_size=5.0, fg_thr=-7.0))}
for name, (bg, kw) in cases.items():
    heads = []
    for (h, w) in SHAPES:
        t = rng.standard_normal((1, 9, h, w)).astype(np.float32)
        t[:, :5] *= 2; t[:, 0] += bg; t[:, 5:] *= 0.5
        t[:, 1:3, :1, :4] = 0.75
        heads.append(t)
    r, p = pyref.boxoutput(heads, FIELD, FIELD, DS, **kw)
    for j, t in enumerate(heads):
        out[f"boxout_{name}_head{j}"] = t.astype(np.float16).astype(np.float32)   # fp16-representable inputs: small files
    heads = [out[f"boxout_{name}_head{j}"] for j in range(7)]
    r, p = pyref.boxoutput(heads, FIELD, FIELD, DS, **kw)
    out[f"boxout_{name}_rois"] = r; out[f"boxout_{name}_props"] = p
    out[f"boxout_{name}_kw"] = np.array(repr(kw))
# ---- ROIPooling
feat = rng.standard_normal((2, 6, 36, 120)).astype(np.float16).astype(np.float32)
rr = rois(64, 288, 960, 2)
out["roipool_feat"] = feat; out["roipool_rois"] = rr
for tag, (ph, pw, sc, pad) in {"org": (7, 7, 0.125, 0.0), "ctx": (7, 7, 0.125, 0.25), "ped": (7, 5, 0.125, 0.25), "cal": (8, 4, 0.125, 0.0)}.items():
    out[f"roipool_{tag}"] = pyref.roipool(feat, rr, ph, pw, sc, pad)
# ---- DecodeBBox
prior = rois(100, 576, 1920); bbox = rng.standard_normal((100, 8)).astype(np.float32)
out["decode_prior"] = prior; out["decode_bbox"] = bbox
out["decode_out"] = pyref.decode_bbox(bbox, prior, (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2))
# ---- conv / pool / inner product / deconv (fp32 within 1e-4: BLAS order)
x = rng.standard_normal((1, 16, 12, 20)).astype(np.float32); w = (rng.standard_normal((24, 16, 3, 3)) * 0.1).astype(np.float32)
b = rng.standard_normal(24).astype(np.float32)
out["conv_x"] = x; out["conv_w"] = w; out["conv_b"] = b; out["conv_y"] = pyref.conv2d(x, w, b, (1, 1))
wh = (rng.standard_normal((9, 16, 7, 7)) * 0.05).astype(np.float32)
out["head_w"] = wh; out["head_y"] = pyref.conv2d(x, wh, b[:9], (3, 3))
out["pool_y"] = pyref.pool2d(x[:, :, :11, :19])            # ceil-mode edges
xi = rng.standard_normal((5, 16 * 12 * 20)).astype(np.float32); wi = (rng.standard_normal((10, 16 * 12 * 20)) * 0.02).astype(np.float32)
out["ip_x"] = xi; out["ip_w"] = wi; out["ip_y"] = pyref.inner_product(xi, wi, b[:10])
out["deconv_y"] = pyref.deconv2d(x, None, None, (1, 1), (2, 2), 16, kernel=(4, 4), num_output=16, bilinear=True)
np.savez_compressed(os.path.join(HERE, "reference_layers.npz"), **out)
print("wrote", os.path.join(HERE, "reference_layers.npz"), os.path.getsize(os.path.join(HERE, "reference_layers.npz")) // 1024, "KiB,", len(out), "arrays")
