#!/usr/bin/env python3
"""Second golden file (layers added with the cascade / WiderFace deploys): ROIAlign, Eltwise, AVE pooling, Softmax, from the
REFERENCE's own layer sources (oracle/_ref).  Separate from make_golden.py so that the first file's random draws -- and
therefore its committed arrays -- stay untouched.

    python tests/golden/make_golden_b.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref  # noqa: E402

assert pyref.available(), "build oracle/_ref first: make -C oracle ref"
rng = np.random.default_rng(20170402)
out = {}
feat = rng.standard_normal((2, 6, 36, 120)).astype(np.float16).astype(np.float32)
R = 64
x1 = rng.uniform(-60, 960, R); y1 = rng.uniform(-60, 288, R)
rois = np.stack([rng.integers(0, 2, R), x1, y1, x1 + rng.uniform(0, 600, R), y1 + rng.uniform(0, 400, R)], 1).astype(np.float32)
rois[0] = [0, -700, -500, -400, -300]          # far outside the map
rois[1, 3] = rois[1, 1] - 3                    # x2 < x1
out["roialign_feat"] = feat; out["roialign_rois"] = rois
for tag, (ph, pw, sc, pad) in {"a": (7, 7, 0.125, 0.0), "b": (7, 7, 0.125, 0.25), "c": (4, 6, 0.25, 0.5)}.items():
    out[f"roialign_{tag}"] = pyref.roialign(feat, rois, ph, pw, sc, pad)
a = rng.standard_normal((3, 5, 7, 9)).astype(np.float32); b = rng.standard_normal((3, 5, 7, 9)).astype(np.float32)
c = rng.standard_normal((3, 5, 7, 9)).astype(np.float32)
out["elt_a"] = a; out["elt_b"] = b; out["elt_c"] = c
out["elt_sum"] = pyref.eltwise([a, b, c], "SUM")
out["elt_avg"] = pyref.eltwise([a, b], "SUM", [0.5, 0.5])
out["elt_prod"] = pyref.eltwise([a, b, c], "PROD")
out["elt_max"] = pyref.eltwise([a, b, c], "MAX")
ra = pyref.roialign(feat, rois[:9], 7, 7, 0.125, 0.0)                     # (9, 6, 8, 8): the blob AVE-pooled 2x2 / stride 1
out["ave_x"] = ra
out["ave_y"] = pyref.pool2d(ra, (2, 2), (0, 0), (1, 1), "AVE")
s = (rng.standard_normal((11, 5)) * 3).astype(np.float32)
out["softmax_x"] = s; out["softmax_y"] = pyref.softmax(s)
path = os.path.join(HERE, "reference_layers_b.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
