#!/usr/bin/env python3
"""Per-layer parity margins of the HIP path against the CPU oracle (not collected by pytest; run on a GPU box):

    python tests/parity_report.py [model] > profiles/rNN_parity_margins.txt

For every Convolution / InnerProduct layer of the net the oracle is fed the DEVICE's own bottom blobs, so the number is the
error of that one layer: max |gpu - oracle| / max(1, |oracle|), the metric the parity tests bound by 1e-4.  Listed with the
kernel family the plan picked (direct igemm / Winograd F(2x2,3x3) / F(3x3,3x3) / M=4 head kernel / stream-K GEMM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mscnn_amd import net as mnet, synth, zoo
from oracle import pynet

model = sys.argv[1] if len(sys.argv) > 1 else "kitti_car/mscnn-7s-576"
size = dict(height=192, width=640, max_nms_num=300)
n = mnet.Net(prototxt_text=zoo.prototxt(model, **size))
print(f"# {model} at {size['height']}x{size['width']}, synthetic weights; error = max |gpu - oracle| / max(1, |oracle|), bound 1e-4")
for regime in ("mid",):
    ws = synth.load_into(n, regime)
    n.set_blob("data", synth.frame(size["height"], size["width"]))
    n.forward()
    relu_inplace = {n.layer_bottoms(i)[0] for i in range(len(n.layer_names))
                    if n.layer_types[i] == "ReLU" and n.layer_bottoms(i) == n.layer_tops(i)}
    print(f"## regime {regime}: R = {n.blob_shape('proposals')[0]} proposals")
    for i, nm in enumerate(n.layer_names):
        t = n.layer_types[i]
        if t not in ("Convolution", "InnerProduct"):
            continue
        bots, tops = n.layer_bottoms(i), n.layer_tops(i)
        layer = (nm, t, bots, tops, n.layer_param_text(i))
        ref = pynet.forward([layer], ws, {b: n.get_blob(b) for b in bots})[tops[0]]
        if tops[0] in relu_inplace:
            ref = np.maximum(ref, 0)
        got = n.get_blob(tops[0]).astype(np.float64)
        ref = ref.reshape(got.shape).astype(np.float64)
        err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
        kern = n.layer_kernel(i) if t == "Convolution" else "gemm"
        print(f"{nm:14s} {kern:34s} |out|max {np.abs(ref).max():9.3f}   err {err:.2e}")
