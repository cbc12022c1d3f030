#!/usr/bin/env python3
"""End-to-end error of trunk / head blobs against the oracle's OWN run (errors accumulate through the layers), for A/B runs of
the convolution paths on a GPU box -- not collected by pytest:

    python tests/e2e_margin.py                                  # default: Winograd F(3x3,3x3)
    MSCNN_WINOGRAD_PLANE_M=2 python tests/e2e_margin.py         # F(2x2,3x3)
    MSCNN_WINOGRAD=0 python tests/e2e_margin.py                 # direct implicit GEMM only
"""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from mscnn_amd import net as mnet, synth, zoo
from oracle import pynet
size = dict(height=192, width=640, max_nms_num=300)
n = mnet.Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576", **size))
ws = synth.load_into(n, "mid")
x = synth.frame(size["height"], size["width"])
n.set_blob("data", x); n.forward()
layers = [(n.layer_names[i], n.layer_types[i], n.layer_bottoms(i), n.layer_tops(i), n.layer_param_text(i)) for i in range(len(n.layer_names))]
names = [l[0] for l in layers]
ref = pynet.forward(layers[:names.index("proposals")], ws, {"data": x})
for b in ("conv1_2", "conv3_3", "conv4_3", "loss1_conv1", "conv5_3", "conv6_1", "pool6", "LFCN_1_7x7", "LFCN_3_5x5"):
    a = n.get_blob(b).astype(np.float64); r = ref[b].reshape(a.shape).astype(np.float64)
    print(f"{b:12s} {n.layer_kernel(names.index(b)) if b in names and n.layer_types[names.index(b)]=='Convolution' else '':20s} end-to-end err {float((np.abs(a-r)/np.maximum(1,np.abs(r))).max()):.2e}  |max| {np.abs(r).max():.1f}")
