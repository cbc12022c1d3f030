#!/usr/bin/env python3
"""Which step of a batched stream spikes, and what changed in it (ROI counts, kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mscnn_amd import net as mnet, synth, zoo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = "caltech/mscnn-7s-480"
net = mnet.Net(prototxt_text=zoo.prototxt(model, batch=B))
synth.load_into(net, "mid")
H, W = 480, 640
frames = [torch.from_numpy(np.concatenate([synth.frame(H, W, seed=1701 + i * B + b, org_hw=(480, 640)) for b in range(B)], 0)).cuda() for i in range(4)]
kw = dict(cls_id=2, ratios=(1.0, 1.0), org_hw=(480, 640))
kern0 = None
for i in range(40):
    t0 = time.perf_counter()
    net.set_blob("data", frames[i % 4]); net.forward()
    t1 = time.perf_counter()
    Rs = []
    for b in range(B):
        d, ids, R = net.detect_image(b, **kw); Rs.append(R)
    t2 = time.perf_counter()
    kern = [net.layer_kernel(j) for j in range(len(net.layer_names))]
    changed = [] if kern0 is None else [(net.layer_names[j], kern0[j], kern[j]) for j in range(len(kern)) if kern[j] != kern0[j]]
    kern0 = kern
    print(f"step {i:2d} fwd {1e3*(t1-t0):7.2f} ms det {1e3*(t2-t1):7.2f} ms R {Rs} sum {sum(Rs)} checks {net.auto_calibrate_state()[0]} {changed}")
