#!/usr/bin/env python3
"""Images/sec of every deploy configuration in the zoo on one MI355X: frame resident in HBM -> forward ->
final detection stage -> detections on the host, synthetic weights / frames (tools counterpart of bench.py, which measures
the BASELINE.json configuration only).  Usage: python tools/bench_models.py [--regime mid] [--steps 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mscnn_amd import net as mnet, synth, zoo

ap = argparse.ArgumentParser(); ap.add_argument("--regime", default="mid"); ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
CLS = {"kitti_car": 2, "kitti_ped_cyc": 2, "caltech": 2}
print(f"# regime {a.regime}, {a.steps} steps after 5 warm-up, batch 1, fp32")
for model, (kw, _) in zoo.MODELS.items():
    H, W = kw["height"], kw["width"]
    n = mnet.Net(prototxt_text=zoo.prototxt(model))
    synth.load_into(n, a.regime)
    frames = [torch.from_numpy(synth.frame(H, W, seed=1701 + i)).cuda() for i in range(4)]
    dk = dict(cls_id=CLS[model.split("/")[0]], ratios=(H / 375.0, W / 1242.0), org_hw=(375, 1242))
    Rs, Ds = [], []
    def step(i):
        n.set_blob("data", frames[i % 4]); n.forward()
        dets, ids, R = n.detect(**dk)
        Rs.append(R); Ds.append(len(dets))
    for i in range(5): step(i)
    Rs.clear(); Ds.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(a.steps): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    wino = sum(1 for i in range(len(n.layer_names)) if n.layer_types[i] == "Convolution" and n.layer_kernel(i).startswith("winograd"))
    print(f"{model:34s} {H:4d}x{W:<5d} {1e3*dt:7.2f} ms/image {1/dt:7.1f} images/s   R~{np.mean(Rs):6.0f}  dets~{np.mean(Ds):5.0f}  winograd layers {wino}")
    del n, frames
