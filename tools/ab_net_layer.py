#!/usr/bin/env python3
"""In-net A/B of ONE layer's tuning knobs (mscnn_net_set_conv_tuning: the product library ignores MSCNN_TUNE_* environment variables):
whole forwards of the default model timed with the layer on each setting, alternating in one process.
    python tools/ab_net_layer.py --layer conv1_2 --flags 0,32768 [--variant 0,0] [--iters 100 --rounds 4]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mscnn_amd import net as mnet, synth, zoo

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="kitti_car/mscnn-7s-576")
ap.add_argument("--layer", default="conv1_2")
ap.add_argument("--flags", default="0,32768")
ap.add_argument("--variant", default="")
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()
flags = [int(v) for v in a.flags.split(",")]
variants = [int(v) for v in a.variant.split(",")] if a.variant else [0] * len(flags)
n = mnet.Net(prototxt_text=zoo.prototxt(a.model))
synth.load_into(n, "mid")
shape = n.blob_shape("data")
n.set_blob("data", synth.frame(shape[2], shape[3]))
li = n.layer_names.index(a.layer)
ms = [[] for _ in flags]
kern = [None] * len(flags)
for r in range(a.rounds + 1):
    for k, (f, v) in enumerate(zip(flags, variants)):
        n.set_conv_tuning(li, v, 0, f)
        for _ in range(3):
            n.forward()
        kern[k] = n.layer_kernel(li)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters):
            n.forward()
        torch.cuda.synchronize()
        if r > 0:
            ms[k].append((time.perf_counter() - t0) * 1e3 / a.iters)
for k, f in enumerate(flags):
    print(f"{a.layer} flags={f} variant={variants[k]} {kern[k]:34s} forward {np.median(ms[k]):.4f} ms (rounds: {' '.join('%.4f' % x for x in ms[k])})")
