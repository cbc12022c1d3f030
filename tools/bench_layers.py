#!/usr/bin/env python3
"""Per-layer timing of the mscnn-7s-576 trunk convolutions at full size (1x3x576x1920) with HIP events.
Usage: python tools/bench_layers.py [--iters 20] [--only conv4_2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mscnn_amd import hipapi as hip

LAYERS = [  # name, Cin, H, W, Cout, k, pad
    ("conv1_2", 64, 576, 1920, 64, 3, 1), ("conv2_1", 64, 288, 960, 128, 3, 1), ("conv2_2", 128, 288, 960, 128, 3, 1),
    ("conv3_1", 128, 144, 480, 256, 3, 1), ("conv3_2", 256, 144, 480, 256, 3, 1),
    ("conv4_1", 256, 72, 240, 512, 3, 1), ("conv4_2", 512, 72, 240, 512, 3, 1), ("conv4_3", 512, 72, 240, 512, 3, 1),
    ("conv5_1", 512, 36, 120, 512, 3, 1), ("conv6_1", 512, 18, 60, 512, 3, 1),
    ("LFCN_1_5x5", 512, 72, 240, 9, 5, 2), ("LFCN_1_7x7", 512, 72, 240, 9, 7, 3),
    ("LFCN_2_7x7", 512, 36, 120, 9, 7, 3),
]
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--only", default=""); ap.add_argument("--zero", action="store_true", help="zero-filled operands (DVFS probe)")
a = ap.parse_args()
torch.manual_seed(0)
tot_f = tot_t = 0.0
for name, Cin, H, W, Cout, k, pad in LAYERS:
    if a.only and a.only not in name: continue
    x = torch.randn(1, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    if a.zero: x.zero_(); w.zero_()
    plan = hip.ConvPlan(1, Cin, H, W, Cout, k, k, (pad, pad), relu=True); plan.pack(w)
    y = plan.forward(x, b)
    for _ in range(3): plan.forward(x, b, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.iters): plan.forward(x, b, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    tf = plan.flops / ms / 1e9
    tot_f += plan.flops; tot_t += ms
    print(f"{name:12s} {plan.kernel:28s} {ms*1e3:9.1f} us  {tf:7.1f} TFLOP/s  {100*tf/157.3:5.1f}% of fp32 MFMA peak")
print(f"total {tot_t:.3f} ms  {tot_f/tot_t/1e9:.1f} TFLOP/s")
