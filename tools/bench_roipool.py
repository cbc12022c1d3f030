#!/usr/bin/env python3
"""ROI pooling + roi_c1 input stage on the mscnn-7s-576 geometry (conv4_3: 512 x 72 x 240, 7 x 7 bins, pad 0 / 0.25): the two-launch
path of round 3 (roipool_rows_kernel<2> writes the R x 1024 x 7 x 7 blob, wino33_input_kernel reads it back) against round 4's fused
input stage (nchw_to_nhwc_kernel + roipool_wino33_kernel -> V), per stage by HIP events.  ROIs: tools/data/rois_7s576_mid.npy = the 676
proposals of the benchmark frame (seed 1701, "mid" regime) as the reference's CPU BoxOutput leaves them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mscnn_amd import hipapi as hip
rois_np = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "rois_7s576_mid.npy")).astype(np.float32)
R = len(rois_np)
torch.manual_seed(0)
feat = torch.relu(torch.randn(1, 512, 72, 240, device="cuda"))
rois = torch.tensor(rois_np, device="cuda")
w = torch.randn(512, 1024, 3, 3, device="cuda") * 0.01
b = torch.randn(512, device="cuda")
plan = hip.ConvPlan(R, 1024, 7, 7, 512, 3, 3, (0, 0), relu=True)
plan.pack(w)
plan.set_profiling(True)
ev = lambda: torch.cuda.Event(enable_timing=True)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = ev(), ev()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


pooled = hip.roipool_pair(feat, rois, 7, 7, 0.125, 0.0, 0.25)
y0 = plan.forward(pooled, b).clone()
y1 = plan.forward_roipool_pair(feat, rois, 0.125, 0.0, 0.25, b).clone()
print(f"R = {R}; fused == unfused bitwise: {bool(torch.equal(y0, y1))}")
t_pool = timed(lambda: hip.roipool_pair(feat, rois, 7, 7, 0.125, 0.0, 0.25))
t_conv = timed(lambda: plan.forward(pooled, b)); st_u = plan.stage_ms()
t_fused = timed(lambda: plan.forward_roipool_pair(feat, rois, 0.125, 0.0, 0.25, b)); st_f = plan.stage_ms()
print(f"unfused: roipool pair {t_pool:.1f} us + roi_c1 {t_conv:.1f} us (input transform {st_u[0]*1e3:.1f}, gemm {st_u[1]*1e3:.1f}, out {st_u[2]*1e3:.1f}) = {t_pool + t_conv:.1f} us")
print(f"fused:   roi_c1 {t_fused:.1f} us (transpose + pooling + transform {st_f[0]*1e3:.1f}, gemm {st_f[1]*1e3:.1f}, out {st_f[2]*1e3:.1f})")
