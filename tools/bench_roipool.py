#!/usr/bin/env python3
"""ROI pooling timing on the mscnn-7s-576 geometry (conv4_3: 512x72x240, 7x7 bins, 700 proposals, pad 0 and 0.25).
MSCNN_ROIPOOL_PERBIN=1 selects the per-output kernel for an A/B comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mscnn_amd import hipapi as hip
rng = np.random.default_rng(0)
R = 700
torch.manual_seed(0)
feat = torch.relu(torch.randn(1, 512, 72, 240, device="cuda"))
# KITTI-car-like proposals: widths 20..400 px, aspect ~0.4-0.8
w = np.exp(rng.uniform(np.log(20), np.log(400), R)); h = w * rng.uniform(0.4, 0.8, R)
x1 = rng.uniform(0, 1920 - w); y1 = rng.uniform(100, 576 - h).clip(0)
rois = torch.tensor(np.stack([np.zeros(R), x1, y1, x1 + w, y1 + h], 1).astype(np.float32), device="cuda")
out = torch.empty(R, 1024, 7, 7, device="cuda")
def both():
    hip.roipool(feat, rois, 7, 7, 0.125, 0.0, out=out, c_total=1024, c_offset=0)
    hip.roipool(feat, rois, 7, 7, 0.125, 0.25, out=out, c_total=1024, c_offset=512)
for _ in range(3): both()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): both()
e1.record(); torch.cuda.synchronize()
print(f"roipool org+ctx: {e0.elapsed_time(e1)/20*1e3:.1f} us  (perbin={os.environ.get('MSCNN_ROIPOOL_PERBIN','0')} dbg={os.environ.get('MSCNN_ROIPOOL_DBG','0')} cpb={os.environ.get('MSCNN_ROIPOOL_CPB','0')})  checksum {float(out.sum()):.6e}")
