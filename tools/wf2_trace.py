#!/usr/bin/env python3
"""Phase timeline of the one-launch Winograd kernel (wf2conv.hip) from the trace build (make -C mscnn_amd/csrc wf2trace): shader-clock
stamps of workgroup 0, waves 0 (transform first) and 4 (MFMAs first), units 8 .. 39 of conv1_2 at full size."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mscnn_amd import hipapi as hip
hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libmscnn_hip_wf2trace.so")
x = torch.relu(torch.randn(1, 64, 576, 1920, device="cuda")); w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
p = hip.ConvPlan(1, 64, 576, 1920, 64, 3, 3, (1, 1), relu=True)
assert p.kernel.startswith("winograd2x2_fused"), p.kernel
p.pack(w)
pool = torch.empty(1, 64, 288, 960, device="cuda")
for _ in range(5):
    p.forward(x, b, pool_out=pool)
torch.cuda.synchronize()
out = np.zeros(2 * 32 * 8, np.uint64)
assert hip.lib().mscnn_debug_wf2_trace(out.ctypes.data_as(C.c_void_p)) == 0
t = out.reshape(2, 32, 8).astype(np.int64)
print("# shader-clock cycles per unit; waves 0-3: dma + transform | mfma | epilogue | - | barrier wait;  waves 4-7: - | mfma | epilogue | dma + transform | barrier wait")
for u in range(32):
    a, bq = t[0, u], t[1, u]
    print(f"unit {u + 8:3d} c={(u + 8) % 8}  w0: dma+tr {a[1]-a[0]:6d} mfma {a[2]-a[1]:6d} epi {a[3]-a[2]:6d} wait {a[5]-a[4]:6d} | total {a[5]-a[0]:6d}"
          f"   w4: mfma {bq[2]-bq[1]:6d} epi {bq[3]-bq[2]:6d} dma+tr {bq[4]-bq[3]:6d} wait {bq[5]-bq[4]:6d} | total {bq[5]-bq[0]:6d}")
