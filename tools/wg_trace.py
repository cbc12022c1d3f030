#!/usr/bin/env python3
"""Per-workgroup phase timeline of the igemm kernels (debug library: make -C mscnn_amd/csrc trace).

Every workgroup's thread 0 stamps s_memrealtime (100 MHz) at: kernel entry, then per segment {start, K loop done, epilogue done}.
Prints the distribution of: dispatch skew, per-segment K-loop time, epilogue time, workgroup lifetime, and how many workgroups
shared a CU.   python tools/wg_trace.py --only conv4_2 [--algo 3] [--variant V] [--grid G]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mscnn_amd import hipapi as hip
hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libmscnn_hip_trace.so")
from tools_layers import LAYERS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="conv4_2")
ap.add_argument("--algo", type=int, default=0)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--grid", type=int, default=0)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--dump", default="")
ap.add_argument("--warm", type=int, default=30)
a = ap.parse_args()
L = hip.lib()
import ctypes as C
L.mscnn_debug_set_wg_trace.argtypes = [C.c_void_p]
for name, N, Cin, H, W, Cout, k, pad in LAYERS:
    if a.only not in name:
        continue
    x = torch.relu(torch.randn(N, Cin, H, W, device="cuda")); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    p = hip.ConvPlan(N, Cin, H, W, Cout, k, k, (pad, pad), relu=True, algo=a.algo, tune_variant=a.variant, tune_grid=a.grid, tune_flags=a.flags)
    p.pack(w)
    y = p.forward(x, b)
    for _ in range(a.warm):
        p.forward(x, b, out=y)
    tr = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    L.mscnn_debug_set_wg_trace(C.c_void_p(tr.data_ptr()))
    p.forward(x, b, out=y)
    torch.cuda.synchronize()
    L.mscnn_debug_set_wg_trace(None)
    t = tr.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 1] != 0]
    G = len(t)
    hw = t[:, 0] & 0xffffffff; xcc = t[:, 0] >> 32
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7     # gfx9 HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    t0 = t[:, 1].min()
    us = lambda v: (v - t0) / 100.0      # noqa: E731
    nseg = int(((t[:, 2:14] != 0).sum(1) // 3).max())
    print(f"{name} {p.kernel} G={G} segments/wg<={nseg}")
    start = us(t[:, 1]); print(f"  dispatch: start min {start.min():.1f} med {np.median(start):.1f} max {start.max():.1f} us")
    last = np.zeros(G)
    for s in range(nseg):
        c = t[:, 2 + 3 * s: 5 + 3 * s]
        ok = c[:, 2] != 0
        kl = (c[ok, 1] - c[ok, 0]) / 100.0; ep = (c[ok, 2] - c[ok, 1]) / 100.0
        last[ok] = us(c[ok, 2])
        print(f"  seg {s}: n={ok.sum()} start med {np.median(us(c[ok,0])):.1f}  K-loop med {np.median(kl):.1f} (p10 {np.percentile(kl,10):.1f} p90 {np.percentile(kl,90):.1f} max {kl.max():.1f})"
              f"  epilogue med {np.median(ep):.1f} (p90 {np.percentile(ep,90):.1f} max {ep.max():.1f}) us")
    life = last - start
    mhz = (t[:, 15] - t[:, 14]) / np.maximum(life, 1e-3)
    print(f"  shader clock over the workgroup lifetime: median {np.median(mhz):.0f} MHz (p10 {np.percentile(mhz,10):.0f} p90 {np.percentile(mhz,90):.0f})")
    print(f"  lifetime med {np.median(life):.1f} p10 {np.percentile(life,10):.1f} p90 {np.percentile(life,90):.1f}; kernel end {last.max():.1f} us; "
          f"end p10 {np.percentile(last,10):.1f} med {np.median(last):.1f}")
    uniq, cnt = np.unique(cuid, return_counts=True)
    print(f"  CUs used {len(uniq)}; workgroups per CU histogram: {dict(zip(*np.unique(cnt, return_counts=True)))}")
    # concurrency: number of resident workgroups over time on the busiest CUs
    late = start > 1.0
    print(f"  workgroups that started > 1 us after the first: {late.sum()} (median start of those {np.median(start[late]) if late.any() else 0:.1f} us)")
    if a.dump:
        np.save(a.dump, t)
