#!/usr/bin/env python3
"""Per-layer timing of the mscnn-7s-576 convolutions at full size (1x3x576x1920, roi_c1 at R = 700) with HIP events, with the
Winograd stage split {input transform, MFMA GEMM, output transform} and executed-FLOP rates.  A/B mode runs every variant of a
tuning knob on the same box, interleaved.

  python tools/bench_layers.py [--iters 20] [--only conv4_2]
  python tools/bench_layers.py --ab flags=0,4            # phase stagger on / off (mscnn_conv_desc::tune_flags bit 2)
  python tools/bench_layers.py --ab algo=0,1 --only conv3 # Winograd heuristic vs direct
  python tools/bench_layers.py --ab grid=0,500,750 --only conv4_2
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mscnn_amd import hipapi as hip

from tools_layers import LAYERS
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--zero", action="store_true", help="zero-filled operands (DVFS probe)")
ap.add_argument("--ab", default="", help="knob=v1,v2,...  with knob in {algo, variant, grid, flags}")
ap.add_argument("--fixed", default="", help="other knobs held fixed, e.g. variant=107,grid=1000")
ap.add_argument("--lib", default="", help="another build of libmscnn_hip.so (dev ablations)")
ap.add_argument("--pool", default="", choices=["", "both", "only"], help="run with the fused 2x2 pooling: y and the pooled blob, or the pooled blob only")
a = ap.parse_args()
if a.lib:
    hip.LIB_PATH = os.path.abspath(a.lib)
knob, vals = None, [0]
if a.ab:
    knob, v = a.ab.split("=")
    vals = [int(t) for t in v.split(",")]
torch.manual_seed(0)
for name, N, Cin, H, W, Cout, k, pad in LAYERS:
    if a.only and a.only not in name:
        continue
    x = torch.relu(torch.randn(N, Cin, H, W, device="cuda")); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    if a.zero:
        x.zero_(); w.zero_()
    plans = []
    for v in vals:
        kw = dict(algo=0, tune_variant=0, tune_grid=0, tune_flags=0)
        names = {"algo": "algo", "variant": "tune_variant", "grid": "tune_grid", "flags": "tune_flags"}
        for kv in filter(None, a.fixed.split(",")):
            kw[names[kv.split("=")[0]]] = int(kv.split("=")[1])
        if knob:
            kw[{"algo": "algo", "variant": "tune_variant", "grid": "tune_grid", "flags": "tune_flags"}[knob]] = v
        p = hip.ConvPlan(N, Cin, H, W, Cout, k, k, (pad, pad), relu=True, **kw)
        p.pack(w)
        p.set_profiling(True)
        plans.append(p)
    y = plans[0].forward(x, b)
    if a.pool:
        import ctypes as C
        yp = torch.empty((N, Cout, (y.shape[2] + 1) // 2, (y.shape[3] + 1) // 2), device="cuda")
        def fwd(p):
            wsb = p.ws.numel() * 4 if p.ws is not None else 0
            hip._check(hip.lib().mscnn_conv2d_fwd_pool_f32(p._p, hip._dev(x), hip._dev(w), hip._dev(p.packed), hip._dev(b),
                                                           hip._dev(y) if a.pool == "both" else None, hip._dev(yp), hip._dev(p.ws), wsb, hip._stream()))
        for p in plans:
            p.forward = (lambda p_: (lambda *args, **kw_: fwd(p_)))(p)
    ms = [0.0] * len(plans); st = [[0.0, 0.0, 0.0] for _ in plans]
    for p in plans:
        for _ in range(3):
            p.forward(x, b, out=y)
    rounds = 4
    for r in range(rounds):              # interleave the variants: clock / thermal drift hits all of them alike
        for i, p in enumerate(plans):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(max(1, a.iters // rounds)):
                p.forward(x, b, out=y)
            e1.record(); torch.cuda.synchronize()
            ms[i] += e0.elapsed_time(e1) / max(1, a.iters // rounds) / rounds
            s = p.stage_ms()
            for j in range(3):
                st[i][j] += s[j] / rounds
    for i, p in enumerate(plans):
        tf = p.executed_flops / ms[i] / 1e9
        tag = f"{knob}={vals[i]}" if knob else ""
        stage = f"  [in {st[i][0]*1e3:6.1f} | gemm {st[i][1]*1e3:6.1f} ({p.executed_flops / max(st[i][1], 1e-9) / 1e9:5.1f} TF) | out {st[i][2]*1e3:6.1f}]" if p.kernel.startswith("wino") else ""
        print(f"{name:12s} {tag:10s} {p.kernel:30s} {ms[i]*1e3:9.1f} us  {tf:6.1f} TF executed = {100*tf/157.3:5.1f}% of fp32 MFMA peak"
              f"  ({p.flops / ms[i] / 1e9:6.1f} TF algorithmic){stage}", flush=True)
