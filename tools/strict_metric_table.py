"""Evidence for the metric of the layers' numerical self-check (ConvolutionLayer::ErrorAgainstDirect: max |dy| / max(1, |y|, rms(y))):
on the "vgg_like" weight statistics at full size, per 3x3 trunk layer and ON THE SAME BOTTOM BLOB (the device net's own, post-ReLU),

    truth   float64 direct convolution (torch CPU)
    direct  the HIP direct / igemm kernel         (algo DIRECT)
    auto    what the plan picks by itself         (Winograd F(4x4,3x3) / F(3x3,3x3) where it pays)
    _ref    the reference's own CPU layer, compiled from /root/reference (oracle/_ref/libmscnn_ref.so: im2col + SGEMM)

under the STRICT metric max |a - truth| / max(1, |truth|) and the RELAXED one (/ max(1, |truth|, rms(truth))).  The point: where the
activations are hot (rms 10 .. 100) two fp32 summation orders of the direct form -- the HIP kernel and the reference's SGEMM -- already
differ from each other by more than 1e-4 under the strict metric, so a strict 5e-5 gate would reject the reference itself.

    python tools/strict_metric_table.py [--model kitti_car/mscnn-7s-576] [--style vgg_like] > profiles/r04_strict_metric.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscnn_amd import hipapi as hip, net as mnet, synth, zoo   # noqa: E402
from oracle import pyref                                       # noqa: E402  (a measurement tool, not the product path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="kitti_car/mscnn-7s-576")
    ap.add_argument("--style", default="vgg_like")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    a = ap.parse_args()
    size = dict(height=a.height, width=a.width) if a.height else {}
    n = mnet.Net(prototxt_text=zoo.prototxt(a.model, **size))
    ws = synth.load_into(n, "mid", style=a.style)
    H, W = n.blob_shape("data")[2:]
    n.set_blob("data", synth.frame(H, W))
    n.forward()
    have_ref = pyref.available()
    if have_ref:
        pyref.set_threads(os.cpu_count() or 1)
    torch.set_num_threads(os.cpu_count() or 1)
    print(f"# {a.model} {H}x{W}, weights '{a.style}', regime mid; every row: one 3x3 layer on the device net's own bottom blob")
    print(f"# strict = max|a - truth| / max(1, |truth|); relaxed = / max(1, |truth|, rms(truth)); truth = float64; _ref = "
          f"{'oracle/_ref/libmscnn_ref.so' if have_ref else 'NOT AVAILABLE on this box'}")
    print(f"{'layer':12s} {'kernel(auto)':22s} {'|y|max':>9s} {'rms':>7s} | strict: {'direct':>9s} {'auto':>9s} {'_ref':>9s} {'dir-vs-ref':>10s} |"
          f" relaxed: {'direct':>9s} {'auto':>9s} {'_ref':>9s} {'dir-vs-ref':>10s}")
    names, types = n.layer_names, n.layer_types
    worst = {}
    for i, (nm, ty) in enumerate(zip(names, types)):
        if ty != "Convolution" or nm.startswith("LFCN_") or nm.startswith("roi_"):
            continue
        w, b = ws[nm][0], ws[nm][1]
        if w.shape[2:] != (3, 3):
            continue
        x = n.get_blob(n.layer_bottoms(i)[0])
        N, Cin, h, wd = x.shape
        truth = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)
        truth = torch.relu(truth).numpy()
        rms = float(np.sqrt((truth ** 2).mean()))
        den_s = np.maximum(1.0, np.abs(truth))
        den_r = np.maximum(den_s, rms)
        xd, wd_, bd = (torch.from_numpy(t).cuda() for t in (x, w, b))
        outs = {}
        for key, algo in (("direct", hip.ALGO_DIRECT), ("auto", hip.ALGO_AUTO)):
            p = hip.ConvPlan(N, Cin, h, wd, w.shape[0], 3, 3, (1, 1), relu=True, algo=algo)
            p.pack(wd_)
            outs[key] = p.forward(xd, bd).cpu().numpy().astype(np.float64)
            if key == "auto":
                kname = p.kernel
            del p
        if have_ref:
            outs["_ref"] = np.maximum(pyref.conv2d(x, w, b, (1, 1)), 0).astype(np.float64)
        row = {}
        for metric, den in (("s", den_s), ("r", den_r)):
            for key in ("direct", "auto", "_ref"):
                row[metric + key] = float((np.abs(outs[key] - truth) / den).max()) if key in outs else float("nan")
            row[metric + "dvr"] = float((np.abs(outs["direct"] - outs["_ref"]) / den).max()) if have_ref else float("nan")
        for k, v in row.items():
            worst[k] = max(worst.get(k, 0.0), v) if v == v else worst.get(k, float("nan"))
        print(f"{nm:12s} {kname:22s} {np.abs(truth).max():9.1f} {rms:7.2f} |         {row['sdirect']:9.2e} {row['sauto']:9.2e} {row['s_ref']:9.2e} {row['sdvr']:10.2e} |"
              f"          {row['rdirect']:9.2e} {row['rauto']:9.2e} {row['r_ref']:9.2e} {row['rdvr']:10.2e}", flush=True)
    print(f"{'worst':12s} {'':22s} {'':9s} {'':7s} |         {worst['sdirect']:9.2e} {worst['sauto']:9.2e} {worst['s_ref']:9.2e} {worst['sdvr']:10.2e} |"
          f"          {worst['rdirect']:9.2e} {worst['rauto']:9.2e} {worst['r_ref']:9.2e} {worst['rdvr']:10.2e}")


if __name__ == "__main__":
    main()
