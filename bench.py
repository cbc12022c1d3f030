#!/usr/bin/env python3
"""MS-CNN hot-path benchmark (BASELINE.json metric: images/sec, mscnn-7s-576 KITTI-car inference, fp32).

A step = one synthetic KITTI-shaped frame (already resident in HBM as the 1x3xHxW net input) through the whole path:
VGG-16 trunk + proposal heads (MFMA convs) -> BoxOutput (decode / top-2000 / NMS) -> 2 x ROIPooling -> roi_c1 + fc6 +
cls/bbox heads -> final bbox transform + per-class NMS, with the detections delivered to the host.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--model M] [--regime dense|mid|sparse]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)
A plain `python bench.py --gpus N` with N > 1 (no RANK / WORLD_SIZE in the environment) launches itself under
torch.distributed.run with N ranks; with fewer than N devices visible it exits non-zero instead of measuring fewer GPUs.

Multi-GPU: independent images per GPU (weak scaling, no data-path collective); per step ONE ncclAllGather of the
device-resident detection pack of every rank (libmscnn_dist.so calls RCCL directly; mscnn_amd/dist.py).

Rank 0 prints ONE JSON line:
  value / ms_per_step   whole-job images/sec over the K timed steps (max over ranks); step_ms = median / p10 / p90 per step
  roofline              the DOMINANT kernel (wgemm_kernel: the 25 / 36 plane GEMMs of the Winograd layers): FLOPs its MFMAs
                        EXECUTE / its HIP-event time / 157.3 TFLOP/s (never above 1); the conv3_1..conv5_3 block the north star
                        names is reported beside it, executed and as algorithmic-equivalent rate
  robustness            the same step with "vgg_like" weights (tap sums not zero, log-normal gains, dead filters, hot activations):
                        which Winograd layers the calibration sends to the direct kernel, the resulting images/sec, and the
                        floor with EVERY Winograd layer on the direct kernel (value_all_direct)
  cpu_baseline          oracle/_ref (the reference's own CPU layers, MKL sgemm) on one full frame, all host cores, plus a
                        1-thread row on a bounded sample; the same frame goes through the HIP path: parity_ok asserts it
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# model -> (net input H, W), original image (h, w) of the dataset, class column of the final stage
MODELS = {
    "kitti_car/mscnn-7s-576": dict(hw=(576, 1920), org_hw=(375, 1242), cls_id=2),
    "kitti_car/mscnn-7s-384": dict(hw=(384, 1280), org_hw=(375, 1242), cls_id=2),
    "kitti_car/mscnn-8s-768-trainval": dict(hw=(768, 2560), org_hw=(375, 1242), cls_id=2),
    "kitti_ped_cyc/mscnn-7s-576-2x": dict(hw=(576, 1920), org_hw=(375, 1242), cls_id=2),
    "caltech/mscnn-7s-480": dict(hw=(480, 640), org_hw=(480, 640), cls_id=2),
}
DEFAULT_MODEL = "kitti_car/mscnn-7s-576"
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
FP32_MFMA_MEASURED_TFLOPS = 154.5      # MFMA-only loop on this part (tools/micro/mfma_clock.hip, profiles/r03_micro_mfma_clock.txt): 64.00 cycles per
                                        # v_mfma_f32_32x32x2_f32 and SIMD at 2.39 GHz once the clock has ramped (tens of ms of load); the 141.7 of
                                        # round 2 was a 1 ms run on a clock still ramping (SURVEY 8d asks for this reference too)
FP16_MFMA_PEAK_TFLOPS = 2500.0         # dense fp16 / bf16 MFMA (cdna_hip_programming.md: ~2.5 PF; 16x the fp32 MFMA rate)
# fp16 mode (BASELINE config 5; no reference counterpart): per-blob error relative to the blob's rms, detection matching
# (a) blobs: max error / rms(blob) < 1e-2 against the reference's CPU path; (b) detections: fp16 noise (1e-3 .. 1e-2 of a score)
# flips which member of a cluster of near-equal candidates survives the two greedy NMS stages, so the strict one-to-one match
# (IoU >= 0.95, |dscore| <= 5e-3) is REPORTED, and the gate is coverage: every detection of either run has a partner in the other
# with IoU >= 0.5 (the NMS overlap: a swapped survivor overlaps the one it replaced by more than that) and |dscore| <= 0.05
F16_BLOB_BOUND, F16_IOU, F16_DSCORE, F16_MATCH = 1e-2, 0.95, 5e-3, 0.95
F16_COVER_IOU, F16_COVER_DSCORE, F16_COVER = 0.5, 0.05, 0.90
ROOFLINE_LAYERS = ["conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3"]
CALIBRATION_TOL = 5e-5                  # Winograd vs the direct kernel on the first frame; above it the layer falls back
PARITY_BOUND = 1e-4                     # north star: fp32 scores / boxes within 1e-4 of the reference's CPU path


def _layers_of(n):
    return [(n.layer_names[i], n.layer_types[i], n.layer_bottoms(i), n.layer_tops(i), n.layer_param_text(i))
            for i in range(len(n.layer_names))]


def _full_size_parity(net, x, blobs, kw, dtype="f32"):
    """The GPU path against the reference's own CPU layers on the SAME full-size frame and weights (the cpu_baseline run):
    end-to-end fp32 error of trunk / head / sub-net blobs and matched final detections.  Checker only."""
    from oracle import pyoracle as orc

    f16 = dtype == "f16"
    bound, iou_min, dscore, need = (F16_BLOB_BOUND, F16_IOU, F16_DSCORE, F16_MATCH) if f16 else (PARITY_BOUND, 0.99, PARITY_BOUND, 0.98)

    def err(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64).reshape(a.shape)
        if f16:     # fp16 operands: error relative to the blob's scale
            return float(np.abs(a - b).max() / max(np.sqrt((b ** 2).mean()), 1e-6))
        return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())

    net.set_blob("data", x)
    net.forward()
    heads = [b for b in net.blob_names if b.startswith("LFCN_") and "split" not in b]
    names = [b for b in ("conv3_3", "conv4_3", "conv5_3", "conv6_1") if b in blobs] + heads[1:2] + heads[-2:-1]
    errs = {b: float(f"{err(net.get_blob(b), blobs[b]):.3g}") for b in names}
    Rg, Rr = net.blob_shape("proposals")[0], blobs["proposals"].shape[0]
    dets, ids, _ = net.detect(**kw)
    dref, _ = orc.detections(blobs["bbox_pred"], blobs["cls_pred"], blobs["proposals_score"].reshape(Rr, 6), **kw)
    matched = 1.0 if len(dets) == len(dref) == 0 else 0.0
    diag, cover = {}, matched
    if len(dets) and len(dref):
        a = np.stack([dets[:, 0], dets[:, 1], dets[:, 0] + dets[:, 2], dets[:, 1] + dets[:, 3]], 1)
        b = np.stack([dref[:, 0], dref[:, 1], dref[:, 0] + dref[:, 2], dref[:, 1] + dref[:, 3]], 1)
        x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
        x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
        inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
        iou = inter / (((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None, :] - inter)
        j = iou.argmax(1)
        best_iou, ds = iou[np.arange(len(a)), j], np.abs(dets[:, 4] - dref[j, 4])
        matched = float(((best_iou >= iou_min) & (ds <= dscore)).mean())
        near = (iou >= F16_COVER_IOU) & (np.abs(dets[:, None, 4] - dref[None, :, 4]) <= F16_COVER_DSCORE)
        cover = float(min(near.any(1).mean(), near.any(0).mean()))
        diag = {"coverage_iou50_dscore05": round(cover, 4),
                "iou_min": round(float(best_iou.min()), 4), "iou_p05": round(float(np.quantile(best_iou, 0.05)), 4),
                "dscore_p50": float(f"{np.median(ds):.3g}"), "dscore_p95": float(f"{np.quantile(ds, 0.95):.3g}"),
                "dscore_max": float(f"{ds.max():.3g}")}
        # north star: "fp32 scores / boxes within 1e-4": coordinate error of the matched pairs, in the reference's own metric
        # (x, y, w, h of the output rows, run_mscnn_detection.m:100-107), beside the IoU gate
        mm = (best_iou >= iou_min) & (ds <= dscore)
        if mm.any():
            dabs = np.abs(dets[mm, :4].astype(np.float64) - dref[j[mm], :4])
            dc = dabs / np.maximum(1.0, np.abs(dref[j[mm], :4]))
            diag["dbox_max"] = float(f"{dc.max():.3g}")             # strict: every coordinate against max(1, |itself|)
            diag["dbox_p95"] = float(f"{np.quantile(dc.max(1), 0.95):.3g}")
            diag["dbox_abs_max_px"] = float(f"{dabs.max():.3g}")
            # against the box's own extent: a corner is proposal corner + delta x std x proposal size, so a bbox_pred that is 1.5e-5 off
            # moves a 600-pixel box's corner by ~1e-3 px whatever the corner's own value (x = 2 px at the left image edge included)
            ext = np.maximum(1.0, np.maximum(np.abs(dref[j[mm], :4]).max(1), dref[j[mm], 2:4].max(1)))[:, None]
            diag["dbox_vs_box_extent_max"] = float(f"{(dabs / ext).max():.3g}")
    slack = 0.05 if f16 else 0.02
    ok = (max(errs.values()) < bound and abs(Rg - Rr) <= max(2, slack * Rr) and (cover >= F16_COVER if f16 else matched >= need)
          and abs(len(dets) - len(dref)) <= max(2, slack * len(dref)) and (f16 or (diag.get("dbox_vs_box_extent_max", 0.0) < bound and diag.get("dbox_max", 0.0) < 10 * bound)))
    # the detection sub-net end to end, on the ROIs both runs selected: rows of `proposals` paired by coordinates (the device's heads differ
    # from the CPU's in the last bits, so the two top-K / NMS runs keep almost, not exactly, the same boxes in almost the same order)
    sub = _subnet_err_on_matched_rois(net, blobs, err)
    return {"ok": bool(ok), "max_err_vs_reference_cpu": errs, "bound": bound,
            "policy": ("fp16 operands: blob error / rms(blob) < 1e-2; detections: mutual coverage (IoU >= 0.5, |dscore| <= 0.05) >= 90 %, "
                       "strict one-to-one match (IoU >= 0.95, |dscore| <= 5e-3) reported in detections_matched" if f16 else
                       "fp32: |a - b| / max(1, |b|) < 1e-4; detections matched at IoU >= 0.99, |dscore| <= 1e-4, >= 98 %; box coordinates (x, y, w, h) of the "
                       "matched pairs: |d| / max(1, |coordinate|, box w, box h) < 1e-4 (dbox_vs_box_extent_max) and the strict per-coordinate "
                       "|d| / max(1, |coordinate|) (dbox_max, reported) < 1e-3"),
            "proposals_gpu": int(Rg), "proposals_reference": int(Rr), "detections_gpu": int(len(dets)),
            "detections_reference": int(len(dref)), "detections_matched": round(matched, 4), "detections_diag": diag,
            "subnet_err_vs_reference_cpu": sub}


def _subnet_err_on_matched_rois(net, blobs, err):
    """cls_pred / bbox_pred of the device against the reference's CPU run, row by row on the ROIs BOTH runs selected.  A row pair = the
    same image and box (every coordinate within 1e-3 px: the decode of head outputs that agree to ~1e-6).  The pooled bins of a pair are
    the same integers unless a coordinate sits on a rounding edge of roi_pooling_layer.cpp:71-74 -- such rows show up in `rows_above_bound`
    instead of being dropped."""
    pg = net.get_blob("proposals").reshape(-1, 5).astype(np.float64)
    pr = np.asarray(blobs["proposals"], np.float64).reshape(-1, 5)
    if not len(pg) or not len(pr):
        return {"rows_paired": 0}
    d = np.abs(pg[:, None, :] - pr[None, :, :]).max(2)
    j = d.argmin(1)
    keep = d[np.arange(len(pg)), j] <= 1e-3
    gi, ri = np.nonzero(keep)[0], j[keep]
    out = {"rows_paired": int(keep.sum()), "rows_gpu": int(len(pg)), "rows_reference": int(len(pr)),
           "same_order": bool(len(pg) == len(pr) and keep.all() and np.array_equal(ri, np.arange(len(pr))))}
    for b in ("cls_pred", "bbox_pred"):
        g = np.asarray(net.get_blob(b), np.float64).reshape(len(pg), -1)[gi]
        r = np.asarray(blobs[b], np.float64).reshape(len(pr), -1)[ri]
        e = (np.abs(g - r) / np.maximum(1.0, np.abs(r))).max(1) if len(gi) else np.zeros(0)
        out[b] = {"max": float(f"{e.max():.3g}") if len(e) else None, "p99": float(f"{np.quantile(e, 0.99):.3g}") if len(e) else None,
                  "rows_above_bound": int((e >= PARITY_BOUND).sum())}
    return out


def _batch_parity(net, mnet, zoo, synth, args, x, kw, B, device):
    """Image b of a batched forward against the batch-1 forward of the same frame (same library, same weights): detections matched
    one to one (fp32: IoU >= 0.99, |dscore| <= 1e-4, >= 98 %; fp16: the coverage policy of the fp16 mode)."""
    one = mnet.Net(prototxt_text=zoo.prototxt(args.model), device=device)
    synth.load_into(one, args.regime)
    if args.dtype != "f32":
        one.set_precision(args.dtype)
    net.set_blob("data", x)
    net.forward()
    f16 = args.dtype == "f16"
    iou_min, dscore, need = (F16_COVER_IOU, F16_COVER_DSCORE, F16_COVER) if f16 else (0.99, PARITY_BOUND, 0.98)
    worst, counts = 1.0, []
    for b in range(B):
        db, _, Rb = net.detect_image(b, **kw)
        one.set_blob("data", x[b:b + 1])
        one.forward()
        d1, _, R1 = one.detect(**kw)
        counts.append([int(Rb), int(R1), len(db), len(d1)])
        if len(db) == 0 or len(d1) == 0:
            worst = min(worst, 1.0 if len(db) == len(d1) else 0.0)
            continue
        a = np.stack([db[:, 0], db[:, 1], db[:, 0] + db[:, 2], db[:, 1] + db[:, 3]], 1)
        c = np.stack([d1[:, 0], d1[:, 1], d1[:, 0] + d1[:, 2], d1[:, 1] + d1[:, 3]], 1)
        x1 = np.maximum(a[:, None, 0], c[None, :, 0]); y1 = np.maximum(a[:, None, 1], c[None, :, 1])
        x2 = np.minimum(a[:, None, 2], c[None, :, 2]); y2 = np.minimum(a[:, None, 3], c[None, :, 3])
        inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
        iou = inter / (((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + ((c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1]))[None, :] - inter)
        near = (iou >= iou_min) & (np.abs(db[:, None, 4] - d1[None, :, 4]) <= dscore)
        worst = min(worst, float(min(near.any(1).mean(), near.any(0).mean())))
    del one
    return {"ok": bool(worst >= need), "worst_image_matched": round(worst, 4), "need": need,
            "rois_dets_per_image [batched R, batch-1 R, batched D, batch-1 D]": counts,
            "policy": f"every image of the batch against its own batch-1 forward: mutual match at IoU >= {iou_min}, |dscore| <= {dscore}"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model, regime, R_gpu, net=None, kw=None, layer_table=None, dtype="f32", alt_dtype=None, other_regimes=()):
    """The reference's CPU forward path timed on this box's host cores (rank 0, N=1 only).

    kind "reference": oracle/_ref -- the reference's OWN layer sources (im2col + cblas_sgemm through MKL, serial
    BoxOutput / ROIPooling / pooling loops) on ONE full frame with the same weights; the scope is Net::Forward only, like
    the reference's tic/toc (run_mscnn_detection.m:72).  `one_core`: the same code with one BLAS thread on the layers up to
    pool2 of the same frame (a bounded sample), scaled by the all-core run's FLOP share of those layers.
    kind "port" (when _ref is not built): the C restatement on a bounded quarter-area sample, scaled."""
    from mscnn_amd import net as mnet, synth, zoo
    from oracle import pynet, pyoracle, pyref
    H, W = MODELS[model]["hw"]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    if pyref.available():
        n = mnet.Net(prototxt_text=zoo.prototxt(model), device=-1)
        layers = _layers_of(n)
        ws = synth.weights(n.layer_names, n.layer_types, [n.param_shapes(i) for i in range(len(n.layer_names))], regime)
        x = synth.frame(H, W, seed=1701, org_hw=MODELS[model]["org_hw"])
        # BLAS thread count: the reference's CPU path is single-threaded outside cblas_sgemm (im2col, ReLU, pooling, BoxOutput,
        # ROI pooling), and MKL with every hardware thread on these skinny GEMMs is SLOWER than a few threads -- time the layers
        # up to pool2 (4 convolutions, a bounded sample) at several thread counts and run the full frame with the best one.
        cut = [l[0] for l in layers].index("pool2") + 1
        sample = {}
        prev = pyref.set_threads(1)
        try:
            for nt in (1, 8, 32):      # (round 2 also timed 128 / all threads: always slower here, and the sweep outweighed the GPU loop)
                if nt > cores:
                    continue
                pyref.set_threads(nt)
                t0 = time.perf_counter()
                pynet.forward(layers[:cut], ws, {"data": x}, backend=pyref)
                sample[nt] = time.perf_counter() - t0
            best = min(sample, key=sample.get)
            pyref.set_threads(best)
            timings = []
            t0 = time.perf_counter()
            blobs = pynet.forward(layers, ws, {"data": x}, backend=pyref, timings=timings)
            dt = time.perf_counter() - t0
        finally:
            pyref.set_threads(prev)
        R = blobs["proposals"].shape[0]
        res = {"value": round(1.0 / dt, 4), "unit": "images/sec", "cores": best, "kind": "reference", "cpu": _cpu_model(),
               "host_threads_available": cores,
               "sample": f"1 full frame (1x3x{H}x{W}, R={R} ROIs) through the reference's own CPU layers (oracle/_ref: im2col + MKL "
                         f"cblas_sgemm on {best} BLAS threads = the fastest of {sorted(sample)} on conv1_1..pool2 of this frame; everything "
                         f"else in that path is serial), Net::Forward scope, {dt:.2f} s",
               "seconds_per_image": round(dt, 2),
               "blas_threads_sample_s": {str(k): round(v, 2) for k, v in sorted(sample.items())}}
        if layer_table is not None:
            layer_table.extend(timings)
        t_all = sum(t for (nm, ty, t) in timings[:cut])
        est = dt * sample[1] / max(t_all, 1e-9)
        res["one_core"] = {"value": round(1.0 / est, 5), "unit": "images/sec", "cores": 1,
                           "sample": f"conv1_1..pool2 of the same frame with 1 BLAS thread: {sample[1]:.2f} s against {t_all:.2f} s on {best}; "
                                     f"whole frame estimated as {dt:.2f} s x that ratio = {est:.1f} s",
                           "seconds_per_image_est": round(est, 1)}
        try:      # a MEASURED one-thread full frame of this model on this CPU model, taken once (tools/cpu_one_core.py), beside the estimate
            oc = json.load(open(os.path.join(ROOT, "profiles", "r04_cpu_one_core.json")))
            if oc.get("model") == model and oc.get("cpu") == res["cpu"]:
                res["one_core"].update({"value": round(1.0 / oc["seconds_per_image"], 5), "seconds_per_image_measured": oc["seconds_per_image"],
                                        "sample": f"1 full frame, 1 BLAS thread, measured once on this CPU model ({oc['cpu']}; profiles/r04_cpu_one_core.json, "
                                                  f"tools/cpu_one_core.py): {oc['seconds_per_image']:.1f} s; this run's estimate from conv1_1..pool2: {est:.1f} s"})
        except (OSError, ValueError, KeyError):
            pass
        if net is not None:
            res["full_size_parity"] = _full_size_parity(net, x, blobs, kw, dtype)
            if alt_dtype:      # the same reference run checks the alternative precision mode
                net.set_precision(alt_dtype)
                res["full_size_parity_alt"] = _full_size_parity(net, x, blobs, kw, alt_dtype)
                net.set_precision(dtype)
            # the other regimes: the reference's trunk blobs do not depend on the regime (only the heads' class-0 bias does), so its
            # heads, BoxOutput and detection sub-net are re-run on them (~5 s each) and the device net is checked in that regime
            if other_regimes:
                tail = [l for l in layers if l[0].startswith("LFCN_")]
                tail += layers[[l[0] for l in layers].index("proposals"):]
                res["full_size_parity_regimes"] = {}
                pyref.set_threads(best)
                try:
                    for rg, other_net in other_regimes:
                        t0 = time.perf_counter()
                        if other_net is not None:      # another net on the same weights (max_rois: BoxOutput never suppresses)
                            lay_o = _layers_of(other_net)
                            tail_o = [l for l in lay_o if l[0].startswith("LFCN_")] + lay_o[[l[0] for l in lay_o].index("proposals"):]
                            blobs_r = pynet.forward(tail_o, ws, blobs, backend=pyref)
                            par = _full_size_parity(other_net, x, blobs_r, kw, dtype)
                            par["reference_tail_s"] = round(time.perf_counter() - t0, 2)
                            res["full_size_parity_regimes"][rg] = par
                            continue
                        ws_r = dict(ws)
                        for nm in ws:
                            if nm.startswith("LFCN_"):
                                b = ws[nm][1].copy(); b[0] = synth.HEAD_BIAS[rg]
                                ws_r[nm] = [ws[nm][0], b]
                        blobs_r = pynet.forward(tail, ws_r, blobs, backend=pyref)
                        synth.set_regime(net, rg)
                        par = _full_size_parity(net, x, blobs_r, kw, dtype)
                        par["reference_tail_s"] = round(time.perf_counter() - t0, 2)
                        res["full_size_parity_regimes"][rg] = par
                finally:
                    pyref.set_threads(prev)
                    synth.set_regime(net, regime)
        return res
    pyoracle.lib()
    h, w = H // 2, W // 2
    n = mnet.Net(prototxt_text=zoo.prototxt(model, height=h, width=w, max_nms_num=32), device=-1)
    layers = _layers_of(n)
    ws = synth.weights(n.layer_names, n.layer_types, [n.param_shapes(i) for i in range(len(n.layer_names))], regime)
    x = synth.frame(h, w, org_hw=MODELS[model]["org_hw"])
    cut = [l[0] for l in layers].index("proposals") + 1
    t0 = time.perf_counter()
    blobs = pynet.forward(layers[:cut], ws, {"data": x})
    t_trunk = time.perf_counter() - t0
    t0 = time.perf_counter()
    blobs = pynet.forward(layers[cut:], ws, blobs)
    t_det = time.perf_counter() - t0
    r_sample = blobs["proposals"].shape[0]
    est = 4.0 * t_trunk + t_det * (R_gpu / max(r_sample, 1))
    return {"value": round(1.0 / est, 4), "unit": "images/sec", "cores": cores, "kind": "port", "cpu": _cpu_model(),
            "sample": f"oracle restatement (im2col + k-ordered GEMM, OpenMP) on a {h}x{w} frame for trunk+heads+BoxOutput ({t_trunk:.2f} s) "
                      f"and the detection sub-net on {r_sample} ROIs ({t_det:.2f} s), scaled x4 pixels and x{R_gpu}/{r_sample} ROIs",
            "seconds_per_image_est": round(est, 2)}


def _kernel_sources_sha16():
    import hashlib
    h = hashlib.sha256()
    for f in ("mscnn_amd/csrc/wgemm.hip", "mscnn_amd/csrc/winograd.hip"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def _measured_traffic():
    """(bytes of one launch | None, where it comes from, extra fields): the newest profiles/rNN_traffic_wgemm.json whose
    kernel_sources_sha16 matches the wgemm.hip / winograd.hip of this tree."""
    sha = _kernel_sources_sha16()
    seen = []
    for rnd in ("r06", "r05", "r04"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_traffic_wgemm.json")
        try:
            t = json.load(open(path))
            if t.get("kernel_sources_sha16") != sha:
                seen.append(f"profiles/{rnd}_traffic_wgemm.json ({t.get('kernel_sources_sha16')})")
                continue
            return (int(t["traffic_bytes_per_launch"]),
                    "PMC FETCH_SIZE x 2 + WRITE_SIZE of ONE launch (conv4_2's plane GEMM), rocprofv3 separate passes, tools/pmc_traffic.py -> "
                    f"profiles/{rnd}_traffic_wgemm.json (kernel sources unchanged since: sha16 " + t["kernel_sources_sha16"] + ")",
                    {"traffic_launch": "conv4_2", "traffic_algorithmic_bytes": int(t["algorithmic_bytes_per_launch"]),
                     "traffic_over_algorithmic": round(t["traffic_bytes_per_launch"] / t["algorithmic_bytes_per_launch"], 3)})
        except (OSError, ValueError, KeyError):
            continue
    if seen:
        return None, "measured on other kernel sources: " + ", ".join(seen) + ": not reported", {}
    return None, "no PMC measurement of this kernel in profiles/ (tools/pmc_traffic.py)", {}


def _measured_mfma_busy():
    """MFMA-pipe utilisation of the plane-GEMM kernel from the PMC counters (tools/pmc_mfma.py -> profiles/rNN_mfma_busy.json), under the
    same rule as `traffic`: reported only while wgemm.hip / winograd.hip are byte-identical to the sources it was measured on."""
    sha, other = _kernel_sources_sha16(), []
    for rnd in ("r06", "r05"):
        rel = f"profiles/{rnd}_mfma_busy.json"
        try:
            t = json.load(open(os.path.join(ROOT, rel)))
        except (OSError, ValueError):
            continue
        try:
            if t.get("kernel_sources_sha16") != sha:
                other.append(f"{rel} ({t.get('kernel_sources_sha16')})")
                continue
            return {"mfma_busy": t["mfma_busy_time_weighted"],
                    "mfma_busy_per_layer": {k: {"busy": v.get("mfma_busy"), "clock_ghz": v.get("clock_ghz"), "avg_us": v.get("avg_us")} for k, v in t["layers"].items()},
                    "mfma_busy_source": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of the kernel stand-alone per layer, time-weighted "
                                        f"(rocprofv3 --pmc, tools/pmc_mfma.py -> {rel}; profiled clock in clock_ghz: busy x clock / 2.4 "
                                        "is the fraction of the 157.3 TFLOP/s peak the pipe was issued at, tile padding included)"}
        except KeyError:
            continue
    if other:
        return {"mfma_busy": None, "mfma_busy_source": "measured on other kernel sources: " + ", ".join(other) + ": not reported"}
    return {"mfma_busy": None, "mfma_busy_source": "no PMC measurement in profiles/ (tools/pmc_mfma.py)"}


class _CudaPtr:
    """A raw device allocation as a __cuda_array_interface__ object (torch.as_tensor wraps it without a copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _set_affinity_all_threads(mask):
    """sched_setaffinity on every thread the process has (worker pools created while the rank was pinned keep their mask otherwise)."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for t in tids:
        try:
            os.sched_setaffinity(t, mask)
        except OSError:
            pass


def _refuse(msg):
    print("bench.py: " + msg, file=sys.stderr, flush=True)
    sys.exit(2)


def _self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    ... bench.py <same arguments>` -- one rank per GPU, rank 0 prints the one JSON line.  Never falls back to fewer GPUs."""
    if not args.launch_check:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            _refuse(f"--gpus {args.gpus} but only {have} GPU(s) visible to this process: refusing to measure fewer GPUs than asked for")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def _launch_check(args, rank, world, local_rank):
    """Launcher self-check (tests/test_dist_cpu.py; no GPU, NOT a measurement): the ranks torch.distributed.run started rendezvous over
    gloo, build the product's exchange (mscnn_amd.dist.RcclGather -> libmscnn_dist.so) on the transport library named by --transport,
    push K synthetic detection packs through the pipelined all-gather exactly like the timed loop does, and rank 0 prints one JSON
    line with n_gpus = the ranks the collective really saw and value = null."""
    import torch.distributed as dist
    from mscnn_amd import dist as mdist, net as mnet
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.transport:
        assert mdist.dist_lib().mscnn_dist_use_transport(args.transport.encode()) == 0, mdist.dist_lib().mscnn_dist_last_error().decode()
    # host placement exactly as a GPU rank gets it, on a synthetic topology (no GPU here to ask for its PCI address): the product's
    # planner decides the slice, this process applies it
    placement = None
    if args.fake_cpulists:
        lists = args.fake_cpulists.split(";")
        assert len(lists) == world, (len(lists), world)
        cpus = mdist.plan_cpus(lists, rank)
        os.sched_setaffinity(0, mdist.parse_cpulist(cpus))
        placement = {"rank": rank, "cpus": cpus, "applied": sorted(os.sched_getaffinity(0))}

    def exchange(b):
        box = [b]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    cap = 16
    gather = mdist.RcclGather(rank, world, local_rank, cap, exchange)
    seen, inflight, packs = set(), 0, []
    t0 = time.perf_counter()
    for i in range(args.steps):
        pack = np.zeros(mnet.detect_pack_bytes(cap), np.uint8)
        pack[:16].view(np.int32)[:] = [1, 1 + rank, cap, 0]
        pack[16:56].view(np.float64)[:] = [i, rank, 1.0, 1.0, 0.5]
        packs.append(pack)                                   # (the stub's "device" memory is host memory: keep it alive until end())
        gather.begin(pack.ctypes.data)
        inflight += 1
        if inflight == 2:
            inflight -= 1
            seen.add(len(gather.end()))
    while inflight:
        inflight -= 1
        per_rank = gather.end()
        seen.add(len(per_rank))
        assert [int(d[0, 1]) for d, _, _ in per_rank] == list(range(world)), "packs out of rank order"
    assert gather.comm_world == world and gather.comm_rank == rank, (gather.comm_world, gather.comm_rank)
    assert gather.ranks_seen == set(range(world)), gather.ranks_seen
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert seen == {world}, seen
    placements = [None] * world
    dist.all_gather_object(placements, placement)
    gather_world, seen_ranks = gather.comm_world, sorted(gather.ranks_seen)
    gather.close()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "metric": "launcher self-check (no measurement)", "value": None, "n_gpus": world,
                          "steps": args.steps, "config": {"gather": f"libmscnn_dist: ncclAllGather of the device pack (pipelined, two in flight), "
                                                                      f"{gather_world} ranks in the communicator (ncclCommCount)",
                                                          "comm_count": gather_world, "ranks_seen": seen_ranks,
                                                          "transport": args.transport or "librccl", "per_rank": placements}}), flush=True)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default=DEFAULT_MODEL, choices=sorted(MODELS))
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "f16x3"],
                    help="f16: fp16 MFMA operands / fp32 accumulate for the 3x3 convolutions and fc6 (BASELINE config 5); "
                         "f16x3: the Winograd plane GEMMs on the fp16 pipe with exactly split fp32 operands (fp32-grade: same "
                         "parity gates as f32)")
    ap.add_argument("--batch", type=int, default=1,
                    help="images per forward (default 1 = the headline configuration).  B > 1: the net is built with `dim: B` (the "
                         "reference's path is batch-generic: box_output_layer.cpp:107, roi_pooling_layer.cpp:62-66), a step = one forward of B "
                         "resident frames + B per-image final stages, value = B * steps / time; reported BESIDE the batch-1 line (one-round "
                         "layers of the 480 x 640 stream config become multi-round); N = 1 only")
    ap.add_argument("--regime", default="mid", choices=["dense", "mid", "sparse"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--alt", action="store_true", help="opt-in (round 5): a second timed loop in the f16x3 mode, reported as alt_precision "
                                                       "(no SURVEY 8 row; the driver's wall time goes to the headline loop instead)")
    ap.add_argument("--no-alt", action="store_true", help="(accepted for old command lines: the f16x3 loop is opt-in now, see --alt)")
    ap.add_argument("--no-regimes", action="store_true", help="skip the dense / sparse legs (SURVEY 8d regimes beside the headline's)")
    ap.add_argument("--no-robust", action="store_true", help="skip the robustness leg (vgg_like weights: calibration fall-backs, all-direct floor)")
    ap.add_argument("--layers", action="store_true", help="print the per-layer tables (caffe time format) to stderr")
    ap.add_argument("--dump-steps", action="store_true", help="print the wall time of every timed step to stderr (outlier hunting)")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "torch"],
                    help="multi-GPU exchange: libmscnn_dist's direct ncclAllGather (default), or torch.distributed's all_gather of the "
                         "same device bytes -- the route taken by itself when the direct communicator cannot be set up")
    ap.add_argument("--gather-mode", default="pipelined", choices=["pipelined", "sync"],
                    help="multi-GPU exchange inside the timed loops: pipelined = mscnn_dist_all_gather_begin/_end (the collective and "
                         "the D2H copy of image i on the communicator's own stream under image i + 1's trunk; all K images' packs are on "
                         "the host before the closing barrier), sync = one blocking exchange per image")
    ap.add_argument("--no-pin", action="store_true",
                    help="do not confine the rank to the CPUs of its GPU's NUMA node (mscnn_dist_pin_host_thread; default: pinned, ranks "
                         "sharing a node get disjoint core slices)")
    ap.add_argument("--fake-cpulists", default="",
                    help="launch-check only: ';'-separated GPU-local CPU lists, one per rank (a synthetic topology for mscnn_dist_plan_cpus)")
    ap.add_argument("--detect-mode", default="sync", choices=["pipelined", "sync"],
                    help="N = 1, batch 1: how a frame's detections reach the host inside the timed loops.  sync (default, the headline of every "
                         "round): one blocking mscnn_net_detect per frame -- a step's wall time IS that frame's latency; pipelined (what the "
                         "N > 1 loops do with the exchange): mscnn_net_detect_begin / _end -- the final stage's pack of frame i is copied to "
                         "pinned memory behind an event and collected under frame i + 1's trunk (two in flight; all K frames' detections are "
                         "on the host before the timed region ends): + 0.6 % throughput over 300 frames (tools/sessions/r06_s8.sh), but a "
                         "step's host time no longer belongs to one frame.  The line reports the other mode beside the headline (detect_modes)")
    ap.add_argument("--tiles", default="auto", choices=["auto", "split", "whole"],
                    help="scheduling of the plane GEMM's last partial round: split = stream-K hand-off between co-resident workgroups, whole = "
                         "whole tiles only (mscnn_wgemm_force_whole_tiles: no workgroup ever waits for another; fc6 then runs the register-"
                         "staged GEMM -- what the Net falls back to after a hand-off time-out).  Measured on one GPU: whole costs 2.1 % (223.7 "
                         "-> 218.8 images/s, profiles/r06_ab_whole_tiles.txt).  auto = split at N = 1, whole at N > 1: RCCL's kernels share "
                         "the CUs with the next frame's trunk there, a time-out would cost 0.5 s and a frame, and nobody could rehearse an "
                         "8-GPU run -- 2 % for a run that can not stall (config.handoff.tiles says which)")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher self-check without a GPU (CPU tests): rendezvous + the product's exchange on --transport, no measurement")
    ap.add_argument("--transport", default="", help="collective library for mscnn_dist_use_transport (an RCCL build elsewhere, or the test stub)")
    args = ap.parse_args()
    if args.gpus < 1:
        _refuse(f"--gpus {args.gpus}")
    if args.batch < 1 or (args.batch > 1 and args.gpus > 1):
        _refuse(f"--batch {args.batch} with --gpus {args.gpus}: the batched mode is a single-GPU side table")

    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args)                                    # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:      # never a silent fall-back to another GPU count than the one asked for
        _refuse(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    if args.launch_check:
        _launch_check(args, rank, world, local_rank)
    # launched by torch.distributed.run (even with one rank): take the multi-GPU path -- process group, direct RCCL gather
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        _refuse(f"rank {rank} needs GPU {local_rank}; {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible (bench.py needs MI355X GPUs)")
    # host placement first (threads created from here on inherit it): the CPUs of this rank's GPU's NUMA node, cut into disjoint slices
    # for the ranks that share the node.  The CPU-reference leg (N = 1) gets the original mask back.
    mask0 = os.sched_getaffinity(0)
    placement = None
    if not args.no_pin:
        from mscnn_amd import dist as mdist_pin
        try:
            placement = mdist_pin.pin_host_thread(local_rank, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
            os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, placement["n_cpus"]))))
        except Exception as e:      # never a reason to lose the run: report it in the line
            placement = {"error": str(e)}
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)       # "nccl" is RCCL on ROCm: barrier + time reduction

    from mscnn_amd import net as mnet, synth, zoo
    cfg = MODELS[args.model]
    H, W = cfg["hw"]
    B = args.batch
    net = mnet.Net(prototxt_text=zoo.prototxt(args.model, batch=B), device=local_rank)
    synth.load_into(net, args.regime)
    if args.dtype != "f32":
        net.set_precision(args.dtype)
    # a handful of distinct frames per rank (B > 1: of B-frame blobs), resident in HBM before the timed region
    host_frames = [np.concatenate([synth.frame(H, W, seed=1701 + 97 * rank + i * B + b, org_hw=cfg["org_hw"]) for b in range(B)], 0) for i in range(4)]
    frames = [torch.from_numpy(f).cuda() for f in host_frames]
    kw = dict(cls_id=cfg["cls_id"], ratios=(H / cfg["org_hw"][0], W / cfg["org_hw"][1]), org_hw=cfg["org_hw"])
    cap = int(zoo.MODELS[args.model][0].get("max_nms_num", 2000))          # BoxOutput's top-K bounds the ROI count
    gather, gather_kind = None, "none"
    if dist is not None:
        from mscnn_amd import dist as mdist

        def exchange(b):
            box = [b]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        try:
            if args.gather == "torch":
                raise RuntimeError("--gather torch")
            gather = mdist.RcclGather(rank, world, local_rank, cap, exchange)
            gather_kind = "libmscnn_dist: ncclAllGather of the device pack"
        except Exception as e:      # a second route to the same bytes, so that the scaling run is never lost
            print(f"[rank {rank}] direct RCCL gather unavailable ({e}); using torch.distributed", file=sys.stderr)
            tg = mdist.TorchGather(cap, "cuda")
            pb = mnet.detect_pack_bytes(cap)
            gather = lambda ptr: tg(torch.as_tensor(_CudaPtr(ptr, pb), device="cuda"))   # noqa: E731
            gather_kind = "torch.distributed all_gather_into_tensor (RCCL) of the device pack"
        flags = [gather_kind.startswith("libmscnn_dist")]
        allf = [None] * world
        dist.all_gather_object(allf, flags[0])
        assert all(f == allf[0] for f in allf), "ranks disagree on the gather route"
    stats = {"R": [], "D": []}
    pipe = {"on": False, "inflight": 0, "gather_s": []}
    local_pipe = gather is None and B == 1 and args.detect_mode == "pipelined"      # N = 1: the final stage pipelined like the exchange
    can_pipeline = (hasattr(gather, "begin") and args.gather_mode == "pipelined") or local_pipe
    comm_count = getattr(gather, "comm_world", None)      # ncclCommCount of the product's communicator (None: the torch route / N = 1)
    if gather is not None:
        gather_kind += (" (pipelined, two in flight)" if can_pipeline else " (blocking)") + \
            (f", {comm_count} ranks in the communicator (ncclCommCount)" if comm_count is not None else f", {world} ranks (torch.distributed)")
        assert comm_count in (None, world), (comm_count, world)

    def take(per_rank):
        # every exchange of the timed loop: as many packs as the communicator has ranks, slot r stamped by rank r (RcclGather._split
        # raises otherwise) -- the line can only be printed if RCCL really delivered N ranks' packs every step
        assert len(per_rank) == (world if gather is not None else 1), (len(per_rank), world)
        dets, ids, R = per_rank[rank if gather is not None else 0]
        stats["R"].append(R); stats["D"].append(len(dets))
        return dets

    def step(i):
        net.set_blob("data", frames[i % len(frames)])        # D2D: the frame is already in HBM
        net.forward()
        if B > 1:                                             # one final stage per image (its NMS never mixes images)
            out = None
            for b in range(B):
                out = take([net.detect_image(b, **kw)])
            return out
        if gather is None:
            if pipe["on"] and local_pipe:                     # frame i's detections are collected under frame i + 1's trunk
                net.detect_begin(cap, **kw)
                pipe["inflight"] += 1
                if pipe["inflight"] == 2:
                    pipe["inflight"] -= 1
                    return take([net.detect_end(cap)])
                return None
            return take([net.detect(**kw)])                   # final stage on device; detections land on the host
        if pipe["on"]:                                        # final stage into the device pack; exchange i runs under image i + 1
            ptr = net.detect_device(cap, **kw)
            tg = time.perf_counter()
            gather.begin(ptr)
            pipe["inflight"] += 1
            out = None
            if pipe["inflight"] == 2:
                pipe["inflight"] -= 1
                out = take(gather.end())
            pipe["gather_s"].append(time.perf_counter() - tg)      # host time inside the exchange calls (begin + the wait in end)
            return out
        ptr = net.detect_device(cap, **kw)
        tg = time.perf_counter()
        out = take(gather(ptr))                               # one blocking all_gather, packs on the host
        pipe["gather_s"].append(time.perf_counter() - tg)
        return out

    def drain():                                              # the last image's packs (inside the timed region, before the barrier)
        while pipe["inflight"]:
            pipe["inflight"] -= 1
            take(gather.end() if gather is not None else [net.detect_end(cap)])

    def timed_loop(nsteps, warm=3, mode=None):
        """`warm` untimed steps, then `nsteps` steps between two sync()s with the loop's detections all on the host before the second:
        the contract of the headline loop, for the side legs (regimes, robustness, the other detect mode).  Returns seconds."""
        on = can_pipeline if mode is None else (mode == "pipelined" and can_pipeline)
        for i in range(warm):
            step(i)
        sync()
        pipe["on"] = on
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i)
        drain()
        pipe["on"] = False
        sync()
        return time.perf_counter() - t0

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def numerics_since(mark):
        """What the Net's OWN first-forward checks did since `mark` (no calibration call is made anywhere in this file: every Winograd
        layer compares itself with the direct kernel on the first frame after a weight / algorithm change and falls back by itself)."""
        checks, switched, errs = net.auto_calibrate_state()
        return {"winograd_layers": checks - mark[0], "max_err_vs_direct_kernel": float(f"{max(errs.values(), default=0.0):.3g}"),
                "tol": CALIBRATION_TOL, "fallback_layers": switched[mark[1]:],
                "how": "automatic (Net default): first forward after a weight change, tol 5e-5; numerics watch every 25 frames (one band of one layer, deferred verdict)"}

    def numerics_mark():
        checks, switched, _ = net.auto_calibrate_state()
        return (checks, len(switched))

    from mscnn_amd import hipapi as _hipapi
    tiles_mode = args.tiles if args.tiles != "auto" else ("whole" if world > 1 else "split")
    if tiles_mode == "whole":
        _hipapi.wgemm_force_whole_tiles(True)
    handoff_ev0 = _hipapi.wgemm_handoff_event()      # the device's hand-off status word before any frame
    numerics = None
    mark = (0, 0)
    for i in range(max(args.warmup, 1)):
        step(i)
        if i == 0:      # the first frame after the weights were loaded: the layers' own Winograd-vs-direct checks ran inside it (untimed)
            numerics = numerics_since(mark)
    stats = {"R": [], "D": []}
    # (the interpreter's cyclic garbage collector is kept out of the timed loops: a full collection over the modules torch imports
    # takes ~40 ms and showed up as ONE 40 - 55 ms step around the 144th final-stage call of a batched stream -- tools/sessions/r05_s23.sh;
    # it is the harness's, not the library's)
    import gc
    gc.collect()
    gc.disable()
    sync()
    step_s = []
    pipe["on"] = can_pipeline
    pipe["gather_s"] = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        step(i)                                               # ends with the detections on the host (stream-synchronised)
        step_s.append(time.perf_counter() - ts)
    drain()
    pipe["on"] = False
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    gc.enable()
    if args.dump_steps and rank == 0:
        print("step_ms: " + " ".join(f"{1e3 * v:.2f}" for v in step_s), file=sys.stderr)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    # every rank's own view of the timed loop (N > 1): where a scaling loss comes from -- a slow rank (placement, clocks), or the exchange
    per_rank = None
    if dist is not None:
        ss_r, gs_r = np.sort(np.array(step_s)) * 1e3, np.sort(np.array(pipe["gather_s"] or [0.0])) * 1e6
        mine = {"rank": rank, "step_ms_p50": round(float(np.median(ss_r)), 4), "step_ms_p90": round(float(ss_r[int(round(0.9 * (len(ss_r) - 1)))]), 4),
                "step_ms_max": round(float(ss_r[-1]), 4), "gather_host_us_p50": round(float(np.median(gs_r)), 1),
                "gather_host_us_p90": round(float(gs_r[int(round(0.9 * (len(gs_r) - 1)))]), 1), "loop_s": round(elapsed_local, 5),
                "placement": placement}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    main_stats, stats = stats, {"R": [], "D": []}      # the headline loop's ROI / detection counts (later loops append to their own)

    # ---- the other detect mode beside the headline (N = 1, batch 1): a short loop each way, same contract
    detect_modes = None
    if gather is None and B == 1:
        ds = max(10, min(args.steps, 50))
        other = "sync" if local_pipe else "pipelined"
        keep = stats
        stats = {"R": [], "D": []}
        local_pipe_saved = local_pipe
        if other == "pipelined":
            local_pipe, can_pipeline = True, True
        dt_o = timed_loop(ds, mode=other)
        local_pipe = local_pipe_saved
        can_pipeline = (hasattr(gather, "begin") and args.gather_mode == "pipelined") or local_pipe
        stats = keep
        detect_modes = {"headline": args.detect_mode, other: {"value": round(ds / dt_o, 3), "unit": "images/sec", "steps": ds},
                        "what": "pipelined = mscnn_net_detect_begin / _end (frame i's detections collected under frame i + 1's trunk, two in flight, "
                                "all on the host before the timed region ends; the N > 1 loops do the same with the exchange); sync = one blocking "
                                "mscnn_net_detect per frame (the headline of rounds 1 - 5)"}

    # ---- second timed loop, same contract, in the split-fp16 mode (fp32-grade: held to the fp32 parity gates below).  The
    # headline `value` stays the true-fp32-MFMA path; this is reported beside it as `alt_precision`.
    alt = None
    if args.dtype == "f32" and args.alt and not args.no_alt:
        mark = numerics_mark()
        net.set_precision("f16x3")
        a_num = None
        for i in range(max(3, args.warmup // 2)):
            step(i)
            if i == 0:      # the same contract as the fp32 path: the switch re-armed every layer's first-forward check
                a_num = numerics_since(mark)
        sync()
        a_s = []
        pipe["on"] = can_pipeline
        t0 = time.perf_counter()
        for i in range(args.steps):
            ts = time.perf_counter()
            step(i)
            a_s.append(time.perf_counter() - ts)
        drain()
        pipe["on"] = False
        sync()
        a_el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([a_el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            a_el = float(t.item())
        x3_layers = [net.layer_names[i] for i in range(len(net.layer_names)) if net.layer_dtype(i) == "f16x3"]
        net.set_precision("f32")
        for l in (numerics or {}).get("fallback_layers", []):
            net.set_conv_algo(l, 1)
        step(0)                                               # re-plan / re-pack the fp32 kernels before the roofline passes
        sync()
        alt = {"dtype": "f16x3", "value": round(world * args.steps / a_el, 3), "unit": "images/sec", "ms_per_step": round(1e3 * a_el / args.steps, 4),
               "step_ms_median": round(float(np.median(a_s)) * 1e3, 4), "vs_fp32_path": round(world * args.steps / a_el / value, 3),
               "layers": x3_layers, "numerics": a_num,
               "what": "the same step with every MFMA-bound layer on the fp16 MFMA pipe, each fp32 operand split exactly into fp16 hi + lo "
                       "(three MFMAs per pair, fp32 accumulate; scales from max |x| measured on the device each frame): conv1_2 .. "
                       "conv2_2 as a direct 3x3 implicit GEMM, conv3_1 .. conv6_1 and roi_c1 as Winograd F(3x3,3x3) plane GEMMs, fc6 as a "
                       "k-split GEMM.  22-bit operands: held to the fp32 parity gates (parity_ok in this object)"}

    # ---- the other two regimes of SURVEY 8(d) beside the headline's (N = 1, batch 1, fp32): the same weights with the heads' class-0 bias
    # of the regime -- "dense": every one of the 45,630 anchors passes fg_thr (N_pre = top-K = 2000: the worst case for the sort and the NMS),
    # "sparse": ~5 % pass (fewer candidates than the top-K).  Timed like the headline (a short loop each), R recorded; each regime's parity
    # against the reference's CPU layers is asserted in the cpu_baseline leg below (the trunk blobs of the reference run are shared).
    regimes, net_cap = None, None
    if args.dtype == "f32" and world == 1 and B == 1 and not args.no_regimes:
        regimes = {args.regime: {"value": round(value, 3), "unit": "images/sec", "steps": args.steps, "mean_rois": round(float(np.mean(main_stats["R"])), 1),
                                 "mean_detections": round(float(np.mean(main_stats["D"])), 1), "headline": True}}
        rs = max(5, min(args.steps, 30))
        for rg in ("dense", "mid", "sparse"):
            if rg == args.regime:
                continue
            synth.set_regime(net, rg)
            stats = {"R": [], "D": []}
            dt = timed_loop(rs)
            stats = {"R": stats["R"][-rs:], "D": stats["D"][-rs:]}
            regimes[rg] = {"value": round(rs / dt, 3), "unit": "images/sec", "steps": rs, "mean_rois": round(float(np.mean(stats["R"])), 1),
                           "mean_detections": round(float(np.mean(stats["D"])), 1)}
        synth.set_regime(net, args.regime)
        step(0)
        sync()
        # On these synthetic heads the three regimes end at nearly the same R: the top-K cap (2000) binds in all of them (5 % of 45,630
        # anchors is still 2,281) and the NMS keeps ~700 of the 2000.  The sub-net's worst case is R = the cap itself; no choice of
        # WEIGHTS reaches it, so it is measured on the same deploy with BoxOutput's iou_thr raised above 1 (no box is ever suppressed:
        # R = 2000 whatever the data; the greedy scan runs its longest form too) -- a bound, not a deploy the reference ships.
        net_main = net
        net_cap = mnet.Net(prototxt_text=zoo.prototxt(args.model, batch=B, iou_thr=1.01), device=local_rank)
        synth.load_into(net_cap, args.regime)
        net = net_cap
        stats = {"R": [], "D": []}
        dt = timed_loop(rs)
        stats = {"R": stats["R"][-rs:], "D": stats["D"][-rs:]}
        regimes["max_rois"] = {"value": round(rs / dt, 3), "unit": "images/sec", "steps": rs, "mean_rois": round(float(np.mean(stats["R"])), 1),
                               "mean_detections": round(float(np.mean(stats["D"])), 1),
                               "what": "the same deploy and weights with box_output_param.iou_thr = 1.01 (no proposal suppressed): R = the top-K cap, the "
                                       "upper bound of the detection sub-net's work for any weights -- not a configuration the reference ships"}
        net = net_main
        stats = {"R": [], "D": []}

    # ---- robustness leg (untimed for the headline; N = 1 only): the same step with "vgg_like" weights -- tap sums not zero, log-normal
    # per-filter gains, dead filters, biases, activations ~4x hotter (mscnn_amd/synth.py).  The Winograd forms carry ~10x the rounding
    # error of the direct sum; the calibration step decides per layer on THIS data which ones stay.  Reported: the layers that fell
    # back, the images/sec that results, and the floor with every Winograd layer on the direct kernel.
    robust = None
    if args.dtype == "f32" and world == 1 and not args.no_robust and B == 1:
        rs = max(5, min(args.steps, 20))

        def timed(nsteps):
            return nsteps / timed_loop(nsteps)
        he_fallbacks = list((numerics or {}).get("fallback_layers", []))
        mark = numerics_mark()
        synth.load_into(net, args.regime, style="vgg_like")
        net.set_conv_algo(-1, 0)                              # every convolution back to AUTO (clears calibration marks)
        step(0)                                               # first frame on the new weights: the layers check themselves
        r_num = numerics_since(mark)
        r_errs, r_sw = {i: 0 for i in range(r_num["winograd_layers"])}, r_num["fallback_layers"]
        v_cal = timed(rs)
        r_rois = float(np.mean(stats["R"][-rs:]))
        wino_layers = [net.layer_names[i] for i in range(len(net.layer_names)) if net.layer_kernel(i).startswith("winograd")]
        for l in wino_layers:
            net.set_conv_algo(l, 1)
        v_dir = timed(rs)
        robust = {"weights": "vgg_like (mscnn_amd/synth.py: centre-weighted taps + a low-pass part, log-normal filter gains, 3 % dead filters, "
                             "biases, conv1_1 scaled for activations of rms ~4 instead of ~1), same frames (BGR - mean, [-123, 151])",
                  "value": round(v_cal, 3), "unit": "images/sec", "steps": rs, "mean_rois": round(r_rois, 1),
                  "winograd_layers_checked": len(r_errs), "max_err_vs_direct_kernel": r_num["max_err_vs_direct_kernel"],
                  "tol": CALIBRATION_TOL, "fallback_layers": r_sw,
                  "value_all_direct": round(v_dir, 3), "all_direct_layers": wino_layers + r_sw,
                  "note": "value = after the Net's automatic first-forward checks on this data; value_all_direct = every Winograd layer forced onto the direct "
                          "implicit-GEMM kernel (the floor of the fp32 path, whatever the data)"}
        synth.load_into(net, args.regime)                     # back to the headline configuration for the roofline passes
        net.set_conv_algo(-1, 0)
        for l in he_fallbacks:
            net.set_conv_algo(l, 1)
        step(0)
        sync()
        stats = {"R": [], "D": []}

    result = None
    rc = 0
    if rank == 0:
        # ---- roofline: HIP events on the net's stream, outside the timed region --------------------------------------------
        # two passes: per-layer times with plain HIP events around every layer, then the per-stage split of the convolution
        # plans (their extra event records would inflate the small layers of the first table)
        net.set_layer_timing(True)
        L = len(net.layer_names)
        acc = np.zeros(L); stage = np.zeros((L, 3)); reps = 5; stage_reps = []
        for i in range(reps):
            net.set_blob("data", frames[i % len(frames)])
            net.forward()
            acc += np.array(net.layer_ms())
        net.set_conv_profiling(True)
        for i in range(reps):
            net.set_blob("data", frames[i % len(frames)])
            net.forward()
            stage_reps.append(np.array([net.layer_stage_ms(j) for j in range(L)]))
            stage += stage_reps[-1]
        net.set_layer_timing(False)
        net.set_conv_profiling(False)
        lay_ms, stage = acc / reps, stage / reps
        flops = np.array([net.layer_flops(i) for i in range(L)])
        xflops = np.array([net.layer_executed_flops(i) for i in range(L)])
        kern = [net.layer_kernel(i) for i in range(L)]
        wino = [i for i in range(L) if kern[i].startswith("winograd_f")]      # F(3x3,3x3) and F(4x4,3x3) layers: the same GEMM kernel
        idx = [net.layer_names.index(nm) for nm in ROOFLINE_LAYERS]
        dtypes = [net.layer_dtype(i) for i in range(L)]
        conv16 = [i for i in range(L) if dtypes[i] == "f16" and net.layer_types[i] == "Convolution"]
        if args.dtype == "f16":     # dominant kernel family = the fp16 implicit-GEMM 3x3 kernels (one launch per layer + fix-up)
            wino = conv16
        # dominant kernel = the MFMA GEMM of the F(3x3,3x3) layers: one launch per layer
        g_flops, g_ms = float(xflops[wino].sum()), float(stage[wino, 1].sum())
        achieved = g_flops / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        # (per-layer device time: the plan's own stage events where it has them -- they bracket exactly the layer's kernels; the
        # net-level per-layer events also contain the host's launch gap after each per-layer synchronisation)
        dev_ms = np.where(stage.sum(axis=1) > 0, stage.sum(axis=1), lay_ms)
        blk_ms = float(dev_ms[idx].sum())
        blk_exec = float(xflops[idx].sum()) / (blk_ms * 1e-3) / 1e12
        blk_alg = float(flops[idx].sum()) / (blk_ms * 1e-3) / 1e12
        conv_idx = [i for i, t in enumerate(net.layer_types) if t == "Convolution"]
        # HBM bytes of the dominant kernel are a PMC measurement (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes): not
        # something this process can take of itself.  The line carries the latest such measurement (tools/pmc_traffic.py -> profiles/)
        # only while the kernel sources it was taken on are byte-identical to the ones running now; else null.
        traffic, tsrc, textra = _measured_traffic()
        if args.model != DEFAULT_MODEL or args.dtype != "f32":      # (the measured launch is conv4_2 of the default model's fp32 path)
            traffic, tsrc, textra = None, "the PMC measurement in profiles/ is of the default model's fp32 plane GEMM (conv4_2): not reported for this run", {}
        peak = FP16_MFMA_PEAK_TFLOPS if args.dtype in ("f16", "f16x3") else FP32_MFMA_PEAK_TFLOPS
        roofline = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": tsrc, **textra,
            **(_measured_mfma_busy() if args.model == DEFAULT_MODEL and args.dtype == "f32" else {}),
            **({"frac_of_measured_mfma_peak": round(achieved / FP32_MFMA_MEASURED_TFLOPS, 4),
                "measured_mfma_peak": FP32_MFMA_MEASURED_TFLOPS} if args.dtype == "f32" else {}),
            "kernel": ("igemm_kernel<Cfg<...,F16>> (igemm16_*): direct 3x3 implicit GEMM on v_mfma_f32_32x32x16_f16, operands rounded to "
                       "fp16 while staged into LDS, fp32 accumulate (+ its stream-K fix-up)") if args.dtype == "f16" else
                      ("x3_gemm_kernel<128|256> -- the 25 plane GEMMs of the Winograd F(3x3,3x3) layers on v_mfma_f32_32x32x16_f16 with "
                       "every fp32 operand split exactly into fp16 hi + lo: three MFMAs per product pair (all three counted as "
                       "executed FLOPs), fp32 accumulate") if args.dtype == "f16x3" else
                      "wgemm_kernel<256x128 | 256x160 | 128x256 | 256x96 | 128x128, ck32> (mscnn_amd/csrc/wgemm.hip) -- the batched [Cout x Cin] x [Cin x tiles] GEMMs of the "
                      "Winograd layers (25 planes F(3x3,3x3), 36 planes F(4x4,3x3)) on v_mfma_f32_32x32x2_f32: 8 waves per CU, operands by "
                      "LDS-DMA into a 3-stage ring (+ its fix-up launch where the tile count is split stream-K style)",
            "flops_note": "achieved = FLOPs the kernel's MFMAs execute for the real problem (2 * planes * Cout * Cin * tiles per launch) / "
                          "its HIP-event time on the net's stream, summed over its launches of one image",
            **({"fp32_equivalent_tflops": round(achieved / 3, 2),
                "fp32_equivalent_vs_fp32_mfma_peak": round(achieved / 3 / FP32_MFMA_PEAK_TFLOPS, 4)} if args.dtype == "f16x3" else {}),
            "launches_per_image": len(wino), "avg_launch_us": round(1e3 * g_ms / max(len(wino), 1), 1),
            "gemm_ms_over_passes": {"min": round(float(min(r[wino, 1].sum() for r in stage_reps)), 4),
                                    "max": round(float(max(r[wino, 1].sum() for r in stage_reps)), 4), "passes": reps},
            "executed_gflop_per_image": round(g_flops / 1e9, 2),
            "layers": [net.layer_names[i] for i in wino],
            "conv3_5_block": {"layers": ROOFLINE_LAYERS, "ms_per_image": round(blk_ms, 4),
                              "executed_tflops": round(blk_exec, 2), "executed_frac": round(blk_exec / peak, 4),
                              "algorithmic_equiv_tflops": round(blk_alg, 2),
                              "algorithmic_gflop_per_image": round(float(flops[idx].sum()) / 1e9, 2),
                              "note": "all kernels of the nine layers (input transform + GEMM + output transform); executed = the MFMA "
                                      "FLOPs really issued; algorithmic_equiv = the reference's direct-convolution 2*MACs / the same time"},
            "winograd_stage_ms": {"input_transform": round(float(stage[wino, 0].sum()), 4), "gemm": round(g_ms, 4),
                                  "output_transform": round(float(stage[wino, 2].sum()), 4)},
            "all_conv_executed_tflops": round(float(xflops[conv_idx].sum()) / (float(lay_ms[conv_idx].sum()) * 1e-3) / 1e12, 2)}
        if args.layers:
            for i, nm in enumerate(net.layer_names):
                if lay_ms[i] > 0:
                    tf = xflops[i] / (lay_ms[i] * 1e-3) / 1e12 if xflops[i] else 0
                    st = f" [in {stage[i, 0]*1e3:6.1f} gemm {stage[i, 1]*1e3:6.1f} out {stage[i, 2]*1e3:6.1f}]" if stage[i].sum() > 0 and kern[i].startswith("wino") else ""
                    print(f"{nm:28s} {net.layer_types[i]:14s} {kern[i]:30s} {lay_ms[i]*1e3:9.1f} us {tf:7.1f} TF executed{st}", file=sys.stderr)
        Rm = float(np.mean(main_stats["R"]))
        ss = np.sort(np.array(step_s)) * 1e3
        result = {"metric": f"images/sec {args.model.split('/')[-1]} {args.model.split('/')[0].replace('_', '-')} inference", "value": round(value, 3),
                  "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
                  "step_ms": {"median": round(float(np.median(ss)), 4), "p10": round(float(ss[int(0.10 * (len(ss) - 1))]), 4),
                              "p90": round(float(ss[int(round(0.90 * (len(ss) - 1)))]), 4), "min": round(float(ss[0]), 4),
                              "p99": round(float(ss[int(round(0.99 * (len(ss) - 1)))]), 4),
                              "max": round(float(ss[-1]), 4), "note": "rank 0, wall time per step incl. the host sync at its end"},
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                  "config": {"workload": f"{args.model} {'fp16 MFMA operands / fp32 accumulate' if args.dtype == 'f16' else 'fp32 (Winograd GEMMs as 3 x fp16 MFMA on split operands)' if args.dtype == 'f16x3' else 'fp32'}, batch={B} per GPU, {B}x3x{H}x{W} frame{'s' if B > 1 else ''} resident in HBM -> detections on host "
                                         "(trunk + heads + BoxOutput + ROI pool + det sub-net + final NMS)",
                             "batch": B, "regime": args.regime, "mean_rois": round(Rm, 1), "mean_detections": round(float(np.mean(main_stats["D"])), 1),
                             "parallelism": f"image-parallel x{world}", "gather": gather_kind,
                             # harness settings that changed what the timed loop contains over the rounds: the interpreter's cyclic GC is
                             # off inside the timed loops since round 5; the f16x3 second loop is opt-in (--alt) since round 5
                             "gc_disabled": True, "alt_loop": bool(alt),
                             "detect_mode": (args.detect_mode if gather is None and B == 1 else None), "detect_modes": detect_modes,
                             "host_placement": placement, "per_rank": per_rank,
                             # what the collective library reported (ncclCommCount) and the senders' ranks found in the packs of the timed loop
                             "comm_count": comm_count, "ranks_seen": sorted(getattr(gather, "ranks_seen", [])) if gather is not None else None,
                             "handoff": {"tiles": tiles_mode, **dict(zip(("events_answered", "whole_tiles_forced"), net.handoff_state())),
                                         "status_word_changed": _hipapi.wgemm_handoff_event() != handoff_ev0}},
                  "numerics": numerics, "roofline": roofline}
        if args.model == DEFAULT_MODEL:
            result["metric"] = "images/sec mscnn-7s-576 KITTI-car inference"
        parity_ok = None
        if B > 1:
            # the batched forward against the batch-1 path of the SAME library on the same frames (that path is the one pinned to the
            # reference's CPU layers at full size: the batch-1 line's parity_ok, tests/test_gpu_net.py; the batched net against the
            # reference itself: test_whole_net_batch_n_vs_reference at reduced sizes)
            result["metric"] += f" (batch {B})"
            result["roofline"]["flops_note"] += f"; one step = one forward of {B} images"
            result["batch_parity"] = _batch_parity(net, mnet, zoo, synth, args, host_frames[0], kw, B, local_rank)
            parity_ok = result["batch_parity"]["ok"]
            result["cpu_baseline"] = None      # (the reference's CPU path is timed beside the batch-1 line)
        if not args.no_cpu_baseline and world == 1 and B == 1:      # rank 0 at N = 1 only (host work; other ranks would idle)
            table = []
            _set_affinity_all_threads(mask0)    # the reference's CPU path gets every core the process was given, not this rank's slice
            cb = cpu_baseline(args.model, args.regime, max(1, int(round(Rm))), net=net, kw=kw, layer_table=table, dtype=args.dtype,
                              alt_dtype=alt["dtype"] if alt else None, other_regimes=[(r, net_cap if r == "max_rois" else None) for r in (regimes or {}) if r != args.regime])
            net_cap = None
            result["cpu_baseline"] = cb
            if "full_size_parity" in cb:
                parity_ok = cb["full_size_parity"]["ok"]
                if regimes:
                    regimes[args.regime]["parity_ok"] = parity_ok
                    for rg, par in cb.pop("full_size_parity_regimes", {}).items():
                        regimes[rg]["parity_ok"] = par["ok"]
                        regimes[rg]["full_size_parity"] = par
                        parity_ok = parity_ok and par["ok"]      # the line's parity_ok covers every regime it reports
            if alt and "full_size_parity_alt" in cb:
                alt["parity_ok"] = cb["full_size_parity_alt"]["ok"]
                alt["full_size_parity"] = cb.pop("full_size_parity_alt")
            if args.layers and table:
                print("\n# reference CPU path, per layer (caffe time format, tools/caffe.cpp:401-418)", file=sys.stderr)
                for nm, ty, t in table:
                    print(f"{nm:>28s}\tforward: {t * 1e3:.3f} ms.", file=sys.stderr)
        wchecks, wsw = net.numerics_watch_state()
        if numerics is not None:
            # the watch is ON in the timed loop (Net default: every 25th whole forward one band of one Winograd layer is recomputed with
            # the direct kernel behind the frame, no host synchronisation; that frame also writes the layer's bottom / top blobs although it stays chained).  Timed step i is
            # whole forward number warmup + i + 1 of this net: the watch frames of the headline loop are known, their cost is reported
            wp = 25
            wf = [i for i in range(args.steps) if (max(args.warmup, 1) + i + 1) % wp == 0]
            other = [i for i in range(args.steps) if i not in wf]
            sms = np.array(step_s) * 1e3
            numerics["watch"] = {"period": wp, "checks_so_far": wchecks, "switched": wsw, "watch_frames_in_timed_loop": len(wf),
                                 "watch_frame_ms_median": round(float(np.median(sms[wf])), 4) if wf else None,
                                 "other_frame_ms_median": round(float(np.median(sms[other])), 4) if other else None,
                                 "how": "deferred band check (mscnn_net_set_numerics_watch): verdict collected by a later forward, no frame waits"}
        result["parity_ok"] = parity_ok      # null when the reference leg did not run (N > 1 or --no-cpu-baseline)
        # rows of SURVEY 8 whose oracle has no pin on the reference itself: a17 = the MATLAB final stage (run_mscnn_detection.m:75-120,
        # utils/bbNms.m:112-126) -- no MATLAB / Octave in the image; two independently written restatements agree, nothing more.
        # parity_ok says nothing stronger than that about the final stage.
        result["parity_unpinned"] = ["a17"]
        if regimes:
            result["regimes"] = regimes
        if alt:
            alt.setdefault("parity_ok", None)
            result["alt_precision"] = alt
        if robust:
            result["robustness"] = robust
        stage_ms = {}
        for i, nm in enumerate(net.layer_names):
            t = net.layer_types[i]
            key = ("trunk_conv" if t == "Convolution" and not nm.startswith(("LFCN_", "roi_c1")) else
                   "head_conv" if nm.startswith("LFCN_") else nm if nm in ("roi_c1", "fc6") else t)
            stage_ms[key] = stage_ms.get(key, 0.0) + float(lay_ms[i])
        result["stage_ms"] = {k: round(v, 3) for k, v in sorted(stage_ms.items(), key=lambda kv: -kv[1]) if v > 0.0005}
        if parity_ok is False:
            rc = 3
        if result["config"]["handoff"]["status_word_changed"] and result["config"]["handoff"]["events_answered"] == 0:
            # a stream-K hand-off timed out somewhere in this run and no synchronisation point of the Net answered it (the asynchronous
            # multi-GPU pack path synchronises outside the Net): the frames of that step may carry a poisoned tile -- not a valid line
            print("bench.py: a wgemm hand-off time-out was reported and not answered: the line is INVALID", file=sys.stderr)
            rc = 4
    if dist is not None:
        dist.barrier()
        if hasattr(gather, "close"):
            gather.close()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
        if rc:
            print("bench.py: full-size parity against the reference's CPU path FAILED (see cpu_baseline.full_size_parity)", file=sys.stderr)
    # release every device object before interpreter teardown (HIP calls from destructors after the runtime has
    # shut down can hang under rocprofv3)
    del net, frames
    net_cap = None
    import gc
    gc.collect()
    torch.cuda.synchronize()
    sys.exit(rc)


if __name__ == "__main__":
    main()
