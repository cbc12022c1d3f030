#!/usr/bin/env python3
"""MS-CNN hot-path benchmark (BASELINE.json metric: images/sec, mscnn-7s-576 KITTI-car inference, fp32).

A step = one synthetic KITTI-shaped frame (already resident in HBM as the 1x3x576x1920 net input) through the whole path:
VGG-16 trunk + proposal heads (MFMA implicit-GEMM convs) -> BoxOutput (decode / top-2000 / NMS) -> 2 x ROIPooling ->
roi_c1 + fc6 + cls/bbox heads -> final bbox transform + per-class NMS, with the detections delivered to the host.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

Multi-GPU: independent images per GPU (weak scaling, no data-path collective); the per-step detections of all ranks are
gathered with one RCCL all_gather of a fixed-size padded buffer.

Rank 0 prints ONE JSON line.  `roofline` = the conv3_1..conv5_3 block (60 % of the trunk FLOPs; BASELINE.json's north
star names it) timed per layer with HIP events on the stream the net launches on.  `achieved` counts ALGORITHMIC FLOPs (the
reference's direct convolution: 2*MACs); conv3_1..conv5_3 run the Winograd F(3x3,3x3) path, which executes 3.24x fewer
multiplies, so `frac` can exceed 1 -- `executed_tflops` / `executed_frac` give what the MFMA pipe really does.
`cpu_baseline` = oracle/_ref (the reference's own CPU layers) when built, else the CPU oracle, on a bounded sample of the
same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MODEL = "kitti_car/mscnn-7s-576"
H, W = 576, 1920
ORG_HW = (375, 1242)
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
ROOFLINE_LAYERS = ["conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3"]
MAX_DET = 512                           # padded rows per image in the RCCL gather (detections after final NMS)


def _layers_of(n):
    return [(n.layer_names[i], n.layer_types[i], n.layer_bottoms(i), n.layer_tops(i), n.layer_param_text(i))
            for i in range(len(n.layer_names))]


def _full_size_parity(net, x, blobs, kw):
    """The GPU path against the reference's own CPU layers on the SAME full-size frame and weights (the cpu_baseline run):
    end-to-end fp32 error of trunk / head / sub-net blobs and matched final detections.  Checker only."""
    from oracle import pyoracle as orc

    def err(a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64).reshape(a.shape)
        return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())

    net.set_blob("data", x)
    net.forward()
    out = {"max_err_vs_reference_cpu": {b: float(f"{err(net.get_blob(b), blobs[b]):.3g}")
                                        for b in ("conv3_3", "conv4_3", "conv5_3", "conv6_1", "LFCN_1_7x7", "LFCN_3_5x5")},
           "bound": 1e-4}
    Rg, Rr = net.blob_shape("proposals")[0], blobs["proposals"].shape[0]
    dets, ids, _ = net.detect(**kw)
    dref, _ = orc.detections(blobs["bbox_pred"], blobs["cls_pred"], blobs["proposals_score"].reshape(Rr, 6), **kw)
    matched = 0.0
    if len(dets) and len(dref):
        a = np.stack([dets[:, 0], dets[:, 1], dets[:, 0] + dets[:, 2], dets[:, 1] + dets[:, 3]], 1)
        b = np.stack([dref[:, 0], dref[:, 1], dref[:, 0] + dref[:, 2], dref[:, 1] + dref[:, 3]], 1)
        x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
        x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
        inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
        iou = inter / ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])[:, None] + ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None, :] - inter)
        j = iou.argmax(1)
        matched = float(((iou[np.arange(len(a)), j] >= 0.99) & (np.abs(dets[:, 4] - dref[j, 4]) <= 1e-4)).mean())
    out.update({"proposals_gpu": int(Rg), "proposals_reference": int(Rr), "detections_gpu": int(len(dets)),
                "detections_reference": int(len(dref)), "detections_matched_iou99_score1e-4": round(matched, 4)})
    return out


def cpu_baseline(R_gpu, regime, net=None, kw=None):
    """The reference's CPU forward path timed on this box's host cores (rank 0, N=1 only).

    kind "reference": oracle/_ref -- the reference's OWN layer sources (im2col + cblas_sgemm through MKL, serial
    BoxOutput / ROIPooling / pooling loops) on ONE full 576x1920 frame with the same weights; the scope is Net::Forward
    only, like the reference's tic/toc (run_mscnn_detection.m:72).
    kind "port" (when _ref is not built): the C restatement on a bounded quarter-area sample, scaled."""
    from mscnn_amd import net as mnet, synth, zoo
    from oracle import pynet, pyoracle, pyref
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    if pyref.available():
        n = mnet.Net(prototxt_text=zoo.prototxt(MODEL), device=-1)
        layers = _layers_of(n)
        ws = synth.weights(n.layer_names, n.layer_types, [n.param_shapes(i) for i in range(len(n.layer_names))], regime)
        x = synth.frame(H, W, seed=1701)
        t0 = time.perf_counter()
        blobs = pynet.forward(layers, ws, {"data": x}, backend=pyref)
        dt = time.perf_counter() - t0
        R = blobs["proposals"].shape[0]
        res = {"value": round(1.0 / dt, 4), "unit": "images/sec", "cores": cores, "kind": "reference",
               "sample": f"1 full frame (1x3x576x1920, R={R} ROIs) through the reference's own CPU layers "
                         f"(oracle/_ref: im2col + MKL cblas_sgemm, {cores} threads available), Net::Forward scope, {dt:.2f} s",
               "seconds_per_image": round(dt, 2)}
        if net is not None:
            res["full_size_parity"] = _full_size_parity(net, x, blobs, kw)
        return res
    pyoracle.lib()
    h, w = H // 2, W // 2
    n = mnet.Net(prototxt_text=zoo.prototxt(MODEL, height=h, width=w, max_nms_num=32), device=-1)
    layers = _layers_of(n)
    ws = synth.weights(n.layer_names, n.layer_types, [n.param_shapes(i) for i in range(len(n.layer_names))], regime)
    x = synth.frame(h, w)
    cut = [l[0] for l in layers].index("proposals") + 1
    t0 = time.perf_counter()
    blobs = pynet.forward(layers[:cut], ws, {"data": x})
    t_trunk = time.perf_counter() - t0
    t0 = time.perf_counter()
    blobs = pynet.forward(layers[cut:], ws, blobs)
    t_det = time.perf_counter() - t0
    r_sample = blobs["proposals"].shape[0]
    est = 4.0 * t_trunk + t_det * (R_gpu / max(r_sample, 1))
    return {"value": round(1.0 / est, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle restatement (im2col + k-ordered GEMM, OpenMP) on a 288x960 frame for trunk+heads+BoxOutput ({t_trunk:.2f} s) "
                      f"and the detection sub-net on {r_sample} ROIs ({t_det:.2f} s), scaled x4 pixels and x{R_gpu}/{r_sample} ROIs",
            "seconds_per_image_est": round(est, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regime", default="mid", choices=["dense", "mid", "sparse"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", action="store_true", help="print the per-layer table (caffe time format) to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)       # "nccl" is RCCL on ROCm
    else:
        torch.cuda.set_device(local_rank)
    assert torch.cuda.is_available(), "bench.py needs a MI355X"

    from mscnn_amd import net as mnet, synth, zoo
    net = mnet.Net(prototxt_text=zoo.prototxt(MODEL), device=local_rank)
    synth.load_into(net, args.regime)
    # a handful of distinct frames per rank, resident in HBM before the timed region
    frames = [torch.from_numpy(synth.frame(H, W, seed=1701 + 97 * rank + i)).cuda() for i in range(4)]
    kw = dict(cls_id=2, ratios=(H / ORG_HW[0], W / ORG_HW[1]), org_hw=ORG_HW)
    gather = None
    if world > 1:
        from mscnn_amd import dist as mdist
        gather = mdist.DetectionGather(MAX_DET, "cuda")
    stats = {"R": [], "D": []}

    def step(i):
        net.set_blob("data", frames[i % len(frames)])        # D2D: the frame is already in HBM
        net.forward()
        dets, ids, R = net.detect(**kw)                       # final stage on device; detections land on the host
        stats["R"].append(R); stats["D"].append(len(dets))
        if gather is not None:                                # the only collective of the path: detections -> every rank
            gather(dets, ids)
        return dets

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    stats = {"R": [], "D": []}
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: per-layer HIP events on the net's stream, outside the timed region ----
        net.set_layer_timing(True)
        acc = np.zeros(len(net.layer_names)); reps = 5
        for i in range(reps):
            net.set_blob("data", frames[i % len(frames)])
            net.forward()
            acc += np.array(net.layer_ms())
        net.set_layer_timing(False)
        lay_ms = acc / reps
        flops = np.array([net.layer_flops(i) for i in range(len(net.layer_names))])
        idx = [net.layer_names.index(nm) for nm in ROOFLINE_LAYERS]
        blk_flops, blk_ms = float(flops[idx].sum()), float(lay_ms[idx].sum())
        achieved = blk_flops / (blk_ms * 1e-3) / 1e12
        wino = [nm for nm in ROOFLINE_LAYERS if net.layer_kernel(net.layer_names.index(nm)).startswith("winograd")]
        def _cut(nm):       # multiplies executed per algorithmic multiply: F(3x3,3x3) 25/81, F(2x2,3x3) 16/36
            k = net.layer_kernel(net.layer_names.index(nm))
            return 3.24 if k.startswith("winograd_f3x3") else 2.25 if k.startswith("winograd_f2x2") else 1.0
        exec_flops = sum(float(flops[net.layer_names.index(nm)]) / _cut(nm) for nm in ROOFLINE_LAYERS)
        executed = exec_flops / (blk_ms * 1e-3) / 1e12
        conv_idx = [i for i, t in enumerate(net.layer_types) if t == "Convolution"]
        trunk_tf = float(flops[conv_idx].sum()) / (float(lay_ms[conv_idx].sum()) * 1e-3) / 1e12
        if args.layers:
            for i, nm in enumerate(net.layer_names):
                if lay_ms[i] > 0:
                    tf = flops[i] / (lay_ms[i] * 1e-3) / 1e12 if flops[i] else 0
                    print(f"{nm:28s} {net.layer_types[i]:14s} {net.layer_kernel(i):30s} {lay_ms[i]*1e3:9.1f} us {tf:7.1f} TF", file=sys.stderr)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_conv4_2_v5.json")
        if os.path.exists(tpath):      # PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs), per conv4_2 launch
            traffic = json.load(open(tpath)).get("traffic_bytes_per_launch")
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_note": "fabric-side bytes (incl. Infinity-Cache hits) of one conv4_2 layer (Winograd: transforms + GEMM + fix-up), "
                                    "profiles/r01_traffic_conv4_2_v5.json; algorithmic 80 MB -- the Winograd planes V, M account for 393 MB",
                    "kernel": "conv3_1..conv5_3: " + "; ".join(f"{nm}={net.layer_kernel(net.layer_names.index(nm))}" for nm in ROOFLINE_LAYERS)
                              + " (winograd_* = input transform + igemm_kernel<k1x1> batched GEMMs (+ stream-K fix-up) + output transform;"
                                " others = igemm_kernel<128x128,k3x3> direct)",
                    "flops_note": "achieved = algorithmic (direct-convolution) FLOPs / time; Winograd F(3x3,3x3) / F(2x2,3x3) layers execute "
                                  "3.24x / 2.25x fewer multiplies, see executed_tflops",
                    "executed_tflops": round(executed, 2), "executed_frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4),
                    "algorithmic_gflop_per_image": round(blk_flops / 1e9, 2), "avg_ms_per_image": round(blk_ms, 4),
                    "all_conv_tflops": round(trunk_tf, 2)}
        Rm = float(np.mean(stats["R"]))
        result = {"metric": "images/sec mscnn-7s-576 KITTI-car inference", "value": round(value, 3), "unit": "images/sec",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": "mscnn-7s-576 KITTI-car fp32, batch=1 per GPU, 1x3x576x1920 frame resident in HBM -> "
                                         "detections on host (trunk + heads + BoxOutput + ROI pool + det sub-net + final NMS)",
                             "regime": args.regime, "mean_rois": round(Rm, 1), "mean_detections": round(float(np.mean(stats["D"])), 1),
                             "parallelism": f"image-parallel x{world}, RCCL all_gather of detections"},
                  "roofline": roofline}
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only (20 s of host work; other ranks would idle)
            result["cpu_baseline"] = cpu_baseline(max(1, int(round(Rm))), args.regime, net=net, kw=kw)
        stage = {}
        for i, nm in enumerate(net.layer_names):
            t = net.layer_types[i]
            key = ("trunk_conv" if t == "Convolution" and not nm.startswith(("LFCN_", "roi_c1")) else
                   "head_conv" if nm.startswith("LFCN_") else nm if nm in ("roi_c1", "fc6") else t)
            stage[key] = stage.get(key, 0.0) + float(lay_ms[i])
        result["stage_ms"] = {k: round(v, 3) for k, v in sorted(stage.items(), key=lambda kv: -kv[1]) if v > 0.0005}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
    # release every device object before interpreter teardown (HIP calls from destructors after the runtime has
    # shut down can hang under rocprofv3)
    del net, frames
    import gc
    gc.collect()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
