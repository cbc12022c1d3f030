#!/usr/bin/env python3
"""The reference's KITTI demo driver (examples/kitti_car/run_mscnn_detection.m, and run_mscnn_detection.m of kitti_ped_cyc /
caltech: same flow, other class lists) on the MI355X path: net from a deploy prototxt + weight file, every image of a directory
resized / BGR / mean-subtracted ON THE DEVICE (:64-69), net.forward timed alone like the reference's tic / toc (:72-73), the
MATLAB post-processing block (:75-120) as ONE call per class (mscnn_net_detect), the result file
detections/<comp_id>_<class>.txt written as dlmwrite does (:150-161), optionally the per-image KITTI label files of
examples/kitti_result/writeDetForEval.m.

  python tools/run_mscnn_detection.py --prototxt mscnn_deploy.prototxt --weights model.caffemodel --images /KITTI/testing/image_2
         [--out detections] [--comp-id kitti_7s_576] [--cls-ids 2] [--names bg,car,van,truck,tram] [--precision f32|f16x3|f16]
         [--labels-dir results/data] [--limit N]
  python tools/run_mscnn_detection.py --model kitti_car/mscnn-7s-576 --synthetic 8      # no dataset / weights at hand: the generated
                                                                                        # deploy net, seeded weights, synthetic frames

Everything here is host glue over calls the test-suite covers one by one (Net.set_image, forward, detect, kitti.write_*)."""
import argparse
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

KITTI_NAMES = {"car": ["bg", "car", "van", "truck", "tram"], "ped_cyc": ["bg", "ped", "cyc"], "caltech": ["bg", "ped"]}


def list_images(image_dir, limit=0):
    """dir([image_dir '*.png']) order (:29): by name; jpg accepted as well for the Caltech / CityPersons frames."""
    files = sorted(f for ext in ("png", "jpg", "jpeg") for f in glob.glob(os.path.join(image_dir, "*." + ext)))
    return files[:limit] if limit > 0 else files


def load_rgb_u8(path):
    """imread: H x W x 3 uint8 RGB (grey images are replicated like MATLAB users do before the [3 2 1] permutation)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))


def frame_id(path, k):
    """KITTI frame number of an image file (000123.png -> 123: what ImageSets/*.txt lists and writeLabels names the file after);
    the 0-based position for files that are not numbered."""
    stem = os.path.splitext(os.path.basename(path))[0] if path else ""
    return int(stem) if stem.isdigit() else k - 1


def names_for(prototxt_text, names_arg):
    """Class names: --names, else from the width of cls_pred (5 = KITTI car nets, 3 = ped / cyc, 2 = pedestrian nets)."""
    if names_arg:
        return names_arg.split(",")
    import re
    m = re.search(r'name:\s*"cls_pred".*?num_output:\s*(\d+)', prototxt_text, re.S)
    n = int(m.group(1)) if m else 5
    return {5: KITTI_NAMES["car"], 3: KITTI_NAMES["ped_cyc"], 2: KITTI_NAMES["caltech"]}.get(n, ["bg"] + [f"class{i}" for i in range(1, n)])


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--prototxt"); ap.add_argument("--weights", help=".caffemodel (new or V1 layout) or .h5 snapshot")
    ap.add_argument("--model", help="a deploy net of mscnn_amd.zoo instead of --prototxt (e.g. kitti_car/mscnn-7s-576)")
    ap.add_argument("--images", help="directory of *.png / *.jpg frames")
    ap.add_argument("--synthetic", type=int, default=0, help="N synthetic KITTI-shaped frames instead of --images")
    ap.add_argument("--out", default="detections"); ap.add_argument("--comp-id", default="mscnn_mi355x")
    ap.add_argument("--cls-ids", default="2", help="1-based class columns, comma separated (the reference's cls_ids)")
    ap.add_argument("--names", default=""); ap.add_argument("--labels-dir", default="")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3", "f16"])
    ap.add_argument("--proposal-thr", type=float, default=-10.0); ap.add_argument("--nms-overlap", type=float, default=0.5)
    ap.add_argument("--limit", type=int, default=0); ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    if not (a.prototxt or a.model) or not (a.images or a.synthetic):
        ap.error("need --prototxt or --model, and --images or --synthetic N")

    import torch
    from mscnn_amd import kitti, net as mnet, synth, zoo
    text = open(a.prototxt).read() if a.prototxt else zoo.prototxt(a.model)
    net = mnet.Net(prototxt_text=text, device=a.device)
    if a.weights:
        net.load_caffemodel(a.weights)              # net.cpp:788-795: ".h5" -> HDF5 snapshot, else binary NetParameter
    else:
        print("no --weights: seeded He-normal weights (synth.load_into) -- detections are meaningless, timings are not", file=sys.stderr)
        synth.load_into(net, "mid")
    if a.precision != "f32":
        net.set_precision(a.precision)
    imgH, imgW = net.blob_shape("data")[2:]
    names = names_for(text, a.names)
    cls_ids = [int(c) for c in a.cls_ids.split(",")]
    files = list_images(a.images, a.limit) if a.images else [None] * a.synthetic
    if not files:
        sys.exit(f"no images in {a.images}")
    per_class = {c: [] for c in cls_ids}
    used = 0.0
    for k, path in enumerate(files, start=1):
        if path is None:                             # a frame of KITTI's size from the seeded generator (uint8 RGB, HWC)
            rng = np.random.default_rng(1701 + k)
            img = np.ascontiguousarray(rng.integers(0, 256, (375, 1242, 3), dtype=np.uint8))
        else:
            img = load_rgb_u8(path)
        orgH, orgW = img.shape[:2]
        ratios = (imgH / float(orgH), imgW / float(orgW))                      # :63
        dev_img = torch.from_numpy(img).cuda(a.device)                       # (kept referenced until the pre-processing has run)
        net.set_image("data", dev_img)                                       # :64-69 on the device
        torch.cuda.synchronize(a.device)
        t0 = time.perf_counter()
        net.forward()
        torch.cuda.synchronize(a.device)
        used += time.perf_counter() - t0                                      # :72-73: forward only
        by_type = {}
        for c in cls_ids:
            dets, _, _ = net.detect(cls_id=c, ratios=ratios, org_hw=(orgH, orgW), proposal_thr=a.proposal_thr, nms_overlap=a.nms_overlap)
            per_class[c].append(dets)
            by_type[{"car": "Car", "ped": "Pedestrian", "cyc": "Cyclist"}.get(names[c - 1], names[c - 1])] = dets
        if a.labels_dir:                                                      # writeDetForEval.m:88-89: the frame's own KITTI id
            kitti.write_kitti_labels(a.labels_dir, frame_id(path, k), by_type)
        if k % 100 == 0 or k == len(files):
            print(f"idx {k}/{len(files)}, avgtime={used / k:.4f}s")           # :147
    for c in cls_ids:
        out = os.path.join(a.out, f"{a.comp_id}_{names[c - 1]}.txt")
        kitti.write_detections_dlm(out, per_class[c])                         # :150-161
        print(f"{out}: {sum(len(d) for d in per_class[c])} detections over {len(files)} images")
    return 0


if __name__ == "__main__":
    sys.exit(main())
