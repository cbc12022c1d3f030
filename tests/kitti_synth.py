"""Synthetic KITTI-format ground truth + detections for the evaluator tests (no dataset is reachable offline): objects of every
label type the devkit knows, the full range of occlusion / truncation / height, DontCare areas, detections that are jittered
ground truth (true positives at various IoU), duplicates, misses and random false positives, scores with exact ties."""
import os

import numpy as np

TYPES = ["Car", "Van", "Truck", "Pedestrian", "Person_sitting", "Cyclist", "Tram", "Misc", "DontCare"]


def make(root, n_images=40, seed=7):
    rng = np.random.default_rng(seed)
    gt_dir, det_dir = os.path.join(root, "label_2"), os.path.join(root, "result", "data")
    os.makedirs(gt_dir, exist_ok=True); os.makedirs(det_dir, exist_ok=True)
    ids = ["%06d" % i for i in range(n_images)]
    with open(os.path.join(root, "list.txt"), "w") as f:
        f.write("\n".join(ids) + "\n")
    for k, idx in enumerate(ids):
        gts, dets = [], []
        for _ in range(int(rng.integers(0, 9))):
            t = TYPES[int(rng.choice(len(TYPES), p=[.3, .08, .04, .2, .05, .15, .03, .05, .1]))]
            h = float(rng.choice([18, 24, 26, 39, 41, 60, 110, 180]) + rng.uniform(0, 1))
            w = h * (float(rng.uniform(1.2, 2.2)) if t in ("Car", "Van", "Truck", "Tram", "DontCare") else float(rng.uniform(0.35, 0.6)))
            x1 = float(rng.uniform(0, 1242 - w)); y1 = float(rng.uniform(0, 375 - h))
            trunc = float(rng.choice([0, 0.1, 0.15, 0.2, 0.3, 0.4, 0.6])); occ = int(rng.choice([0, 1, 2, 3]))
            if t == "DontCare":
                trunc, occ = -1, -1
            gts.append((t, trunc, occ, float(rng.uniform(-3.14, 3.14)), x1, y1, x1 + w, y1 + h))
            if t in ("Car", "Pedestrian", "Cyclist", "Van", "DontCare") and rng.uniform() < 0.8:
                for _ in range(int(rng.choice([1, 1, 1, 2]))):                       # sometimes a duplicate detection
                    j = rng.normal(0, 0.07 * h, 4)
                    dt = t if t in ("Car", "Pedestrian", "Cyclist") else "Car"
                    dets.append((dt, x1 + j[0], y1 + j[1], x1 + w + j[2], y1 + h + j[3], float(np.round(rng.uniform(0, 1), 2) * 1000)))
        for _ in range(int(rng.integers(0, 4))):                                       # false positives
            dt = ["Car", "Pedestrian", "Cyclist"][int(rng.integers(0, 3))]
            w, h = float(rng.uniform(20, 200)), float(rng.uniform(20, 150))
            x1, y1 = float(rng.uniform(0, 1242 - w)), float(rng.uniform(0, 375 - h))
            dets.append((dt, x1, y1, x1 + w, y1 + h, float(np.round(rng.uniform(0, 0.6), 2) * 1000)))
        with open(os.path.join(gt_dir, idx + ".txt"), "w") as f:
            for t, tr, oc, al, x1, y1, x2, y2 in gts:
                f.write("%s %.2f %d %.2f %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10\n" % (t, tr, oc, al, x1, y1, x2, y2))
        with open(os.path.join(det_dir, idx + ".txt"), "w") as f:
            for t, x1, y1, x2, y2, s in dets:
                f.write("%s -1 -1 -10 %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10 %.4f\n" % (t, x1, y1, x2, y2, s))
    return gt_dir, os.path.join(root, "result"), os.path.join(root, "list.txt")
