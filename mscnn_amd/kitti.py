"""Result files of the reference's KITTI drivers (host side, no device work).

* `write_detections_dlm`  -- the `dlmwrite(['detections/' comp_id '_' obj_names{id} '.txt'], save_detect_boxes)` at the end of
  examples/kitti_car/run_mscnn_detection.m:150-161: one row per detection, `[image_index x y w h score]`, comma separated,
  MATLAB dlmwrite's default precision (%.5g).
* `write_kitti_labels`    -- what examples/kitti_result/writeDetForEval.m:44-86 hands to the devkit's `writeLabels`: one
  `<idx>.txt` per image in the KITTI object label format with the 2-D box and `score * 1000`; the 3-D fields carry the
  devkit's "unknown" defaults.
"""
import os


def _g5(v):
    s = "%.5g" % float(v)
    return s


def write_detections_dlm(path, per_image_dets):
    """per_image_dets: list over images (1-based index in the file) of arrays [D_i, 5] = [x y w h score]."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        for i, dets in enumerate(per_image_dets, start=1):
            for d in dets:
                f.write(",".join([_g5(i)] + [_g5(v) for v in d[:5]]) + "\n")


def read_detections_dlm(path):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([float(v) for v in line.split(",")])
    return rows


def write_kitti_labels(save_dir, image_index, dets_by_type, score_scale=1000.0):
    """dets_by_type: {"Car": [[x y w h score], ...], "Pedestrian": ..., "Cyclist": ...} for ONE image (writeDetForEval.m:46-80:
    x2 = x + w, y2 = y + h, score * 1000).  File name: %06d.txt as the devkit's writeLabels."""
    os.makedirs(save_dir, exist_ok=True)
    with open(os.path.join(save_dir, "%06d.txt" % image_index), "w") as f:
        for typ in ("Car", "Pedestrian", "Cyclist"):
            for d in dets_by_type.get(typ, []):
                x1, y1, x2, y2 = d[0], d[1], d[0] + d[2], d[1] + d[3]
                # type truncated occluded alpha x1 y1 x2 y2 h w l x y z ry score   (devkit readme; -1 / -10 / -1000 = unknown)
                f.write("%s -1 -1 -10 %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10 %.4f\n"
                        % (typ, x1, y1, x2, y2, d[4] * score_scale))


EVAL_BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kitti_eval")


def evaluate(gt_dir, result_dir, list_file):
    """Runs the evaluator (mscnn_amd/host/tools/kitti_eval.cpp, the devkit's examples/kitti_result/eval/evaluate_object.cpp
    restated) on <result_dir>/data/*.txt.  Returns {class: {"precision": [[41] x easy/moderate/hard], "ap11": [3]}}."""
    import subprocess
    if not os.path.exists(EVAL_BIN):
        raise RuntimeError(f"{EVAL_BIN} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    r = subprocess.run([EVAL_BIN, gt_dir, result_dir, list_file], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("kitti_eval failed: " + r.stderr.strip())
    out = {}
    for cls in ("car", "pedestrian", "cyclist"):
        path = os.path.join(result_dir, f"stats_{cls}_detection.txt")
        if not os.path.exists(path):
            continue
        prec = [[float(v) for v in line.split()] for line in open(path) if line.strip()]
        out[cls] = {"precision": prec, "ap11": [100.0 * sum(p[0:41:4]) / 11.0 for p in prec]}
    return out
