"""mscnn_amd/kitti_eval (host/tools/kitti_eval.cpp) against the reference's own KITTI devkit evaluator: the 3 x 41 precision
samples per class must be IDENTICAL text ("%f" of identical doubles) -- against the committed golden outputs everywhere, and
against the reference binary run side by side where oracle/_ref has it."""
import json
import os
import shutil
import subprocess

import pytest

from mscnn_amd import kitti
from tests import kitti_synth
from tests.golden.make_golden_kitti_eval import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle/_ref/kitti_eval_ref")


@pytest.mark.parametrize("n,seed", CASES)
def test_evaluator_matches_reference(tmp_path, n, seed):
    gt, res, lst = kitti_synth.make(str(tmp_path), n, seed)
    golden = json.load(open(os.path.join(ROOT, "tests/golden/kitti_eval_expected.json")))[f"{n}_{seed}"]
    ref_res = None
    if os.path.exists(REF_BIN):
        ref_res = res + "_ref"
        shutil.copytree(res, ref_res)
        subprocess.run([REF_BIN, gt, ref_res, lst], check=True, capture_output=True)
    out = kitti.evaluate(gt, res, lst)
    assert sorted(out) == sorted(golden)
    for cls in out:
        mine = open(os.path.join(res, f"stats_{cls}_detection.txt")).read()
        assert mine == golden[cls], cls
        if ref_res:
            assert mine == open(os.path.join(ref_res, f"stats_{cls}_detection.txt")).read(), cls
        assert len(out[cls]["precision"]) == 3 and all(len(p) == 41 for p in out[cls]["precision"])
        for p in out[cls]["precision"]:
            assert all(a >= b for a, b in zip(p, p[1:]))            # made monotone from the right


def test_writer_to_evaluator_round_trip(tmp_path):
    """write_kitti_labels -> kitti_eval: perfect detections (= the ground truth boxes, score 1) give precision 1 at every
    recall sample that exists; halving every box (IoU 0.5 < 0.7) gives zero true positives for cars."""
    gt_dir = tmp_path / "label_2"; gt_dir.mkdir()
    boxes = [[100, 100, 120, 80], [400, 150, 200, 90], [700, 120, 90, 60]]           # x y w h, all taller than 40 px
    ids = ["%06d" % i for i in range(20)]          # 60 objects: enough for all 41 recall samples
    (tmp_path / "list.txt").write_text("\n".join(ids) + "\n")
    for idx in ids:
        with open(gt_dir / (idx + ".txt"), "w") as f:
            for x, y, w, h in boxes:
                f.write("Car 0.00 0 -1.57 %.2f %.2f %.2f %.2f -1 -1 -1 -1000 -1000 -1000 -10\n" % (x, y, x + w, y + h))
    for name, shrink, want in (("perfect", 1.0, 1.0), ("half", 0.5, 0.0)):
        res = tmp_path / name
        for i, idx in enumerate(ids):
            kitti.write_kitti_labels(str(res / "data"), i, {"Car": [[x, y, w * shrink, h, 1.0 - 0.1 * k] for k, (x, y, w, h) in enumerate(boxes)]})
        out = kitti.evaluate(str(gt_dir), str(res), str(tmp_path / "list.txt"))
        assert out["car"]["precision"][0][0] == want and out["car"]["precision"][2][0] == want
        assert abs(out["car"]["ap11"][0] - 100.0 * want) < 1e-9
