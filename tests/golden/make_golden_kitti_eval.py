#!/usr/bin/env python3
"""Golden outputs of the REFERENCE's KITTI evaluator (examples/kitti_result/eval/evaluate_object.cpp, compiled by oracle/ref.mk
into oracle/_ref/kitti_eval_ref) on the seeded synthetic label / detection sets of tests/kitti_synth.py.  Run where the
reference checkout exists; the stats files are committed as tests/golden/kitti_eval_expected.json so that the evaluator
test also runs on boxes without it."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import kitti_synth  # noqa: E402

CASES = [(60, 7), (60, 8), (25, 9), (3, 10)]

if __name__ == "__main__":
    ref = os.path.join(ROOT, "oracle/_ref/kitti_eval_ref")
    out = {}
    for n, seed in CASES:
        root = tempfile.mkdtemp()
        gt, res, lst = kitti_synth.make(root, n, seed)
        subprocess.run([ref, gt, res, lst], check=True, capture_output=True)
        out[f"{n}_{seed}"] = {c: open(f"{res}/stats_{c}_detection.txt").read() for c in ("car", "pedestrian", "cyclist")
                              if os.path.exists(f"{res}/stats_{c}_detection.txt")}
        shutil.rmtree(root)
    json.dump(out, open(os.path.join(ROOT, "tests/golden/kitti_eval_expected.json"), "w"), indent=0)
    print({k: sorted(v) for k, v in out.items()})
