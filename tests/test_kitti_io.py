"""Host-side result writers of the KITTI drivers (run_mscnn_detection.m:150-161, writeDetForEval.m:44-86). CPU only."""
import numpy as np
from mscnn_amd import kitti


def test_dlm_roundtrip(tmp_path):
    dets = [np.array([[10.5, 20.25, 30.0, 40.0, 0.987654321]]), np.zeros((0, 5)),
            np.array([[1234.5678, 0.0, 1.0, 2.0, 1e-7], [5.0, 6.0, 7.0, 8.0, 0.5]])]
    p = tmp_path / "detections" / "x_car.txt"
    kitti.write_detections_dlm(str(p), dets)
    lines = p.read_text().strip().split("\n")
    assert lines[0] == "1,10.5,20.25,30,40,0.98765"            # dlmwrite default precision: %.5g
    assert lines[1] == "3,1234.6,0,1,2,1e-07"
    rows = kitti.read_detections_dlm(str(p))
    assert len(rows) == 3 and rows[2] == [3.0, 5.0, 6.0, 7.0, 8.0, 0.5]


def test_kitti_label_file(tmp_path):
    kitti.write_kitti_labels(str(tmp_path), 42, {"Car": [[10.0, 20.0, 30.0, 40.0, 0.5]], "Cyclist": [[1.0, 2.0, 3.0, 4.0, 0.25]]})
    txt = (tmp_path / "000042.txt").read_text().strip().split("\n")
    assert txt[0] == "Car -1 -1 -10 10.00 20.00 40.00 60.00 -1 -1 -1 -1000 -1000 -1000 -10 500.0000"
    assert txt[1].startswith("Cyclist -1 -1 -10 1.00 2.00 4.00 6.00") and txt[1].endswith("250.0000")
