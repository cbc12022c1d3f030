"""Host-side result writers of the KITTI drivers (run_mscnn_detection.m:150-161, writeDetForEval.m:44-86). CPU only."""
import os

import numpy as np
import pytest

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


def test_detection_driver_host_side(tmp_path):
    """tools/run_mscnn_detection.py (the reference's run_mscnn_detection.m as a script over the product API): the host-side
    pieces -- image listing in dir() order, imread semantics, class names from the deploy net, argument checks."""
    import importlib.util
    import numpy as np
    from PIL import Image
    from mscnn_amd import zoo
    spec = importlib.util.spec_from_file_location("run_mscnn_detection", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                 "tools", "run_mscnn_detection.py"))
    drv = importlib.util.module_from_spec(spec); spec.loader.exec_module(drv)
    rng = np.random.default_rng(0)
    for name in ("000010.png", "000002.png", "000001.jpg"):
        Image.fromarray(rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)).save(tmp_path / name)
    Image.fromarray(rng.integers(0, 256, (12, 20), dtype=np.uint8)).save(tmp_path / "000003.png")      # a grey frame
    files = drv.list_images(str(tmp_path))
    assert [os.path.basename(f) for f in files] == ["000001.jpg", "000002.png", "000003.png", "000010.png"]
    assert len(drv.list_images(str(tmp_path), limit=2)) == 2
    img = drv.load_rgb_u8(files[1])
    assert img.dtype == np.uint8 and img.shape == (12, 20, 3) and img.flags["C_CONTIGUOUS"]
    grey = drv.load_rgb_u8(files[2])
    assert grey.shape == (12, 20, 3) and np.array_equal(grey[..., 0], grey[..., 2])
    assert drv.names_for(zoo.prototxt("kitti_car/mscnn-7s-576"), "") == ["bg", "car", "van", "truck", "tram"]
    assert drv.names_for(zoo.prototxt("kitti_ped_cyc/mscnn-7s-576-2x"), "") == ["bg", "ped", "cyc"]
    assert drv.names_for(zoo.prototxt("caltech/mscnn-7s-480"), "") == ["bg", "ped"]
    assert drv.names_for("", "a,b") == ["a", "b"]
    assert drv.frame_id("/x/000123.png", 5) == 123 and drv.frame_id("/x/frame_a.png", 5) == 4 and drv.frame_id(None, 1) == 0
    with pytest.raises(SystemExit):
        drv.main(["--images", str(tmp_path)])                 # neither --prototxt nor --model
