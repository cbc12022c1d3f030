"""Golden vectors produced by the reference's own layer sources (tests/golden/make_golden.py -> reference_layers.npz).
CPU: the oracle restatement reproduces them.  GPU (-m gpu): the HIP path reproduces them through the C ABI."""
import ast
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_layers.npz"))
SHAPES = [(18, 60), (18, 60), (9, 30), (9, 30), (5, 15), (5, 15), (3, 8)]
FIELD = [60, 84, 120, 168, 240, 336, 480]; DS = [8, 8, 16, 16, 32, 32, 64]
BOX_CASES = ["dense", "sparse", "empty", "trunc", "norm"]
ROI_CASES = {"org": (7, 7, 0.125, 0.0), "ctx": (7, 7, 0.125, 0.25), "ped": (7, 5, 0.125, 0.25), "cal": (8, 4, 0.125, 0.0)}


def close(a, b, tol=1e-4):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    assert (np.abs(a - b) / np.maximum(1, np.abs(b))).max() <= tol


# ---------------------------------------------------------------- CPU: oracle vs reference outputs
@pytest.mark.parametrize("case", BOX_CASES)
def test_oracle_boxoutput(orc, case):
    kw = ast.literal_eval(str(G[f"boxout_{case}_kw"]))
    rois, props, _, _ = orc.boxoutput([G[f"boxout_{case}_head{j}"] for j in range(7)], FIELD, FIELD, DS, **kw)
    assert np.array_equal(rois, G[f"boxout_{case}_rois"]) and np.array_equal(props, G[f"boxout_{case}_props"])


def test_oracle_roipool_decode_layers(orc):
    for tag, (ph, pw, sc, pad) in ROI_CASES.items():
        assert np.array_equal(orc.roipool(G["roipool_feat"], G["roipool_rois"], ph, pw, sc, pad), G[f"roipool_{tag}"])
    assert np.array_equal(orc.decode_bbox(G["decode_bbox"], G["decode_prior"], (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2)), G["decode_out"])
    close(orc.conv2d(G["conv_x"], G["conv_w"], G["conv_b"], (1, 1)), G["conv_y"])
    close(orc.conv2d(G["conv_x"], G["head_w"], G["conv_b"][:9], (3, 3)), G["head_y"])
    assert np.array_equal(orc.pool2d(G["conv_x"][:, :, :11, :19]), G["pool_y"])
    close(orc.inner_product(G["ip_x"], G["ip_w"], G["conv_b"][:10]), G["ip_y"])
    assert np.array_equal(orc.deconv2d(G["conv_x"], orc.bilinear_filler((16, 1, 4, 4)), None, (1, 1), (2, 2), 16), G["deconv_y"])


GB = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_layers_b.npz"))
ALIGN_CASES = {"a": (7, 7, 0.125, 0.0), "b": (7, 7, 0.125, 0.25), "c": (4, 6, 0.25, 0.5)}


def test_oracle_cascade_layers(orc):
    """ROIAlign, Eltwise, AVE pooling, Softmax of the cascade / WiderFace deploys against the reference's own outputs
    (tests/golden/make_golden_b.py)."""
    for tag, (ph, pw, sc, pad) in ALIGN_CASES.items():
        assert np.array_equal(orc.roialign(GB["roialign_feat"], GB["roialign_rois"], ph, pw, sc, pad), GB[f"roialign_{tag}"])
    xs = [GB["elt_a"], GB["elt_b"], GB["elt_c"]]
    assert np.array_equal(orc.eltwise(xs, "SUM"), GB["elt_sum"])
    close(orc.eltwise(xs[:2], "SUM", [0.5, 0.5]), GB["elt_avg"], 1e-6)     # reference: MKL saxpy (fma) -> 1 ulp
    assert np.array_equal(orc.eltwise(xs, "PROD"), GB["elt_prod"])
    assert np.array_equal(orc.eltwise(xs, "MAX"), GB["elt_max"])
    close(orc.pool2d(GB["ave_x"], (2, 2), (0, 0), (1, 1), "AVE"), GB["ave_y"], 1e-6)
    close(orc.softmax(GB["softmax_x"]), GB["softmax_y"], 1e-6)


# ---------------------------------------------------------------- GPU: HIP path vs reference outputs
@pytest.fixture(scope="module")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    from mscnn_amd import hipapi
    hipapi.lib()
    return hipapi


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("case", BOX_CASES)
def test_hip_boxoutput(hip, case):
    kw = ast.literal_eval(str(G[f"boxout_{case}_kw"]))
    d = hip.make_boxoutput_desc(SHAPES, 1, 9, FIELD, FIELD, DS, **kw)
    rois, props, aids, nreal = hip.BoxOutput(d).forward([dev(G[f"boxout_{case}_head{j}"]) for j in range(7)])
    assert np.array_equal(rois.cpu().numpy(), G[f"boxout_{case}_rois"])          # selection, order and boxes: bit-exact
    assert np.array_equal(props.cpu().numpy(), G[f"boxout_{case}_props"])


@pytest.mark.gpu
def test_hip_roipool_decode_layers(hip):
    for tag, (ph, pw, sc, pad) in ROI_CASES.items():
        y = hip.roipool(dev(G["roipool_feat"]), dev(G["roipool_rois"]), ph, pw, sc, pad)
        assert np.array_equal(y.cpu().numpy(), G[f"roipool_{tag}"])
    y = hip.decode_bbox(dev(G["decode_bbox"]), dev(G["decode_prior"]), (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2))
    assert np.array_equal(y.cpu().numpy(), G["decode_out"])
    close(hip.conv2d(dev(G["conv_x"]), dev(G["conv_w"]), dev(G["conv_b"]), (1, 1)).cpu().numpy(), G["conv_y"])
    close(hip.conv2d(dev(G["conv_x"]), dev(G["head_w"]), dev(G["conv_b"][:9]), (3, 3)).cpu().numpy(), G["head_y"])
    assert np.array_equal(hip.pool2d(dev(G["conv_x"][:, :, :11, :19])).cpu().numpy(), G["pool_y"])
    close(hip.inner_product(dev(G["ip_x"]), dev(G["ip_w"]), dev(G["conv_b"][:10])).cpu().numpy(), G["ip_y"])
    w = np.tile(np.outer([0.25, 0.75, 0.75, 0.25], [0.25, 0.75, 0.75, 0.25]).astype(np.float32), (16, 1, 1, 1))
    close(hip.deconv_depthwise(dev(G["conv_x"]), dev(w), None, (1, 1), (2, 2)).cpu().numpy(), G["deconv_y"], 1e-6)


@pytest.mark.gpu
def test_hip_cascade_layers(hip):
    for tag, (ph, pw, sc, pad) in ALIGN_CASES.items():
        y = hip.roialign(dev(GB["roialign_feat"]), dev(GB["roialign_rois"]), ph, pw, sc, pad)
        assert np.array_equal(y.cpu().numpy(), GB[f"roialign_{tag}"])
    xs = [dev(GB["elt_a"]), dev(GB["elt_b"]), dev(GB["elt_c"])]
    assert np.array_equal(hip.eltwise(xs, "SUM").cpu().numpy(), GB["elt_sum"])
    close(hip.eltwise(xs[:2], "SUM", [0.5, 0.5]).cpu().numpy(), GB["elt_avg"], 1e-6)
    assert np.array_equal(hip.eltwise(xs, "PROD").cpu().numpy(), GB["elt_prod"])
    assert np.array_equal(hip.eltwise(xs, "MAX").cpu().numpy(), GB["elt_max"])
    close(hip.pool2d(dev(GB["ave_x"]), (2, 2), (0, 0), (1, 1), "AVE").cpu().numpy(), GB["ave_y"], 1e-6)
    close(hip.softmax(dev(GB["softmax_x"])).cpu().numpy(), GB["softmax_y"], 1e-6)
