"""Boundary tests that need no GPU: the prototxt text-format parser, the generated model zoo against the reference's
deploy files, and Net graph construction (legacy input upgrade, Split insertion + naming, in-place tops, alphabetical
outputs, blob shapes) -- reference behaviour: src/caffe/net.cpp:49-284, util/insert_splits.cpp, util/upgrade_proto.cpp."""
import glob
import os

import pytest

from mscnn_amd import net as mnet
from mscnn_amd import zoo

REF = "/root/reference/examples"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")


def _device():
    try:
        import torch
        return 0 if torch.cuda.is_available() else -1
    except Exception:
        return -1


class Net(mnet.Net):
    def __init__(self, path=None, prototxt_text=None, fusion=None):
        super().__init__(path, prototxt_text=prototxt_text, device=_device(), fusion=fusion)


def graph(n):
    return [(n.layer_names[i], n.layer_types[i], n.layer_bottoms(i), n.layer_tops(i), n.param_shapes(i)) for i in range(len(n.layer_names))]


def test_parser_features():
    txt = '''
    name: "t"   # trailing comment
    input: "data" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8
    layer { name: "c" type: "Convolution" bottom: "data" top: "c"
            convolution_param: { num_output: 4 kernel_size: 3 pad: 1 bias_term: false weight_filler: { type: "bilinear" } } }
    layer { name: 'r' type: "ReLU" bottom: "c" top: "c" relu_param { negative_slope: -0.5e-1 } }
    layer { name: "p" type: "Pooling" bottom: "c" top: "p" pooling_param { pool: AVE kernel_size: 2 stride: 2 } }
    '''
    n = Net(prototxt_text=txt)
    assert n.layer_names == ["input", "c", "r", "p"]
    assert n.layer_types == ["Input", "Convolution", "ReLU", "Pooling"]
    assert n.blob_shape("data") == (1, 3, 8, 8) and n.blob_shape("c") == (1, 4, 8, 8) and n.blob_shape("p") == (1, 4, 4, 4)
    assert n.param_shapes(1) == [(4, 3, 3, 3)]                    # bias_term: false
    assert n.outputs == ["p"]
    assert not n.fused_away(2)                                     # negative slope != 0 is not fused


def test_parse_errors_are_reported():
    with pytest.raises(mnet.NetError, match="parse error"):
        Net(prototxt_text='layer { name: "x" type: ')
    with pytest.raises(mnet.NetError, match="Unknown layer type"):
        Net(prototxt_text='input: "d" input_dim: 1 input_dim: 1 input_dim: 2 input_dim: 2 layer { name: "x" type: "Nope" bottom: "d" top: "x" }')
    with pytest.raises(mnet.NetError, match="Unknown bottom blob"):
        Net(prototxt_text='layer { name: "x" type: "ReLU" bottom: "missing" top: "x" }')


def test_7s576_graph_splits_and_shapes():
    n = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576"))
    assert n.blob_shape("data") == (1, 3, 576, 1920)
    # conv4_3 has 4 consumers (loss1_conv1, pool4, roi_pool_org, roi_pool_ctx): split named after the LAST producer (relu4_3)
    i = n.layer_names.index("conv4_3_relu4_3_0_split")
    assert n.layer_types[i] == "Split" and n.layer_bottoms(i) == ["conv4_3"]
    assert n.layer_tops(i) == [f"conv4_3_relu4_3_0_split_{k}" for k in range(4)]
    assert n.layer_bottoms(n.layer_names.index("loss1_conv1")) == ["conv4_3_relu4_3_0_split_0"]
    assert n.layer_bottoms(n.layer_names.index("pool4")) == ["conv4_3_relu4_3_0_split_1"]
    assert n.layer_bottoms(n.layer_names.index("roi_pool_org")) == ["conv4_3_relu4_3_0_split_2", "proposals_proposals_0_split_0"]
    assert n.blob_shape("conv4_3") == (1, 512, 72, 240)
    assert n.blob_shape("conv5_3") == (1, 512, 36, 120) and n.blob_shape("conv6_1") == (1, 512, 18, 60)
    assert n.blob_shape("pool6") == (1, 512, 9, 30)
    assert n.blob_shape("LFCN_1_7x7") == (1, 9, 72, 240) and n.blob_shape("LFCN_4_5x5") == (1, 9, 9, 30)
    assert n.blob_shape("proposals") == (1, 5, 1, 1) and n.blob_shape("proposals_score") == (1, 6, 1, 1)   # dummy reshape
    assert n.blob_shape("roi_pool") == (1, 1024, 7, 7) and n.blob_shape("roi_c1") == (1, 512, 5, 5)
    assert n.blob_shape("fc6") == (1, 4096) and n.blob_shape("bbox_pred") == (1, 20)
    assert n.outputs == ["bbox_pred", "cls_pred", "proposals_score"]          # std::set order, net.cpp:267-274
    assert n.param_shapes(n.layer_names.index("fc6")) == [(4096, 12800), (4096,)]
    assert n.param_shapes(n.layer_names.index("LFCN_2_7x7")) == [(9, 512, 7, 7), (9,)]
    # every in-place ReLU after a conv / fc is folded into its producer
    for name in ("relu1_1", "relu4_3", "loss_relu1", "roi_c1_relu", "relu6"):
        assert n.fused_away(n.layer_names.index(name))
    assert n.fused_away(n.layer_names.index("roi_pool"))                     # ROIPooling x2 write the Concat top directly
    assert sum(t == "Convolution" for t in n.layer_types) == 23 and sum(t == "Pooling" for t in n.layer_types) == 6


def _expected_splits(layers):
    """The reference's naming rule (util/insert_splits.cpp:14-124), restated independently of the host runtime's implementation: for
    every top VERSION (an in-place layer re-issues its bottom's name) count the bottoms that read it; a version with more than one
    reader gets a Split layer "<blob>_<layer>_<top index>_split" behind its producer and reader k reads "..._split_<k>", in net order."""
    newest, readers, reads = {}, {}, []
    for li, (name, typ, bottoms, tops) in enumerate(layers):
        mine = []
        for b in bottoms:
            mine.append(newest[b])
            readers[newest[b]] = readers.get(newest[b], 0) + 1
        reads.append(mine)
        for ti, t in enumerate(tops):
            newest[t] = (li, ti)
    out, handed = [], {}
    for li, (name, typ, bottoms, tops) in enumerate(layers):
        nb = []
        for b, v in zip(bottoms, reads[li]):
            if readers[v] > 1:
                k = handed.get(v, 0); handed[v] = k + 1
                pl, pt = v
                nb.append(f"{layers[pl][3][pt]}_{layers[pl][0]}_{pt}_split_{k}")
            else:
                nb.append(b)
        out.append((name, typ, nb, list(tops)))
        for ti, t in enumerate(tops):
            n = readers.get((li, ti), 0)
            if n > 1:
                stem = f"{t}_{name}_{ti}_split"
                out.append((stem, "Split", [t], [f"{stem}_{k}" for k in range(n)]))
    return out


def test_split_insertion_on_in_place_layers_multiple_tops_and_fan_out():
    """Net construction's Split insertion (host/src/net.cpp InsertSplits, rewritten in round 6) on the cases the deploy files mix: a top
    read three times, an in-place ReLU between a producer and two readers (the readers must split the ReLU's version of the blob, not
    the convolution's), a layer with two tops each read twice, and a blob read once (no Split).  Names, order and wiring against the
    rule restated above."""
    txt = '''
    name: "s" input: "data" input_dim: 1 input_dim: 3 input_dim: 16 input_dim: 16
    layer { name: "a" type: "Convolution" bottom: "data" top: "a" convolution_param { num_output: 4 kernel_size: 3 pad: 1 } }
    layer { name: "a_relu" type: "ReLU" bottom: "a" top: "a" }
    layer { name: "b" type: "Convolution" bottom: "a" top: "b" convolution_param { num_output: 4 kernel_size: 3 pad: 1 } }
    layer { name: "c" type: "Convolution" bottom: "a" top: "c" convolution_param { num_output: 4 kernel_size: 3 pad: 1 } }
    layer { name: "d" type: "Convolution" bottom: "a" top: "d" convolution_param { num_output: 4 kernel_size: 3 pad: 1 } }
    layer { name: "sum" type: "Eltwise" bottom: "b" bottom: "c" bottom: "d" top: "sum" eltwise_param { operation: SUM } }
    layer { name: "cat" type: "Concat" bottom: "sum" bottom: "sum" top: "cat" }
    layer { name: "e" type: "Convolution" bottom: "cat" top: "e" convolution_param { num_output: 2 kernel_size: 1 } }
    '''
    n = Net(prototxt_text=txt, fusion=False)
    got = [(n.layer_names[i], n.layer_types[i], n.layer_bottoms(i), n.layer_tops(i)) for i in range(len(n.layer_names))]
    src = [("input", "Input", [], ["data"]), ("a", "Convolution", ["data"], ["a"]), ("a_relu", "ReLU", ["a"], ["a"]),
           ("b", "Convolution", ["a"], ["b"]), ("c", "Convolution", ["a"], ["c"]), ("d", "Convolution", ["a"], ["d"]),
           ("sum", "Eltwise", ["b", "c", "d"], ["sum"]), ("cat", "Concat", ["sum", "sum"], ["cat"]), ("e", "Convolution", ["cat"], ["e"])]
    want = _expected_splits(src)
    assert got == want, (got, want)
    names = [g[0] for g in got]
    assert "a_a_relu_0_split" in names and "a_a_0_split" not in names          # the in-place ReLU's version is the one that fans out
    assert got[names.index("a_a_relu_0_split")][3] == ["a_a_relu_0_split_0", "a_a_relu_0_split_1", "a_a_relu_0_split_2"]
    assert got[names.index("cat")][2] == ["sum_sum_0_split_0", "sum_sum_0_split_1"]                       # one layer reading a blob twice
    # a layer with two tops, each read twice: the 7s-576 deploy's BoxOutput ("proposals" feeds both ROI poolings; the score top is an output)
    d = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576"), fusion=False)
    dn = list(d.layer_names)
    assert dn[dn.index("proposals") + 1] == "proposals_proposals_0_split"
    full = [(d.layer_names[i], d.layer_types[i], d.layer_bottoms(i), d.layer_tops(i)) for i in range(len(dn))]
    unsplit = [l for l in full if l[1] != "Split"]
    # undo the split renaming to get the source graph back, then demand the rule reproduces the whole net
    stems = {l[0]: l[2][0] for l in full if l[1] == "Split"}
    def unrename(b):
        for stem, blob in stems.items():
            if b.startswith(stem + "_"):
                return blob
        return b
    source = [(nm, ty, [unrename(b) for b in bo], to) for (nm, ty, bo, to) in unsplit]
    assert _expected_splits(source) == full


def test_convolution_chains_are_wired_at_construction(monkeypatch):
    """Round 4: the Net registers (a) every convolution whose top has exactly one running reader, a 3x3 convolution right behind it
    (the blob between them may stay unwritten while both run chained), (b) every convolution whose top is read by its fused 2x2
    pooling alone.  A convolution feeding a Split (conv4_3, conv5_3), a head or a net output is never in the list; without the
    Net-level fusion (the ReLU layers run) there is none."""
    n = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576"))
    pairs = n.chain_pairs()
    assert set(p for p in pairs if p[1]) == {("conv1_1", "conv1_2"), ("conv2_1", "conv2_2"), ("conv3_1", "conv3_2"), ("conv3_2", "conv3_3"),
                                            ("conv4_1", "conv4_2"), ("conv4_2", "conv4_3"), ("conv5_1", "conv5_2"), ("conv5_2", "conv5_3")}
    assert set(p[0] for p in pairs if p[1] is None) == {"conv1_2", "conv2_2", "conv3_3"}
    u = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576"), fusion=False)
    assert u.chain_pairs() == []
    c = Net(prototxt_text=zoo.prototxt("caltech/mscnn-7s-480"))
    assert ("conv3_1", "conv3_2") in c.chain_pairs() and ("conv1_2", None) in c.chain_pairs()


def test_other_configs_shapes():
    n = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-8s-768-trainval"))
    assert n.blob_shape("data") == (1, 3, 768, 2560) and n.blob_shape("LFCN_4_7x7") == (1, 9, 12, 40)
    assert sum(t == "Convolution" for t in n.layer_types) == 24
    n = Net(prototxt_text=zoo.prototxt("kitti_ped_cyc/mscnn-7s-576-2x"))
    assert n.blob_shape("conv4_3_2x") == (1, 512, 144, 480) and n.blob_shape("roi_pool") == (1, 1024, 7, 5)
    assert n.blob_shape("roi_c1") == (1, 512, 5, 3) and n.blob_shape("LFCN_1_5x7") == (1, 7, 72, 240)
    assert n.param_shapes(n.layer_names.index("LFCN_1_3x5")) == [(7, 512, 5, 3), (7,)]     # kernel_h 5, kernel_w 3
    assert n.param_shapes(n.layer_names.index("conv4_3_2x")) == [(512, 1, 4, 4)]
    w = n.get_param("conv4_3_2x", 0)                                                         # bilinear filler
    assert abs(w[0, 0, 0, 0] - 0.0625) < 1e-7 and abs(w[5, 0, 1, 2] - 0.5625) < 1e-7
    n = Net(prototxt_text=zoo.prototxt("caltech/mscnn-7s-480"))
    assert n.blob_shape("pool6") == (1, 512, 8, 10)                                          # ceil-mode pooling: 15 -> 8
    assert n.blob_shape("roi_c1") == (1, 512, 8, 4) and n.blob_shape("bbox_pred") == (1, 8)


def _semantic_params(n, i):
    """The layer's parameter message with everything that does not change a TEST-phase forward removed (fillers, lr / decay
    multipliers, propagate_down, phase) and numbers normalised (0.25 == 0.250, "IOU" == IOU)."""
    from oracle import pynet

    def norm(v):
        if isinstance(v, dict):
            return {k: [norm(x) for x in vs] for k, vs in sorted(v.items()) if k not in ("weight_filler", "bias_filler")}
        try:
            return float(v)
        except ValueError:
            return v.strip('"')
    d = pynet.parse_param_text(n.layer_param_text(i))
    for k in ("name", "type", "bottom", "top", "param", "propagate_down", "phase"):
        d.pop(k, None)
    return norm(d)


@needs_ref
@pytest.mark.parametrize("model", sorted(zoo.MODELS))
def test_generated_net_equals_reference_file(model):
    ref = Net(os.path.join(REF, zoo.MODELS[model][1]))
    gen = Net(prototxt_text=zoo.prototxt(model))
    assert graph(gen) == graph(ref)
    assert gen.blob_names == ref.blob_names and gen.outputs == ref.outputs
    for b in gen.blob_names:
        assert gen.blob_shape(b) == ref.blob_shape(b)
    for i, name in enumerate(gen.layer_names):          # every numeric parameter: fields, strides, thresholds, stds, coeffs ...
        assert _semantic_params(gen, i) == _semantic_params(ref, i), name


@pytest.mark.parametrize("model", sorted(zoo.MODELS))
def test_generated_net_matches_the_committed_fingerprint_of_the_shipped_file(model):
    """The same statement on a box WITHOUT the reference checkout (the GPU box): the canonical description of the generated net
    (graph, blob shapes, outputs, every TEST-phase parameter; tests/deploy_fingerprint.py) hashes to the value
    tests/golden/make_deploy_fingerprints.py computed from the reference's own mscnn_deploy.prototxt."""
    import json
    from deploy_fingerprint import fingerprint
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deploy_fingerprints.json")))
    assert model in want, "fixture out of date: run tests/golden/make_deploy_fingerprints.py"
    gen = Net(prototxt_text=zoo.prototxt(model))
    assert len(gen.layer_names) == want[model]["layers"]
    assert fingerprint(gen) == want[model]["sha256"], want[model]["file"]


@needs_ref
def test_committed_deploy_fingerprints_are_those_of_the_reference_files():
    import json
    from deploy_fingerprint import fingerprint
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deploy_fingerprints.json")))
    assert sorted(want) == sorted(zoo.MODELS)
    for model in sorted(zoo.MODELS):
        assert fingerprint(Net(os.path.join(REF, zoo.MODELS[model][1]))) == want[model]["sha256"], model


@needs_ref
def test_all_reference_deploys_build():
    """Every shipped deploy file (23: KITTI car / ped-cyc, Caltech, CityPersons, cascade-*, WiderFace ROIAlign) goes through
    the text parser AND builds its graph: all 15 layer types they use are registered."""
    files = sorted(glob.glob(os.path.join(REF, "*/*/mscnn_deploy.prototxt")))
    assert len(files) == 23
    types = set()
    for f in files:
        n = Net(f)
        assert n.outputs, f
        types.update(n.layer_types)
    assert types == {"Input", "Split", "Convolution", "Deconvolution", "Pooling", "ReLU", "InnerProduct", "Concat", "Dropout",
                     "Softmax", "Eltwise", "ROIPooling", "ROIAlign", "BoxOutput", "DecodeBBox"}


def test_boxoutput_default_max_nms_num_builds():
    """box_output_param.max_nms_num 0 is the caffe.proto default ("no cap", box_output_layer.cpp:172-173).  The full-size nets
    have 45,630 / 81,600 anchors: more than the LDS-resident sort holds, served by the tiled path -- the layer must set up."""
    for model, anchors in (("kitti_car/mscnn-7s-576", 45630), ("kitti_car/mscnn-8s-768-trainval", 81600)):
        n = Net(prototxt_text=zoo.prototxt(model, max_nms_num=0))
        i = n.layer_names.index("proposals")
        assert "max_nms_num: 0" in n.layer_param_text(i).replace("  ", " ")


# ---- .caffemodel (binary NetParameter) reader: Net::CopyTrainedLayersFrom, net.cpp:750-803 / blob.cpp:448-482 ----
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):          # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _blobproto(arr, legacy=False):
    import numpy as np
    a = np.ascontiguousarray(arr, np.float32)
    body = b""
    if legacy:                    # num/channels/height/width = fields 1..4 (varint)
        # legacy blobs index from the END of the shape: a bias is 1 x 1 x 1 x N (blob.cpp:392-406)
        dims = [1] * (4 - a.ndim) + list(a.shape) if a.ndim <= 4 else list(a.shape)
        for f, d in enumerate(dims[:4], start=1):
            body += _varint((f << 3) | 0) + _varint(d)
    else:                         # shape = 7 { dim = 1 packed }
        body += _ld(7, _ld(1, b"".join(_varint(d) for d in a.shape)))
    body += _ld(5, a.tobytes())   # data = 5, packed floats
    return body


def _caffemodel(layers):
    """layers: [(name, type, [arrays])] -> bytes of NetParameter{ name=1, layer=100{ name=1, type=2, blobs=7 } }"""
    out = _ld(1, b"synthetic")
    for i, (name, typ, blobs) in enumerate(layers):
        lp = _ld(1, name.encode()) + _ld(2, typ.encode())
        for b in blobs:
            lp += _ld(7, _blobproto(b, legacy=(i % 2 == 1)))
        lp += _varint((10 << 3) | 0) + _varint(1)      # an unrelated varint field (phase = 10) must be skipped
        out += _ld(100, lp)
    return out


def test_caffemodel_reader(tmp_path):
    import numpy as np
    n = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576", height=64, width=128))
    rng = np.random.default_rng(0)
    want = {}
    layers = []
    for name in ("conv1_1", "conv4_3", "LFCN_2_7x7", "fc6", "bbox_pred"):
        shapes = n.param_shapes(n.layer_names.index(name))
        arrs = [rng.standard_normal(s).astype(np.float32) for s in shapes]
        want[name] = arrs
        layers.append((name, "Convolution", arrs))
    layers.append(("layer_not_in_net", "ReLU", []))           # ignored (net.cpp:760-764)
    path = tmp_path / "w.caffemodel"
    path.write_bytes(_caffemodel(layers))
    n.load_caffemodel(path)
    for name, arrs in want.items():
        for p, a in enumerate(arrs):
            assert np.array_equal(n.get_param(name, p), a), (name, p)
    assert not np.any(n.get_param("conv2_1", 0))                # untouched layers keep their constant-0 filler
    # shape mismatch is fatal, like the reference's CHECK in Blob::FromProto
    bad = tmp_path / "bad.caffemodel"
    bad.write_bytes(_caffemodel([("conv1_1", "Convolution", [np.zeros((3, 3), np.float32), np.zeros(64, np.float32)])]))
    with pytest.raises(mnet.NetError, match="shape mismatch"):
        n.load_caffemodel(bad)


def test_caffemodel_written_by_protobuf_python(tmp_path):
    """The reader against an encoder this repository did not write: the protobuf runtime serialises a NetParameter declared
    with caffe.proto's field numbers (tests/caffemodel_pb.py): BlobShape-style, legacy 4-D and double_data blobs, an
    unpacked repeated float (loss_weight) and enum / string fields to skip."""
    import numpy as np
    from tests import caffemodel_pb
    n = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576", height=64, width=128))
    rng = np.random.default_rng(3)
    want, layers = {}, []
    for name, mode in (("conv1_1", "shape"), ("conv3_2", "legacy"), ("LFCN_1_7x7", "double"), ("fc6", "legacy"), ("cls_pred", "shape")):
        shapes = n.param_shapes(n.layer_names.index(name))
        arrs = [rng.standard_normal(sh).astype(np.float32) for sh in shapes]
        want[name] = arrs
        layers.append((name, "Convolution", [(a, mode) for a in arrs]))
    layers.append(("not_in_this_net", "ReLU", []))
    path = tmp_path / "pb.caffemodel"
    data = caffemodel_pb.serialize(layers)
    path.write_bytes(data)
    back = caffemodel_pb.classes()["NetParameter"](); back.ParseFromString(data)          # the fixture itself round-trips
    assert [l.name for l in back.layer][:2] == ["conv1_1", "conv3_2"] and len(back.layer[1].blobs[1].data) == 256
    n.load_caffemodel(path)
    for name, arrs in want.items():
        for p_, a in enumerate(arrs):
            assert np.array_equal(n.get_param(name, p_), a), (name, p_)        # float -> double -> float is exact
    # truncated file: a CHECK failure, never a read past the buffer
    cut = tmp_path / "cut.caffemodel"
    cut.write_bytes(data[:len(data) // 2])
    with pytest.raises(mnet.NetError, match="truncated caffemodel"):
        n.load_caffemodel(cut)
    # wrong shape with the right element count (3x3x64x3 instead of 64x3x3x3): the reference's ShapeEquals refuses it
    w = want["conv1_1"][0]
    bad = tmp_path / "bad.caffemodel"
    bad.write_bytes(caffemodel_pb.serialize([("conv1_1", "Convolution", [(w.reshape(3, 3, 64, 3), "shape"), (want["conv1_1"][1], "shape")])]))
    with pytest.raises(mnet.NetError, match="shape mismatch"):
        n.load_caffemodel(bad)
    # a ".h5" name is read as an HDF5 snapshot (net.cpp:788-795): these bytes are not one
    (tmp_path / "weights.caffemodel.h5").write_bytes(data)
    with pytest.raises(mnet.NetError, match="not an HDF5 file"):
        n.load_caffemodel(tmp_path / "weights.caffemodel.h5")


def test_caffemodel_v1_layers(tmp_path):
    """The deprecated V1 form (NetParameter.layers = 2, V1LayerParameter name = 4 / blobs = 6): the reference upgrades it on load
    (upgrade_proto.cpp UpgradeV1Net) and copies by name; legacy 4-D blobs, as every V1-era file has them."""
    import numpy as np
    from tests import caffemodel_pb
    n = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576", height=64, width=128))
    rng = np.random.default_rng(4)
    want, layers = {}, []
    for name in ("conv1_1", "conv2_2", "cls_pred"):
        arrs = [rng.standard_normal(sh).astype(np.float32) for sh in n.param_shapes(n.layer_names.index(name))]
        want[name] = arrs
        layers.append((name, "Convolution", [(a, "legacy") for a in arrs]))
    path = tmp_path / "v1.caffemodel"
    path.write_bytes(caffemodel_pb.serialize(layers + [("fc8_not_here", "InnerProduct", [])], v1=True))
    n.load_caffemodel(path)
    for name, arrs in want.items():
        for j, a in enumerate(arrs):
            assert np.array_equal(n.get_param(name, j), a), (name, j)


# ---- HDF5 weight snapshots (".h5"): Net::CopyTrainedLayersFromHDF5, net.cpp:788-795, 806-848 / util/hdf5.cpp:9-74 ----
# The product parses the file itself (mscnn_amd/host/src/hdf5_lite.cpp: superblock v0, symbol-table groups, v1 object headers,
# contiguous datasets); the fixtures are written by the real libhdf5 (tests/golden/make_hdf5_weights.py).
def test_hdf5_weights_committed_fixture(tmp_path):
    import numpy as np
    from tests.golden import make_hdf5_weights as mk
    n = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576", height=64, width=128))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "weights_small.h5")
    n.load_caffemodel(path)                                    # the reference dispatches on the ".h5" suffix too
    want = dict(mk.small_fixture()[:3])
    for name, arrs in want.items():
        for j, a in enumerate(arrs):                           # double / int datasets arrive converted (H5LTread_dataset_float)
            assert np.array_equal(n.get_param(name, j), a.astype(np.float32)), (name, j)
    assert not np.any(n.get_param("conv2_1", 0))               # layers absent from the file keep their values
    # a truncated copy and a file that is not HDF5 at all: errors, never reads past the buffer
    raw = open(path, "rb").read()
    for cut in (len(raw) // 2, 5000, 700, 90):
        bad = tmp_path / f"cut{cut}.h5"
        bad.write_bytes(raw[:cut])
        with pytest.raises(mnet.NetError, match="HDF5"):
            n.load_caffemodel(bad)
    junk = tmp_path / "junk.h5"
    junk.write_bytes(_caffemodel([("conv1_1", "Convolution", [np.zeros(3, np.float32)])]) * 40)
    with pytest.raises(mnet.NetError, match="not an HDF5 file"):
        n.load_caffemodel(junk)
    missing = tmp_path / "missing.h5"
    with pytest.raises(mnet.NetError, match="Couldn't open"):
        n.load_caffemodel(missing)


def test_hdf5_weights_written_by_libhdf5(tmp_path):
    """Every parameter layer of the net through a snapshot libhdf5 writes here (group B-tree with internal nodes: 40 layers),
    plus the reference's failure modes."""
    import numpy as np
    from tests.golden import make_hdf5_weights as mk
    if not mk.available():
        pytest.skip("libhdf5 is not installed on this machine (the committed fixture covers the reader)")
    n = Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576", height=64, width=128))
    rng = np.random.default_rng(9)
    layers, want = [], {}
    for i, name in enumerate(n.layer_names):
        shapes = n.param_shapes(i)
        if not shapes:
            continue
        arrs = [rng.standard_normal(s).astype(np.float32 if i % 3 else np.float64) for s in shapes]
        layers.append((name, arrs)); want[name] = arrs
    assert len(layers) >= 25
    path = tmp_path / "all.h5"
    # + 300 groups the net does not know: more than one B-tree node can index (32 leaves x 8 links) -> a two-level B-tree
    extra = [(f"zzz_extra_{k:03d}", [np.full((2, 2), k, np.float32)]) for k in range(300)]
    mk.write(path, layers[:20] + extra + layers[20:])
    n.load_caffemodel(path)
    for name, arrs in want.items():
        for j, a in enumerate(arrs):
            assert np.array_equal(n.get_param(name, j), a.astype(np.float32)), (name, j)
    w, b = want["conv1_1"]
    # element count differs: refused with both shapes in the message
    p1 = tmp_path / "count.h5"; mk.write(p1, [("conv1_1", [w[:32].astype(np.float32), b.astype(np.float32)])])
    with pytest.raises(mnet.NetError, match="shape mismatch"):
        n.load_caffemodel(p1)
    # more datasets than the layer has blobs (net.cpp:827-830) / a blob without its dataset (:836-845)
    p2 = tmp_path / "more.h5"; mk.write(p2, [("conv1_1", [w.astype(np.float32), b.astype(np.float32), b.astype(np.float32)])])
    with pytest.raises(mnet.NetError, match="Incompatible number of blobs"):
        n.load_caffemodel(p2)
    p3 = tmp_path / "fewer.h5"; mk.write(p3, [("conv1_1", [w.astype(np.float32)])])
    with pytest.raises(mnet.NetError, match="Incompatible number of blobs"):
        n.load_caffemodel(p3)
    # same count, other dims (a legacy 1x1x1xN bias): the reference reshapes the blob to the file's dims and reads it
    p4 = tmp_path / "legacy.h5"; mk.write(p4, [("conv1_1", [w.astype(np.float32), b.astype(np.float32).reshape(1, 1, 1, -1)])])
    n.load_caffemodel(p4)
    assert np.array_equal(n.get_param("conv1_1", 1).reshape(-1), b.astype(np.float32))
    # attributes on the groups and datasets (h5py / pycaffe tooling adds them): object-header continuation blocks and
    # attribute messages the reader has to walk past
    import ctypes as C
    h5, h5l = mk.libs()[0], mk.libs()[1]
    hid = C.c_int64
    h5.H5Fopen.restype = hid; h5.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    h5.H5Gopen2.restype = hid; h5.H5Gopen2.argtypes = [hid, C.c_char_p, hid]
    h5l.H5LTset_attribute_string.argtypes = [hid, C.c_char_p, C.c_char_p, C.c_char_p]
    h5l.H5LTset_attribute_float.argtypes = [hid, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t]
    p6 = tmp_path / "attrs.h5"
    w2 = (w * 0.5).astype(np.float32)
    mk.write(p6, [("conv1_1", [w2, b.astype(np.float32)])])
    f = h5.H5Fopen(os.fsencode(str(p6)), 1, 0)                # H5F_ACC_RDWR
    g = h5.H5Gopen2(f, b"data", 0)
    stat = (C.c_float * 16)(*range(16))
    for k in range(12):
        assert h5l.H5LTset_attribute_string(g, b"conv1_1", f"note{k}".encode(), b"x" * 120) >= 0
        assert h5l.H5LTset_attribute_float(g, b"conv1_1/0", f"stat{k}".encode(), stat, 16) >= 0
    assert h5l.H5LTset_attribute_string(f, b"data", b"origin", b"caffe snapshot") >= 0
    h5.H5Gclose(g); h5.H5Fclose(f)
    n.load_caffemodel(p6)
    assert np.array_equal(n.get_param("conv1_1", 0), w2)
    # no "data" group
    p5 = tmp_path / "nodata.h5"
    f = h5.H5Fcreate(os.fsencode(str(p5)), 2, 0, 0); g = h5.H5Gcreate2(f, b"weights", 0, 0, 0); h5.H5Gclose(g); h5.H5Fclose(f)
    with pytest.raises(mnet.NetError, match='no group "data"'):
        n.load_caffemodel(p5)
