"""Generator of the MS-CNN deploy network descriptions (Caffe prototxt text).

The reference ships its nets as examples/<dataset>/<model>/mscnn_deploy.prototxt; those files are not copied into
this repository.  The same graphs are emitted here from a handful of parameters, in the same legacy text format
(`input:` + 4 x `input_dim:`), so the C++ runtime loads them exactly as it loads the reference's files.
tests/test_prototxt.py checks -- where the reference checkout is present -- that every generated net parses to the
same layer graph (names, types, bottoms, tops, parameters) as the reference file it stands for.
"""

VGG_TRUNK = [  # (block, convs, channels)
    (1, 2, 64), (2, 2, 128), (3, 3, 256), (4, 3, 512), (5, 3, 512),
]


def _layer(lines, **kw):
    lines.append("layer {")
    for b in kw.get("bottom", []):
        lines.append(f'  bottom: "{b}"')
    for t in kw.get("top", []):
        lines.append(f'  top: "{t}"')
    lines.append(f'  name: "{kw["name"]}"')
    lines.append(f'  type: "{kw["type"]}"')
    for pname, fields in kw.get("params", []):
        lines.append(f"  {pname} {{")
        for k, v in fields:
            lines.append(f"    {k}: {v}")
        lines.append("  }")
    lines.append("}")


def _conv(lines, bottom, name, cout, kh, kw, ph, pw):
    if kh == kw:
        f = [("num_output", cout)] + ([("pad", ph)] if ph else []) + [("kernel_size", kh)]
    else:
        f = [("num_output", cout), ("pad_w", pw), ("pad_h", ph), ("kernel_w", kw), ("kernel_h", kh)]
    _layer(lines, bottom=[bottom], top=[name], name=name, type="Convolution", params=[("convolution_param", f)])


def _relu(lines, blob, name):
    _layer(lines, bottom=[blob], top=[blob], name=name, type="ReLU")


def _pool(lines, bottom, name):
    _layer(lines, bottom=[bottom], top=[name], name=name, type="Pooling",
           params=[("pooling_param", [("pool", "MAX"), ("kernel_size", 2), ("stride", 2)])])


def mscnn_deploy(height, width, cls_num, head_kernels, field_w, field_h, n_heads=7, fg_thr=-5, iou_thr=0.65,
                 max_nms_num=2000, roi_pooled=(7, 7), upsample2x=False, roi_c1_pad=0, fc6=4096, min_size=None):
    """head_kernels: [(kw, kh) small, (kw, kh) large], e.g. [(5, 5), (7, 7)] or [(3, 5), (5, 7)] (names are WxH)."""
    L = ['name: "MSCNN"', 'input: "data"', "input_dim: 1", "input_dim: 3", f"input_dim: {height}", f"input_dim: {width}"]
    bottom = "data"
    heads = []
    cout_head = cls_num + 4

    def head(bottom, level, which):
        kw, kh = head_kernels[which]
        name = f"LFCN_{level}_{kw}x{kh}"
        _conv(L, bottom, name, cout_head, kh, kw, kh // 2, kw // 2)
        heads.append(name)

    for block, nconv, ch in VGG_TRUNK:
        for i in range(1, nconv + 1):
            name = f"conv{block}_{i}"
            _conv(L, bottom, name, ch, 3, 3, 1, 1)
            _relu(L, name, f"relu{block}_{i}")
            bottom = name
        if block == 4:
            _conv(L, "conv4_3", "loss1_conv1", 512, 3, 3, 1, 1)
            _relu(L, "loss1_conv1", "loss_relu1")
            head("loss1_conv1", 1, 0); head("loss1_conv1", 1, 1)
        if block == 5:
            head("conv5_3", 2, 0); head("conv5_3", 2, 1)
        _pool(L, bottom, f"pool{block}")
        bottom = f"pool{block}"
    _conv(L, "pool5", "conv6_1", 512, 3, 3, 1, 1)
    _relu(L, "conv6_1", "relu6_1")
    head("conv6_1", 3, 0); head("conv6_1", 3, 1)
    _pool(L, "conv6_1", "pool6")
    head("pool6", 4, 0)
    if n_heads == 8:
        head("pool6", 4, 1)
    assert len(heads) == n_heads == len(field_w) == len(field_h)
    ds = [8, 8, 16, 16, 32, 32, 64, 64][:n_heads]
    bo = [("fg_thr", fg_thr), ("iou_thr", iou_thr), ("nms_type", '"IOU"')]
    bo += [("field_w", v) for v in field_w] + [("field_h", v) for v in field_h] + [("downsample_rate", v) for v in ds]
    bo += [("field_whr", 2), ("field_xyr", 2), ("max_nms_num", max_nms_num)]
    if min_size is not None:
        bo.append(("min_size", min_size))
    _layer(L, bottom=heads, top=["proposals", "proposals_score"], name="proposals", type="BoxOutput", params=[("box_output_param", bo)])
    feat, scale = "conv4_3", 0.125
    if upsample2x:
        _layer(L, bottom=["conv4_3"], top=["conv4_3_2x"], name="conv4_3_2x", type="Deconvolution",
               params=[("convolution_param", [("kernel_size", 4), ("stride", 2), ("num_output", 512), ("group", 512), ("pad", 1),
                                              ("weight_filler", '{ type: "bilinear" }'), ("bias_term", "false")]),
                       ("param", [("lr_mult", 0), ("decay_mult", 0)])])
        feat, scale = "conv4_3_2x", 0.25
    for name, pad in (("roi_pool_org", 0), ("roi_pool_ctx", 0.25)):
        _layer(L, name=name, type="ROIPooling", bottom=[feat, "proposals"], top=[name],
               params=[("roi_pooling_param", [("pooled_w", roi_pooled[1]), ("pooled_h", roi_pooled[0]), ("spatial_scale", scale), ("pad_ratio", pad)])])
    _layer(L, name="roi_pool", type="Concat", bottom=["roi_pool_org", "roi_pool_ctx"], top=["roi_pool"])
    _layer(L, bottom=["roi_pool"], top=["roi_c1"], name="roi_c1", type="Convolution",
           params=[("convolution_param", [("num_output", 512), ("kernel_size", 3)] + ([("pad", roi_c1_pad)] if roi_c1_pad else []))])
    _relu(L, "roi_c1", "roi_c1_relu")
    _layer(L, name="fc6", type="InnerProduct", bottom=["roi_c1"], top=["fc6"], params=[("inner_product_param", [("num_output", fc6)])])
    _relu(L, "fc6", "relu6")
    _layer(L, name="drop6", type="Dropout", bottom=["fc6"], top=["fc6"], params=[("dropout_param", [("dropout_ratio", 0.5)])])
    _layer(L, name="cls_pred", type="InnerProduct", bottom=["fc6"], top=["cls_pred"], params=[("inner_product_param", [("num_output", cls_num)])])
    _layer(L, name="bbox_pred", type="InnerProduct", bottom=["fc6"], top=["bbox_pred"], params=[("inner_product_param", [("num_output", 4 * cls_num)])])
    return "\n".join(L) + "\n"


KITTI_CAR_FIELDS = [60, 84, 120, 168, 240, 336, 480]

# name -> (generator kwargs, reference file it mirrors under the reference's examples/)
MODELS = {
    "kitti_car/mscnn-7s-384": (dict(height=384, width=1280, cls_num=5, head_kernels=[(5, 5), (7, 7)], field_w=[40, 56, 80, 112, 160, 224, 320],
                                    field_h=[40, 56, 80, 112, 160, 224, 320]), "kitti_car/mscnn-7s-384/mscnn_deploy.prototxt"),
    "kitti_car/mscnn-7s-576": (dict(height=576, width=1920, cls_num=5, head_kernels=[(5, 5), (7, 7)], field_w=KITTI_CAR_FIELDS,
                                    field_h=KITTI_CAR_FIELDS), "kitti_car/mscnn-7s-576/mscnn_deploy.prototxt"),
    "kitti_car/mscnn-8s-768-trainval": (dict(height=768, width=2560, cls_num=5, head_kernels=[(5, 5), (7, 7)], n_heads=8,
                                             field_w=KITTI_CAR_FIELDS + [672], field_h=KITTI_CAR_FIELDS + [672]),
                                        "kitti_car/mscnn-8s-768-trainval/mscnn_deploy.prototxt"),
    "kitti_ped_cyc/mscnn-7s-576-2x": (dict(height=576, width=1920, cls_num=3, head_kernels=[(3, 5), (5, 7)], fg_thr=-7,
                                           field_w=[40, 56, 80, 112, 160, 224, 360], field_h=KITTI_CAR_FIELDS, roi_pooled=(7, 5),
                                           upsample2x=True, fc6=2048), "kitti_ped_cyc/mscnn-7s-576-2x/mscnn_deploy.prototxt"),
    "caltech/mscnn-7s-480": (dict(height=480, width=640, cls_num=2, head_kernels=[(3, 5), (5, 7)], field_w=[20, 28, 40, 56, 80, 112, 160],
                                  field_h=[40, 56, 80, 112, 160, 224, 320], roi_pooled=(8, 4), roi_c1_pad=1, fc6=2048),
                             "caltech/mscnn-7s-480/mscnn_deploy.prototxt"),
}


def prototxt(model, **overrides):
    kw = dict(MODELS[model][0])
    kw.update(overrides)
    return mscnn_deploy(**kw)
