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
        if pname == "propagate_down":          # (training-only hint of the cascade DecodeBBox layers; kept for fidelity)
            lines += ["  propagate_down: 0", "  propagate_down: 0"]
            continue
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


BBOX_STDS = {1: (0.1, 0.1, 0.2, 0.2), 2: (0.05, 0.05, 0.1, 0.1), 3: (0.033, 0.033, 0.067, 0.067)}
SUFFIX = {1: "", 2: "_2nd", 3: "_3rd"}


def _bbox_reg(std):
    return ("bbox_reg_param", [("bbox_mean", 0)] * 4 + [("bbox_std", v) for v in std])


def _roi_features(L, feat, scale, rois, sfx, roi_pooled, align):
    """roi_pool_org / roi_pool_ctx (+ Concat) on `rois`: ROIPooling, or ROIAlign followed by the 2x2 / stride 1 AVE pooling."""
    for name, pad in (("org", 0), ("ctx", 0.25)):
        rp = ("roi_pooling_param", [("pooled_w", roi_pooled[1]), ("pooled_h", roi_pooled[0]), ("spatial_scale", scale), ("pad_ratio", pad)])
        if align:
            _layer(L, name=f"roi_grid_{name}{sfx}", type="ROIAlign", bottom=[feat, rois], top=[f"roi_grid_{name}{sfx}"], params=[rp])
            _layer(L, name=f"roi_pool_{name}{sfx}", type="Pooling", bottom=[f"roi_grid_{name}{sfx}"], top=[f"roi_pool_{name}{sfx}"],
                   params=[("pooling_param", [("pool", "AVE"), ("kernel_size", 2), ("stride", 1)])])
        else:
            _layer(L, name=f"roi_pool_{name}{sfx}", type="ROIPooling", bottom=[feat, rois], top=[f"roi_pool_{name}{sfx}"], params=[rp])
    _layer(L, name=f"roi_pool{sfx}", type="Concat", bottom=[f"roi_pool_org{sfx}", f"roi_pool_ctx{sfx}"], top=[f"roi_pool{sfx}"])


def _det_head(L, pooled, tag, roi_c1_pad, fc6, cls_num, bbox_out):
    """roi_c1 -> fc6 -> cls_pred (-> bbox_pred when bbox_out) on the pooled ROI features; layer names carry `tag`."""
    _layer(L, bottom=[pooled], top=[f"roi_c1{tag}"], name=f"roi_c1{tag}", type="Convolution",
           params=[("convolution_param", [("num_output", 512), ("kernel_size", 3)] + ([("pad", roi_c1_pad)] if roi_c1_pad else []))])
    _relu(L, f"roi_c1{tag}", f"roi_c1_relu{tag}")
    _layer(L, name=f"fc6{tag}", type="InnerProduct", bottom=[f"roi_c1{tag}"], top=[f"fc6{tag}"], params=[("inner_product_param", [("num_output", fc6)])])
    _relu(L, f"fc6{tag}", f"relu6{tag}")
    _layer(L, name=f"drop6{tag}", type="Dropout", bottom=[f"fc6{tag}"], top=[f"fc6{tag}"], params=[("dropout_param", [("dropout_ratio", 0.5)])])
    _layer(L, name=f"cls_pred{tag}", type="InnerProduct", bottom=[f"fc6{tag}"], top=[f"cls_pred{tag}"],
           params=[("inner_product_param", [("num_output", cls_num)])])
    if bbox_out:
        _layer(L, name=f"bbox_pred{tag}", type="InnerProduct", bottom=[f"fc6{tag}"], top=[f"bbox_pred{tag}"],
               params=[("inner_product_param", [("num_output", bbox_out)])])


def _detection_subnet(L, feat, scale, cls_num, roi_pooled, roi_c1_pad, fc6, stages, ensemble_3rd, outputs, align):
    """The detection sub-net of the deploy files.  stages == 1 and not outputs: the plain MS-CNN head (class-specific
    bbox_pred, 4 * cls_num).  Cascade (stages == 3): class-agnostic bbox_pred (8 outputs, columns 4..7 used), each stage's
    DecodeBBox turns its regression into the next stage's proposals; with ensemble_3rd the third stage also runs copies of
    the first two stages' heads on its ROI features and averages the three probabilities (citypersons / widerface)."""
    agnostic = stages > 1 or outputs
    for st in range(1, stages + 1):
        sfx = SUFFIX[st]
        rois = "proposals" + sfx
        if st > 1:
            _layer(L, name=rois, type="DecodeBBox", bottom=["bbox_pred" + SUFFIX[st - 1], "proposals" + SUFFIX[st - 1]], top=[rois],
                   params=[_bbox_reg(BBOX_STDS[st - 1]), ("propagate_down", [])])
        _roi_features(L, feat, scale, rois, sfx, roi_pooled, align)
        if st == 3 and ensemble_3rd:
            _det_head(L, "roi_pool_3rd", "_1st_3rd", roi_c1_pad, fc6, cls_num, 0)
            _det_head(L, "roi_pool_3rd", "_2nd_3rd", roi_c1_pad, fc6, cls_num, 0)
        _det_head(L, "roi_pool" + sfx, sfx, roi_c1_pad, fc6, cls_num, 8 if agnostic else 4 * cls_num)
    if outputs:
        order = {1: "1st", 2: "2nd", 3: "3rd"}
        for st in range(1, stages + 1):
            _layer(L, name=f"output_bbox_{order[st]}", type="DecodeBBox", bottom=["bbox_pred" + SUFFIX[st], "proposals" + SUFFIX[st]],
                   top=[f"output_bbox_{order[st]}"], params=[_bbox_reg(BBOX_STDS[st])])
        probs = [("cls_prob_1st", "cls_pred")] + ([("cls_prob_2nd", "cls_pred_2nd")] if stages >= 2 else [])
        if stages == 3:
            probs += ([("cls_prob_1st_3rd", "cls_pred_1st_3rd"), ("cls_prob_2nd_3rd", "cls_pred_2nd_3rd")] if ensemble_3rd else []) + [("cls_prob_3rd", "cls_pred_3rd")]
        for top, bottom in probs:
            _layer(L, name=top, type="Softmax", bottom=[bottom], top=[top], params=[("softmax_param", [("axis", 1)])])
        if stages == 3 and ensemble_3rd:
            _layer(L, name="cls_prob_3rd_avg", type="Eltwise", bottom=["cls_prob_1st_3rd", "cls_prob_2nd_3rd", "cls_prob_3rd"], top=["cls_prob_3rd_avg"],
                   params=[("eltwise_param", [("operation", "SUM")] + [("coeff", 0.33333333)] * 3)])


def mscnn_deploy(height, width, cls_num, head_kernels, field_w, field_h, n_heads=7, fg_thr=-5, iou_thr=0.65,
                 max_nms_num=2000, roi_pooled=(7, 7), upsample2x=False, roi_c1_pad=0, fc6=4096, min_size=None,
                 stages=1, ensemble_3rd=False, outputs=False, proposal_bbox_reg=False, batch=1):
    """head_kernels: [(kw, kh) small, (kw, kh) large], e.g. [(5, 5), (7, 7)] or [(3, 5), (5, 7)] (names are WxH).
    stages / ensemble_3rd / outputs: the cascade deploys (see _detection_subnet); proposal_bbox_reg: BoxOutput carries the
    bbox_reg_param normalisation (citypersons)."""
    L = ['name: "MSCNN"', 'input: "data"', f"input_dim: {batch}", "input_dim: 3", f"input_dim: {height}", f"input_dim: {width}"]      # (the shipped files: dim 1)
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
    _layer(L, bottom=heads, top=["proposals", "proposals_score"], name="proposals", type="BoxOutput",
           params=[("box_output_param", bo)] + ([_bbox_reg(BBOX_STDS[1])] if proposal_bbox_reg else []))
    feat, scale = "conv4_3", 0.125
    if upsample2x:
        _layer(L, bottom=["conv4_3"], top=["conv4_3_2x"], name="conv4_3_2x", type="Deconvolution",
               params=[("convolution_param", [("kernel_size", 4), ("stride", 2), ("num_output", 512), ("group", 512), ("pad", 1),
                                              ("weight_filler", '{ type: "bilinear" }'), ("bias_term", "false")]),
                       ("param", [("lr_mult", 0), ("decay_mult", 0)])])
        feat, scale = "conv4_3_2x", 0.25
    _detection_subnet(L, feat, scale, cls_num, roi_pooled, roi_c1_pad, fc6, stages, ensemble_3rd, outputs, align=False)
    return "\n".join(L) + "\n"


WIDERFACE_FIELDS = [12, 16, 24, 32, 48, 64, 96, 128, 196, 256, 384, 480]


def widerface_cascade_deploy(height=512, width=512, max_nms_num=3000, min_size=5, batch=1):
    """examples/widerface/cascade-mscnn-12s-align: a 3x3 `rpn_k_conv` per scale feeding 1x1 proposal heads named by their
    field size (12 heads on conv4_3 / conv5_3 / pool5 / an AVE-pooled pool6), ROIAlign 5x5 + 2x2 AVE pooling instead of
    ROIPooling, three cascade stages with the third-stage ensemble."""
    L = ['name: "MSCNN"', 'input: "data"', f"input_dim: {batch}", "input_dim: 3", f"input_dim: {height}", f"input_dim: {width}"]      # (the shipped files: dim 1)
    bottom = "data"
    heads = []

    def rpn(level, src, fields):
        _conv(L, src, f"rpn_{level}_conv", 512, 3, 3, 1, 1)
        _relu(L, f"rpn_{level}_conv", f"rpn_{level}_relu")
        for f in fields:
            name = f"LFCN_{level}_{f}x{f}"
            _layer(L, name=name, type="Convolution", bottom=[f"rpn_{level}_conv"], top=[name],
                   params=[("convolution_param", [("num_output", 6), ("pad", 0), ("kernel_size", 1)])])
            heads.append(name)

    for block, nconv, ch in VGG_TRUNK:
        for i in range(1, nconv + 1):
            name = f"conv{block}_{i}"
            _conv(L, bottom, name, ch, 3, 3, 1, 1)
            _relu(L, name, f"relu{block}_{i}")
            bottom = name
        if block == 4:
            rpn(1, "conv4_3", [12, 16, 24, 32, 48])
        if block == 5:
            rpn(2, "conv5_3", [64, 96])
        _pool(L, bottom, f"pool{block}")
        bottom = f"pool{block}"
    rpn(3, "pool5", [128, 192])
    _layer(L, bottom=["pool5"], top=["pool6"], name="pool6", type="Pooling", params=[("pooling_param", [("pool", "AVE"), ("kernel_size", 2), ("stride", 2)])])
    rpn(4, "pool6", [256, 384, 480])
    ds = [8] * 5 + [16] * 2 + [32] * 2 + [64] * 3
    bo = [("fg_thr", -3), ("iou_thr", 0.65), ("nms_type", '"IOU"')]
    bo += [("field_w", v) for v in WIDERFACE_FIELDS] + [("field_h", v) for v in WIDERFACE_FIELDS] + [("downsample_rate", v) for v in ds]
    bo += [("field_whr", 4), ("field_xyr", 1), ("min_size", min_size), ("max_nms_num", max_nms_num)]
    _layer(L, bottom=heads, top=["proposals", "proposals_score"], name="proposals", type="BoxOutput",
           params=[("box_output_param", bo), _bbox_reg(BBOX_STDS[1])])
    _detection_subnet(L, "conv4_3", 0.125, 2, (5, 5), 1, 2048, 3, True, True, align=True)
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


CITYPERSONS = dict(height=1344, width=2688, cls_num=2, head_kernels=[(3, 5), (5, 7)], n_heads=8, fg_thr=-3,
                   field_w=[30, 42, 60, 84, 120, 168, 240, 336], field_h=[60, 84, 120, 168, 240, 336, 480, 672], roi_pooled=(8, 4),
                   upsample2x=True, roi_c1_pad=1, fc6=2048, outputs=True, proposal_bbox_reg=True)
MODELS.update({
    "kitti_car/cascade-mscnn-7s-576-2x": (dict(height=576, width=1920, cls_num=5, head_kernels=[(5, 5), (7, 7)], field_w=KITTI_CAR_FIELDS,
                                               field_h=KITTI_CAR_FIELDS, upsample2x=True, stages=3, outputs=True),
                                          "kitti_car/cascade-mscnn-7s-576-2x/mscnn_deploy.prototxt"),
    "citypersons/mscnn-8s-1344-2x": (dict(CITYPERSONS), "citypersons/mscnn-8s-1344-2x/mscnn_deploy.prototxt"),
    "citypersons/cascade-mscnn-8s-1344-2x": (dict(CITYPERSONS, stages=3, ensemble_3rd=True),
                                             "citypersons/cascade-mscnn-8s-1344-2x/mscnn_deploy.prototxt"),
    "widerface/cascade-mscnn-12s-align": (dict(height=512, width=512, widerface=True), "widerface/cascade-mscnn-12s-align/mscnn_deploy.prototxt"),
})


def prototxt(model, **overrides):
    kw = dict(MODELS[model][0])
    kw.update(overrides)
    if kw.pop("widerface", False):
        return widerface_cascade_deploy(**kw)
    return mscnn_deploy(**kw)
