"""Second witnesses for the two stages whose oracle is a from-source restatement with no MATLAB run to pin it:
the final detection stage (examples/kitti_car/run_mscnn_detection.m:75-120 + utils/bbNms.m:112-126) and `imresize`
(run_mscnn_detection.m:66).  Written independently of oracle/mscnn_oracle.c and oracle/pyoracle.py, by a different route:

  * final stage: vectorised numpy, one MATLAB statement per line, MATLAB's own types (single until `double(...)`); the greedy
    NMS is expressed with whole-array masks per kept box instead of the i/j double loop;
  * imresize: EXACT rational arithmetic (fractions.Fraction) of the published `contributions` recipe, scalar loops, 1-based
    indices mirrored through the literal `aux = [1:in, in:-1:1]` table -- the true real-number result before the uint8 rounding,
    so it also shows where float64 summation order could flip a .5 case.

TEST INFRASTRUCTURE ONLY."""
from fractions import Fraction

import numpy as np


# ------------------------------------------------------------------------------------------------ final detection stage
def bb_nms_maxg(bbs, overlap=0.5):
    """bbNms(bbs, 'type','maxg','overlap',overlap,'ovrDnm','union') for bbs [n, >=5] = [x y w h score ...] (double)."""
    bbs = np.asarray(bbs, np.float64)
    if len(bbs) == 0:
        return bbs
    order = np.argsort(-bbs[:, 4], kind="stable")            # [~,ord] = sort(bbs(:,5),'descend'): MATLAB's sort is stable
    bbs = bbs[order]
    n = len(bbs)
    kp = np.ones(n, bool)
    area = bbs[:, 2] * bbs[:, 3]
    xs, xe, ys, ye = bbs[:, 0], bbs[:, 0] + bbs[:, 2], bbs[:, 1], bbs[:, 1] + bbs[:, 3]
    for i in range(n):
        if not kp[i]:                                         # greedy: a suppressed box suppresses nothing
            continue
        j = np.arange(i + 1, n)
        iw = np.minimum(xe[i], xe[j]) - np.maximum(xs[i], xs[j])
        ih = np.minimum(ye[i], ye[j]) - np.maximum(ys[i], ys[j])
        hit = kp[j] & (iw > 0) & (ih > 0)
        o = iw * ih
        u = area[i] + area[j] - o
        with np.errstate(divide="ignore", invalid="ignore"):
            hit &= (o / u) > overlap
        kp[j[hit]] = False
    return bbs[kp]


def final_stage(bbox_preds, cls_pred, proposals_score, cls_id, bbox_means=(0, 0, 0, 0), bbox_stds=(0.1, 0.1, 0.2, 0.2),
                proposal_thr=-10.0, ratios=(1.0, 1.0), org_hw=(375, 1242), overlap=0.5):
    """run_mscnn_detection.m:75-120 for one class.  Inputs are the three net outputs as single arrays [R, ...].
    Returns (dets [D, 5] double [x y w h prob], ids [D] 0-based input rows)."""
    f32 = np.float32
    bbox_preds = np.asarray(bbox_preds, f32).reshape(len(proposals_score), -1)
    cls_pred = np.asarray(cls_pred, f32).reshape(len(proposals_score), -1)
    tmp = np.asarray(proposals_score, f32).reshape(len(proposals_score), -1)[:, 1:].copy()    # tmp(:,2:end): x1 y1 x2 y2 score
    tmp[:, 2] = tmp[:, 2] - tmp[:, 0]
    tmp[:, 3] = tmp[:, 3] - tmp[:, 1]
    proposal_pred = tmp
    proposal_score = proposal_pred[:, -1]
    keep_id = np.flatnonzero((proposal_score >= f32(proposal_thr)) & (proposal_pred[:, 2] != 0) & (proposal_pred[:, 3] != 0))
    proposal_pred, bbox_preds, cls_pred = proposal_pred[keep_id], bbox_preds[keep_id], cls_pred[keep_id]
    orgH, orgW = float(org_hw[0]), float(org_hw[1])
    bbox_pred = bbox_preds[:, cls_id * 4 - 4:cls_id * 4]                                       # id*4-3 : id*4 (1-based)
    bbox_pred = bbox_pred * np.asarray(bbox_stds, f32)[None, :]
    bbox_pred = bbox_pred + np.asarray(bbox_means, f32)[None, :]
    exp_score = np.exp(cls_pred)                                                               # single, no max subtraction
    sum_exp_score = np.zeros(len(cls_pred), f32)
    for c in range(cls_pred.shape[1]):                                                         # sum(exp_score,2), left to right
        sum_exp_score = sum_exp_score + exp_score[:, c]
    prob = exp_score[:, cls_id - 1] / sum_exp_score
    ctr_x = proposal_pred[:, 0] + f32(0.5) * proposal_pred[:, 2]
    ctr_y = proposal_pred[:, 1] + f32(0.5) * proposal_pred[:, 3]
    tx = bbox_pred[:, 0] * proposal_pred[:, 2] + ctr_x
    ty = bbox_pred[:, 1] * proposal_pred[:, 3] + ctr_y
    tw = proposal_pred[:, 2] * np.exp(bbox_pred[:, 2])
    th = proposal_pred[:, 3] * np.exp(bbox_pred[:, 3])
    tx = tx - tw / f32(2); ty = ty - th / f32(2)
    # ratios is a double row vector: single ./ double -> single in MATLAB (the double operand is converted)
    tx = tx / f32(ratios[1]); tw = tw / f32(ratios[1])
    ty = ty / f32(ratios[0]); th = th / f32(ratios[0])
    tx = np.maximum(f32(0), tx); ty = np.maximum(f32(0), ty)
    tw = np.minimum(tw, f32(orgW) - tx); th = np.minimum(th, f32(orgH) - ty)
    bbset = np.stack([tx, ty, tw, th, prob], 1).astype(np.float64)                             # double([tx ty tw th prob])
    bbset = np.concatenate([bbset, keep_id[:, None].astype(np.float64)], 1)                    # idlist column
    out = bb_nms_maxg(bbset, overlap)
    return out[:, :5], out[:, 5].astype(np.int32)


# ------------------------------------------------------------------------------------------------ imresize, exact
def _cubic_exact(x):
    """The bicubic kernel of imresize.m (`cubic`, a = -0.5) on a Fraction."""
    ax = abs(x)
    if ax <= 1:
        return Fraction(3, 2) * ax ** 3 - Fraction(5, 2) * ax ** 2 + 1
    if ax <= 2:
        return Fraction(-1, 2) * ax ** 3 + Fraction(5, 2) * ax ** 2 - 4 * ax + 2
    return Fraction(0)


def contributions_exact(in_len, out_len):
    """imresize.m `contributions(in_length, out_length, scale, @cubic, 4, antialiasing = true)`: per output sample the list of
    (1-based input index after mirroring, exact weight)."""
    scale = Fraction(out_len, in_len)
    kernel_width = Fraction(4)
    if scale < 1:
        kernel_width = kernel_width / scale
    aux = list(range(1, in_len + 1)) + list(range(in_len, 0, -1))
    P = -((-kernel_width.numerator) // kernel_width.denominator) + 2              # ceil(kernel_width) + 2
    out = []
    for x in range(1, out_len + 1):
        u = Fraction(x) / scale + Fraction(1, 2) * (1 - 1 / scale)
        left = (u - kernel_width / 2).numerator // (u - kernel_width / 2).denominator   # floor
        taps = []
        for k in range(P):
            idx = left + k
            d = u - idx
            w = scale * _cubic_exact(scale * d) if scale < 1 else _cubic_exact(d)
            taps.append((aux[(idx - 1) % len(aux)], w))
        total = sum(w for _, w in taps)
        out.append([(i, w / total) for i, w in taps])
    return out


def imresize_exact(img, out_h, out_w):
    """imresize(uint8 image, [out_h out_w]) with exact weights: returns (uint8 result, min distance of any pre-rounding value
    to a .5 rounding boundary over both passes) -- a distance of 0 would be a genuine tie, tiny distances are where a float64
    implementation could legitimately differ by one grey level."""
    img = np.asarray(img, np.uint8)
    sh, sw = Fraction(out_h, img.shape[0]), Fraction(out_w, img.shape[1])
    order = (0, 1) if sh <= sw else (1, 0)
    margin = Fraction(1)
    cur = img.astype(object)
    for ax in order:
        n_out = out_h if ax == 0 else out_w
        cb = contributions_exact(cur.shape[ax], n_out)
        a = np.moveaxis(cur, ax, 0)
        res = np.empty((n_out,) + a.shape[1:], object)
        for o, taps in enumerate(cb):
            acc = 0
            for i, w in taps:
                acc = acc + a[i - 1] * w
            res[o] = acc
        flat = res.reshape(-1)
        rounded = np.empty(flat.shape, object)
        for k, v in enumerate(flat):
            v = Fraction(v)
            r = (v + Fraction(1, 2)).numerator // (v + Fraction(1, 2)).denominator       # round half up (values are >= -0.5 here)
            frac = v - (v.numerator // v.denominator)
            margin = min(margin, abs(frac - Fraction(1, 2)))
            rounded[k] = min(255, max(0, r))
        cur = np.moveaxis(rounded.reshape(res.shape), 0, ax)
    return cur.astype(np.uint8), float(margin)
