import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mscnn_amd import hipapi as hip
from oracle import pyoracle as orc
rng = np.random.default_rng(7)
for n in (256, 300, 320, 330, 384, 448, 500, 576, 640, 1100):
    x = rng.uniform(0, 800, n); y = rng.uniform(0, 300, n); w = rng.uniform(20, 200, n); h = rng.uniform(20, 150, n)
    boxes = np.stack([x, y, w, h], 1).astype(np.float32)
    ref = np.asarray(orc.nms_greedy(boxes, 0.65, "IOU"), bool)
    got = hip.nms_greedy(torch.from_numpy(boxes).cuda(), 0.65, "IOU").cpu().numpy().astype(bool)
    bad = np.nonzero(ref != got)[0]
    print(f"n={n}: kept ref {ref.sum()} got {got.sum()} first diffs {bad[:10]}")
    def iou(a, b):
        iw = min(a[0] + a[2], b[0] + b[2]) - max(a[0], b[0]); ih = min(a[1] + a[3], b[1] + b[3]) - max(a[1], b[1])
        if iw <= 0 or ih <= 0: return 0.0
        o = iw * ih; return o / (a[2] * a[3] + b[2] * b[3] - o)
    words = lambda a: [hex(int(sum(int(a[k]) << (k % 64) for k in range(c * 64, min(n, c * 64 + 64))))) for c in range((n + 63) // 64)]      # noqa: E731
    if len(bad):
        print("   ref", words(ref)); print("   got", words(got))
    for j in bad[:1]:
        sup = [(i, bool(ref[i]), bool(got[i])) for i in range(j) if iou(boxes[i], boxes[j]) > 0.65]
        print(f"  box {j} (chunk {j // 64}, bit {j % 64}): ref keep {ref[j]} got {got[j]}; overlapping earlier boxes (idx, ref kept, got kept): {sup}")
