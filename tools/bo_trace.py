#!/usr/bin/env python3
"""Phase timeline of BoxOutput's two one-workgroup kernels on the head outputs of a real frame (debug library:
make -C mscnn_amd/csrc trace).  The net (product library) runs the bench frame; its seven LFCN outputs are then fed to the trace
build's BoxOutput, whose thread 0 stamps s_memrealtime (100 MHz) at the phase boundaries of select_sort_kernel and
nms_scan_emit_kernel and at every 64-box chunk of the greedy scan.   python tools/bo_trace.py [--model kitti_car/mscnn-7s-576]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mscnn_amd import hipapi as hip, net as mnet, synth, zoo
hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libmscnn_hip_trace.so")

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="kitti_car/mscnn-7s-576")
ap.add_argument("--regime", default="mid")
a = ap.parse_args()
n = mnet.Net(prototxt_text=zoo.prototxt(a.model))
synth.load_into(n, a.regime)
shape = n.blob_shape("data")
n.set_blob("data", synth.frame(shape[2], shape[3]))
n.forward()
names = [nm for nm in n.layer_names if nm.startswith("LFCN_")]
heads = [torch.from_numpy(n.get_blob(nm)).cuda() for nm in names]
fields = zoo.KITTI_CAR_FIELDS
ds = [8, 8, 16, 16, 32, 32, 64]
d = hip.make_boxoutput_desc([tuple(h.shape[2:]) for h in heads], 1, heads[0].shape[1], fields, fields, ds, fg_thr=-5.0, iou_thr=0.65,
                            max_nms_num=2000, min_size=15.0)
layer = hip.BoxOutput(d)
L = hip.lib()
L.mscnn_debug_set_bo_trace.argtypes = [C.c_void_p]
for _ in range(5):
    rois, props, aids, nreal = layer.forward(heads)
tr = torch.zeros(256, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
L.mscnn_debug_set_bo_trace(C.c_void_p(tr.data_ptr()))
rois, props, aids, nreal = layer.forward(heads)
torch.cuda.synchronize()
L.mscnn_debug_set_bo_trace(None)
t = tr.cpu().numpy()
us = lambda a_, b_: (int(t[b_]) - int(t[a_])) / 100.0      # noqa: E731
print(f"{a.model} {a.regime}: candidates n = {t[30]}, K = {t[31]}, rows out = {nreal} (net: {n.blob_shape('proposals')[0]})")
print(f"select_sort_kernel: read n + key loads {us(0, 1):.1f} us, radix select {us(1, 2):.1f}, compaction + pad {us(2, 3):.1f}, "
      f"bitonic network {us(3, 4):.1f}, gather {us(4, 5):.1f}; total {us(0, 5):.1f}")
if t[11]:
    print(f"  select detail: digit pass 0 {us(1, 11):.1f} us, pass 1 {us(11, 12) if t[12] else 0:.1f}, sweep {us(12 if t[12] else 11, 20) if t[20] else 0:.1f}, "
          f"list rank ({t[29]} keys) {us(20, 21) if t[21] else 0:.1f}")
ch = t[64:128]
ch = ch[ch != 0]
dt = np.diff(ch) / 100.0
print(f"nms_scan_emit_kernel: to first chunk {(int(ch[0]) - int(t[8])) / 100.0:.1f} us, {len(ch)} chunks: per chunk min {dt.min():.2f} "
      f"median {np.median(dt):.2f} max {dt.max():.2f} us, scan total {us(8, 9):.1f}, emit {us(9, 10):.1f}; total {us(8, 10):.1f}")
print("  per-chunk us:", " ".join(f"{v:.1f}" for v in dt))
print(f"gap select_sort end -> scan start (mask kernel + launches): {us(5, 8):.1f} us")
