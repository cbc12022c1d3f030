#!/usr/bin/env python3
"""The reference's CPU path (oracle/_ref: its own layer sources, im2col + MKL sgemm) on ONE thread over one full frame of the headline
configuration -- SURVEY 8(d)'s 1-core row, measured instead of extrapolated.  Takes 20 - 30 s; run once per CPU model, the result is
cached in profiles/r04_cpu_one_core.json and quoted by bench.py when the CPU model matches.
  python tools/cpu_one_core.py [--model kitti_car/mscnn-7s-576] [--out profiles/r04_cpu_one_core.json]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mscnn_amd import net as mnet, synth, zoo  # noqa: E402
from oracle import pynet, pyref  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default=bench.DEFAULT_MODEL)
ap.add_argument("--regime", default="mid")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_cpu_one_core.json"))
a = ap.parse_args()
assert pyref.available(), "oracle/_ref/libmscnn_ref.so is not built"
H, W = bench.MODELS[a.model]["hw"]
n = mnet.Net(prototxt_text=zoo.prototxt(a.model), device=-1)
layers = bench._layers_of(n)
ws = synth.weights(n.layer_names, n.layer_types, [n.param_shapes(i) for i in range(len(n.layer_names))], a.regime)
x = synth.frame(H, W, seed=1701, org_hw=bench.MODELS[a.model]["org_hw"])
prev = pyref.set_threads(1)
try:
    timings = []
    t0 = time.perf_counter()
    blobs = pynet.forward(layers, ws, {"data": x}, backend=pyref, timings=timings)
    dt = time.perf_counter() - t0
finally:
    pyref.set_threads(prev)
out = {"model": a.model, "regime": a.regime, "cpu": bench._cpu_model(), "threads": 1, "seconds_per_image": round(dt, 2),
       "rois": int(blobs["proposals"].shape[0]), "scope": "Net::Forward of the reference's own CPU layers (oracle/_ref), one 1x3x%dx%d frame" % (H, W),
       "slowest_layers_s": {nm: round(t, 2) for nm, ty, t in sorted(timings, key=lambda r: -r[2])[:6]}}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out))
