#!/usr/bin/env python3
"""HBM-side traffic per launch of the kernels of one Winograd layer from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
need separate passes: MI355X_MICROARCH.md, PMC slots), with the guide's gfx950 correction (FETCH_SIZE counts 64 B per 128-B
request: x2; WRITE_SIZE as is).  Usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json [f4|f3] [T_pad]
The output names the SHA-256 of the kernel sources it was measured on (wgemm.hip, winograd.hip): bench.py puts the figure into
its roofline object only while those files are unchanged."""
import collections, csv, hashlib, json, os, re, sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            key = "gemm" if ("wgemm_kernel" in n or "igemm_kernel" in n) else "gemm_fixup" if "igemm_fixup" in n else \
                "input_transform" if ("wino33_input" in n or "wino44_input" in n) else \
                "output_transform" if ("wino33_output" in n or "wino44_output" in n) else "zero_tail" if "wino_zero_tail" in n else None
            if key:
                acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
kb = {k: 2.0 * fetch.get(k, 0.0) + write.get(k, 0.0) for k in set(fetch) | set(write)}
planes, T_pad = (36, 1120) if len(sys.argv) <= 4 or sys.argv[4] == "f4" else (25, 1920)
if len(sys.argv) > 5:
    T_pad = int(sys.argv[5])
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha16():
    h = hashlib.sha256()
    for f in ("mscnn_amd/csrc/wgemm.hip", "mscnn_amd/csrc/winograd.hip"):
        h.update(open(os.path.join(_root, f), "rb").read())
    return h.hexdigest()[:16]

out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- "
                  "python tools/bench_layers.py --iters 6 --only conv4_2   (conv4_2: 1x512x72x240 -> 512, Winograd "
                  + (f"F(4x4,3x3), 36 planes of 512 x 512 x {T_pad})" if planes == 36 else f"F(3x3,3x3), 25 planes of 512 x 512 x {T_pad})"),
       "kernel_sources_sha16": sources_sha16(), "launch": "conv4_2",
       "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
       "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE = 64 B per 128-B request on gfx950 -> x2; WRITE_SIZE taken as is",
       "kernel": "wgemm_kernel on conv4_2's plane GEMMs (tile shape as planned: 256 x 160 since round 4)",
       "traffic_bytes_per_launch": int(1024 * kb.get("gemm", 0.0)),
       "algorithmic_bytes_per_launch": int(planes * (512 * 512 + 512 * T_pad + 512 * T_pad) * 4),
       "layer_traffic_bytes": {k: int(1024 * v) for k, v in kb.items()},
       "note": "fabric-side (L2 miss) bytes of ONE launch of the dominant kernel: it reads U and V and writes M once (algorithmic); "
               "re-reads beyond that are L2 misses of operand tiles shared between workgroups on different XCDs; the stream-K slabs "
               "(128 KB per workgroup, written through and read once) are part of the measured figure"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
