#!/usr/bin/env python3
"""MFMA-pipe utilisation of the plane-GEMM kernel from rocprofv3 SQ / GRBM counter passes (one counter_collection.csv per layer):
   busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)      (MFMA_BUSY sums busy cycles over every SIMD of the chip,
   GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs: 64 cycles per v_mfma_f32_32x32x2_f32 and SIMD -- MI355X_MICROARCH.md, cycle table)
   clock = GRBM_GUI_ACTIVE / 8 / kernel duration.
Usage: pmc_mfma.py out.json layer=counter_collection.csv [layer=...]
The output names the SHA-256 of the kernel sources it was measured on (wgemm.hip, winograd.hip): bench.py puts `mfma_busy` into its
roofline object only while those files are unchanged."""
import collections, csv, hashlib, json, os, re, sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha16():
    h = hashlib.sha256()
    for f in ("mscnn_amd/csrc/wgemm.hip", "mscnn_amd/csrc/winograd.hip"):
        h.update(open(os.path.join(_root, f), "rb").read())
    return h.hexdigest()[:16]


def gemm_rows(path):
    c = collections.defaultdict(list)
    dur = {}
    name = None
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            if "wgemm_kernel" not in n and "wf2conv_kernel" not in n:      # the plane GEMM, or conv1_2's one-launch Winograd kernel
                continue
            name = n
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not dur:
        return None
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    us = sum(dur.values()) / len(dur)
    out = {"kernel": name, "launches": len(dur), "avg_us": round(us, 1)}
    if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        out["clock_ghz"] = round(avg["GRBM_GUI_ACTIVE"] / 8 / us / 1e3, 3)
        out["mfma_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if "SQ_WAVE_CYCLES" in avg:
        for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in avg:
                out[k.lower() + "_share_of_wave_cycles"] = round(avg[k] / avg["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_BANK_CONFLICT" in avg:
        out["lds_bank_conflict_cycles"] = avg["SQ_LDS_BANK_CONFLICT"]
    return out


layers = {}
for arg in sys.argv[2:]:
    layer, path = arg.split("=", 1)
    row = gemm_rows(path)
    if row:
        layers[layer] = row
tot = sum(v["avg_us"] for v in layers.values() if "mfma_busy" in v)
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY "
                  "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -- python tools/bench_layers.py --only <layer> --iters 6   (one pass per layer)",
       "kernel_sources_sha16": sources_sha16(),
       "formula": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024); clock = GRBM_GUI_ACTIVE / 8 / duration",
       "layers": layers,
       "mfma_busy_time_weighted": round(sum(v["mfma_busy"] * v["avg_us"] for v in layers.values() if "mfma_busy" in v) / tot, 4) if tot else None,
       "note": "the plane-GEMM kernel (wgemm.hip) stand-alone per layer (tools/bench_layers.py), profiled clock; busy x clock / 2.4 GHz is the "
               "fraction of the 157.3 TFLOP/s peak the MFMA pipe was issued at (padding columns of the tile grid included)"}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
