#!/usr/bin/env python3
"""Idle time on the device between consecutive kernels of a rocprofv3 --kernel-trace run (kernel_trace.csv): per pair of
neighbouring kernels (previous -> next) the median gap over the run, for gaps above a threshold.  Where a frame loses time to the
host (the BoxOutput row-count round trip) or to launch latency shows up here, not in the per-kernel table."""
import csv, re, sys
from collections import defaultdict
from statistics import median


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"[<(].*", "", n)[:40]


rows = sorted(({"name": short(r["Kernel_Name"]), "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])}
               for r in csv.DictReader(open(sys.argv[1], newline=""))), key=lambda r: r["s"])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
gaps = defaultdict(list)
busy = sum(r["e"] - r["s"] for r in rows)
for a, b in zip(rows, rows[1:]):
    g = (b["s"] - a["e"]) / 1e3
    if g < 2000:                       # (not the pauses between the benchmark's phases)
        gaps[(a["name"], b["name"])].append(g)
print(f"# {len(rows)} kernels, busy {busy / 1e6:.2f} ms, span {(rows[-1]['e'] - rows[0]['s']) / 1e6:.2f} ms")
tot = 0.0
for (a, b), v in sorted(gaps.items(), key=lambda kv: -median(kv[1]) * len(kv[1])):
    m = median(v)
    if m < thr:
        continue
    tot += m * len(v)
    print(f"{a:40s} -> {b:40s} n={len(v):5d} median {m:8.1f} us  max {max(v):8.1f}")
small = [g for v in gaps.values() for g in v if g < thr]
print(f"# gaps >= {thr} us: {tot / 1e3:.2f} ms in total; {len(small)} gaps below it: median {median(small) if small else 0:.2f} us, sum {sum(small) / 1e3:.2f} ms")
