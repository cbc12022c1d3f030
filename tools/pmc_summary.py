#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv): name, calls, avg duration, counters."""
import csv, re, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = n if len(n) < 90 else n[:87] + "..."
        rows[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[n][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for n, cs in rows.items():
    if flt and flt not in n: continue
    d = list(dur[n].values())
    print(f"{n}\n   calls={len(d)} avg_us={sum(d)/len(d):.1f}  " + "  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(cs.items())))
