#!/usr/bin/env python3
"""Compact view of a rocprofv3 kernel_stats.csv: short kernel name, calls, average us, share."""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1], newline="")):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"\(.*", "", n) if not n.startswith("void") else re.sub(r"\)\s*\(.*", ")", n)
    n = n.replace("void ", "")
    if float(r["Percentage"]) < 0.3: continue
    print(f"{n[:80]:80s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
