#!/usr/bin/env python3
"""Print the kernel-resource-usage table of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kru.o"] + sys.argv[2:],
                     capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:90]
    print(f"{name:92s} V={r.get('VGPRs')} A={r.get('AGPRs')} S={r.get('TotalSGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
