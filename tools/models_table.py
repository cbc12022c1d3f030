#!/usr/bin/env python3
"""One table row per bench.py JSON line (stdin or files): the per-config summary committed as profiles/rNN_models.txt."""
import json, sys
rows = []
for path in sys.argv[1:] or ["/dev/stdin"]:
    for line in open(path):
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        c, rf = r.get("config", {}), r.get("roofline", {})
        name = c.get("workload", "?").split(",")[0]
        blk = rf.get("conv3_5_block", {})
        rows.append(f"{name:46s} B={c.get('batch', 1):<2d} {r['dtype']:5s} {r['value']:8.2f} images/s  {r['ms_per_step']:7.3f} ms/step  "
                    f"roofline.frac {rf.get('frac', float('nan')):.4f}  conv3-5 executed_frac {blk.get('executed_frac', float('nan')):.4f}  "
                    f"parity_ok {r.get('parity_ok')}  mean_rois {c.get('mean_rois')}  steps {r['steps']}")
print("\n".join(rows))
