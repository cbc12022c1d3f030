#!/usr/bin/env python3
"""Kernel-stats summary (name, calls, total/avg/min/max us, %) from a rocprofv3 rocpd SQLite database."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
for n, c, s, a, mn, mx in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    print(f"{n[:100]:100s} {c:6d} {s/1e3:11.1f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
