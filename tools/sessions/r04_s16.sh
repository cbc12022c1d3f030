#!/bin/bash
# round 4, session 16: kernel timeline of the frame tail (heads -> BoxOutput -> ROI pooling) with the side-stream prefetch on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s16; mkdir -p $O; export PYTHONUNBUFFERED=1
cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt --no-robust > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err; cd $GRAFT_REPO_ROOT
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete frame: from the last conv3x3_c3_kernel on
idx = [i for i, r in enumerate(rows) if "conv3x3_c3" in r["Kernel_Name"]]
start = idx[6]      # a frame of the timed loop (the last frames of a bench run are the per-layer timing passes: no prefetch there)
stop = idx[7]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:stop]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?')}  {n}")
PY
rm -rf $O/tr
