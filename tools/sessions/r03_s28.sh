#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s28; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python tools/bo_trace.py 2>&1 | tail -12 ) > $O/bo_trace.txt 2>&1
( timeout 600 python bench.py --no-alt --no-robust --no-cpu-baseline --layers 2>$O/bench.err | tail -1 ) > $O/bench.json
