#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s48; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "boxoutput or nms or detections or final or uncapped" 2>&1 | tail -8 ) > $O/ops.txt 2>&1
( timeout 300 python tools/bo_trace.py 2>&1 | tail -8 ) > $O/bo_trace.txt 2>&1
( timeout 600 python bench.py --no-alt --no-robust --no-cpu-baseline --layers 2>$O/bench.err | tail -1 ) > $O/bench.json
