#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s39; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "roipool or boxoutput or nms" 2>&1 | tail -6 ) > $O/ops.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_net.py tests/test_golden.py -q -x -m "gpu and not slow" -k "layerwise or partial or unfused or dynamic or golden or roi" 2>&1 | tail -6 ) > $O/net.txt 2>&1
( timeout 600 python bench.py --no-alt --no-robust --no-cpu-baseline --layers 2>$O/bench.err | tail -1 ) > $O/bench.json
