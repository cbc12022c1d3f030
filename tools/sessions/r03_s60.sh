#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s60; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "f4 or winograd or fused_pool" 2>&1 | tail -4 ) > $O/ops.txt 2>&1
( timeout 600 python bench.py --no-robust --no-cpu-baseline 2>$O/bench.err | tail -1 ) > $O/bench.json
