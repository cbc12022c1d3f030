#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s59; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "f4" 2>&1 | tail -4 ) > $O/ops.txt 2>&1
( timeout 400 python tools/bench_layers.py --only conv --ab flags=0,4096 --iters 60 2>&1 | grep -v amdgpu | grep -E "winograd_f4" | cut -c1-170 ) > $O/ab.txt 2>&1
