#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s11; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/gputests.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -s -k "robustness" 2>&1 | grep -E "ROBUST|x3 |F\(4x4|passed|failed" ) > $O/robustness.txt 2>&1
