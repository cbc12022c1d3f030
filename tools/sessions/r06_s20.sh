#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s20; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q -x -k "above_4032" 2>&1 | tail -12 ) > $O/tests.txt 2>&1
