#!/bin/bash
# round 4, session 40: the chain net test with the weight-change case
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s40; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_net.py -q -x -k "chains or default_flow" 2>&1 | tail -12 ) > $O/tests.txt 2>&1
