#!/bin/bash
# round 6, session 12: the host-placement test on the box's own topology, the dist tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s12; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -15 ) > $O/tests.txt 2>&1
