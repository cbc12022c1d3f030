#!/bin/bash
# round 4, session 37: wconv (opt-in, whole tiles) parity; conv op tests around it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s37; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "wconv or pool or c3 or direct or igemm" 2>&1 | tail -8 ) > $O/tests.txt 2>&1
