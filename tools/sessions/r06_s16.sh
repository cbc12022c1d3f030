#!/bin/bash
# round 6, session 16: the soak test of the host-side machinery over a stream of different frames
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s16; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -x -k "stream_of_frames" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
