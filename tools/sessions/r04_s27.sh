#!/bin/bash
# round 4, session 27: per-workgroup phase timeline of conv1_2's direct kernel (trace build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s27; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python tools/wg_trace.py --only conv1_2 --algo 1 ) > $O/wg_trace_conv1_2.txt 2>&1
