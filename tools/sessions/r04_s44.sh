#!/bin/bash
# round 4, session 44: BoxOutput phase timeline with the ballot fixed-point scan (trace build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s44; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python tools/bo_trace.py ) > $O/bo_trace.txt 2>&1
