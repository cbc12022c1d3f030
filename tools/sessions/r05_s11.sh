#!/bin/bash
# round 5, session 11: phase timeline of the one-launch Winograd kernel (trace build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s11; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 120 python tools/wf2_trace.py > $O/trace.txt 2>&1
