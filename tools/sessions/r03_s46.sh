#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s46; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python tools/bo_trace.py 2>&1 | grep -v amdgpu | tail -8 ) > $O/bo_trace.txt 2>&1
