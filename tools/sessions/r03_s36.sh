#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s36; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python tools/debug/scan_debug.py 2>&1 | tail -30 ) > $O/dbg.txt 2>&1
