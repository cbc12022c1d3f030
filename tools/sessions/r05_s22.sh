#!/bin/bash
# round 5, session 22: where the 50 ms step of the batched caltech stream comes from
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s22; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 200 python tools/debug/batch_spike.py 4 > $O/spike4.txt 2>&1
