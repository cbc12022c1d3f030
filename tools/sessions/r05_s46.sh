#!/bin/bash
# round 5, session 46: half chunks on the largest 5x5 head too?  (40 tiles: outside AUTO's <= 16-tile rule)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s46; mkdir -p $O; export PYTHONUNBUFFERED=1
B="timeout 120 python tools/bench_layers.py --iters 400"
{ $B --only LFCN_1_5x5 --ab variant=500,501; $B --only LFCN_1_5x5 --ab variant=501,500; } > $O/heads.txt 2>&1
