#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s55; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 400 python tools/bench_layers.py --only LFCN --ab grid=0,256,384,768,1024 --iters 100 2>&1 | grep -v amdgpu | tail -40 ) > $O/ab_heads_grid.txt 2>&1
