#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s50; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python tools/bench_layers.py --only conv1_2 --ab variant=0,1 --iters 60 2>&1 | tail -6 ) > $O/ab_variant.txt 2>&1
( timeout 300 python tools/bench_layers.py --only conv1_2 --ab grid=0,256,512,768,1024 --iters 60 2>&1 | tail -8 ) > $O/ab_grid.txt 2>&1
( timeout 300 python tools/bench_layers.py --only conv1_2 --ab grid=0,512,768 --fixed variant=1 --iters 60 2>&1 | tail -8 ) > $O/ab_grid_pf.txt 2>&1
