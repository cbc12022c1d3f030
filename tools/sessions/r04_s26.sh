#!/bin/bash
# round 4, session 26: conv1_2's direct kernel compiled for three workgroups per CU (variant 4, tune_variant 5) against the default / the pipelined-read build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s26; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python tools/bench_layers.py --ab variant=0,2,5 --only conv1_2 --iters 60 ) > $O/ab_conv1_2.txt 2>&1
( timeout 600 python tools/bench_layers.py --ab variant=0,5 --only conv1_2 --iters 60 ) >> $O/ab_conv1_2.txt 2>&1
