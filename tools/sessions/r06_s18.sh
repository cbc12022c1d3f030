#!/bin/bash
# round 6, session 18: the plane GEMM's tile shape and split / whole scheduling per layer, every shape the kernel table has, interleaved on one box
# tune_variant = 300 + shape (1: 256x128, 2: 128x256, 3: 128x128, 4: 256x96, 5: 256x160) + 256 (force the stream-K split) | + 512 (force whole tiles); 0 = the plan's own choice
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s18; mkdir -p $O; export PYTHONUNBUFFERED=1
V="0,301,557,813,302,558,814,303,559,815,304,560,816,305,561,817"
: > $O/sweep.txt
for L in conv2_1 conv2_2 conv3_1 conv3_2 conv4_1 conv4_2 conv5_1 conv6_1 roi_c1; do
  timeout 300 python tools/bench_layers.py --only $L --iters 24 --ab variant=$V >> $O/sweep.txt 2>> $O/sweep.err
done
