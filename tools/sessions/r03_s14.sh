#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s14; mkdir -p $O; export PYTHONUNBUFFERED=1
( cd tools/micro
for abl in 0 1 2 3; do timeout 60 ./wgemm_bench 36 128 128 17280 514 300 0 $abl; done      # conv2_2 F4, whole tiles
for abl in 0 1 2 3; do timeout 60 ./wgemm_bench 36 256 256 4320 513 300 0 $abl; done       # conv3_2 F4
) > $O/wgemm.txt 2>&1
( for L in conv2_ conv3_ conv4_ conv5_1 conv6_1 roi_c1; do timeout 300 python tools/bench_layers.py --ab flags=0,128 --only $L --iters 100; done ) > $O/ab_wgemm.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_net.py -q -x -s -k "caffemodel_file or vgg_like" 2>&1 | grep -E "FULLSIZE|passed|failed|Error|assert" | tail -15 ) > $O/new_tests.txt 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
