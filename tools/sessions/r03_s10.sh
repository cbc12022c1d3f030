#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s10; mkdir -p $O; export PYTHONUNBUFFERED=1
( cd tools/micro
for v in 257 513; do timeout 60 ./wgemm_bench 25 512 512 100 $v 300 1; timeout 60 ./wgemm_bench 25 512 512 400 $v 300 1; timeout 60 ./wgemm_bench 36 512 512 1080 $v 300 1; done
) > $O/wgemm.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv or wino" 2>&1 | tail -5 ) > $O/conv_tests.txt 2>&1
( for L in conv2_ conv3_ conv4_ conv5_1 conv6_1 roi_c1; do timeout 300 python tools/bench_layers.py --ab flags=0,128 --only $L --iters 100; done ) > $O/ab_wgemm.txt 2>&1
timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
