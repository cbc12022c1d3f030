#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s16; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "f4 or wino_f4 or fused_pool" 2>&1 | tail -6 ) > $O/tests.txt 2>&1
( for L in conv2_ conv3_2 conv4_2; do timeout 300 python tools/bench_layers.py --ab flags=0,256 --only $L --iters 100; done ) > $O/ab_f4_transforms.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 10 --no-alt --no-robust > $O/bench.json 2> $O/bench.err
