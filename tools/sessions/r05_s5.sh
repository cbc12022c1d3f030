#!/bin/bash
# round 5, session 5: the one-launch Winograd F(2x2,3x3) kernel for conv1_2 (wf2conv.hip): op tests, per-layer A/B against the ring kernel, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s5; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wf2conv or wconv_ring" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
( timeout 300 python tools/bench_layers.py --only conv1_2 --iters 20 --ab flags=0,65536 2>&1 | tail -12 ) > $O/ab_conv1_2.txt 2>&1
timeout 300 python bench.py --steps 40 --warmup 10 --no-robust --layers > $O/bench.json 2> $O/bench_layers.txt
