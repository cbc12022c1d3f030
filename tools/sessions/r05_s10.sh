#!/bin/bash
# round 5, session 10: + the two waves of a SIMD in opposite order for the stores / loads too, epilogue of waves 0-3 deferred
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s10; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wf2conv" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
( timeout 300 python tools/bench_layers.py --only conv1_2 --iters 20 --ab flags=0,65536 2>&1 | grep conv1_2 ) > $O/ab_conv1_2.txt 2>&1
( timeout 300 python tools/bench_layers.py --only conv1_2 --iters 20 --pool only --ab flags=0,65536 2>&1 | grep conv1_2 ) >> $O/ab_conv1_2.txt 2>&1
