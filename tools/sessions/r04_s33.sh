#!/bin/bash
# round 4, session 33: conv1_2 on the ring kernel (wconv.hip): parity against the igemm kernel, A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s33; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "wconv" 2>&1 | tail -30 ) > $O/tests.txt 2>&1
( timeout 300 python tools/bench_layers.py --ab flags=0,32768 --only conv1_2 --iters 60 ) > $O/ab.txt 2>&1
( timeout 300 python tools/bench_layers.py --ab flags=0,32768 --only conv1_2 --iters 60 --pool only ) >> $O/ab.txt 2>&1
