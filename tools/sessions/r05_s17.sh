#!/bin/bash
# round 5, session 17: wave priority 3 for everything beside the MFMAs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s17; mkdir -p $O; export PYTHONUNBUFFERED=1
for i in 1 2 3; do ( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wf2conv" 2>&1 | tail -3 ) >> $O/tests.txt 2>&1; done
( timeout 300 python tools/bench_layers.py --only conv1_2 --iters 20 --pool only --ab flags=0,65536 2>&1 | grep conv1_2 ) >> $O/ab_conv1_2.txt 2>&1
timeout 120 python tools/wf2_trace.py > $O/trace.txt 2>&1
