#!/bin/bash
# round 6, session 1: the evidence items of verdict r5 #1 (dense regime at full size, regimes / dbox_max / subnet error in the line) and the
# stream-K contributor list fix
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s1; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "stream_k or test_conv" 2>&1 | tail -8 ) > $O/tests.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -k "test_full_size_parity_vs_reference" --durations=8 -s 2>&1 | grep -E "FULLSIZE|passed|failed|Error|assert" | tail -20 ) >> $O/tests.txt 2>&1
timeout 400 python bench.py --layers > $O/bench.json 2> $O/layers.txt
