#!/bin/bash
# round 6, session 15: op test of the metric kernels / store_words
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s15; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "parity_metric or relu or pool_kat" 2>&1 | tail -15 ) > $O/tests.txt 2>&1
