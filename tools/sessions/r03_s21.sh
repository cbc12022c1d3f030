#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s21; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python tools/bench_layers.py --ab flags=0,512 --only LFCN --iters 200 ) > $O/ab_heads.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "head" 2>&1 | tail -8 ) > $O/tests.txt 2>&1
