#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s23; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "wgemm or kwfold or kernel_selection or ip_ or inner" 2>&1 | tail -8 ) > $O/ops.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_dist.py -q 2>&1 | tail -15 ) > $O/dist.txt 2>&1
( timeout 600 python bench.py --no-alt --no-robust --no-cpu-baseline 2>$O/bench.err | tail -1 ) > $O/bench.json
