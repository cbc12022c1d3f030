#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s56; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_dist.py -q 2>&1 | tail -5 ) > $O/dist.txt 2>&1
( timeout 120 mscnn_amd/detect_multi_gpu /dev/null 2>&1 | tail -2 ) > $O/usage.txt 2>&1
