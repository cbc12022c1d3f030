#!/bin/bash
# round 6, session 19: full-size parity at R = 2000 as a test, the final stage above 4032 ROIs through the Net
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s19; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -x -s -k "above_4032 or (full_size_parity_vs_reference and size3) or x3_vs_reference" 2>&1 | grep -E "FULLSIZE|passed|failed|Error|assert" | tail -12 ) > $O/tests.txt 2>&1
