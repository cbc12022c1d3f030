#!/bin/bash
# round 5, session 26: smoke() with the one-launch Winograd kernel in it, its op tests once more after the plan-time CU count
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s26; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "wf2conv or wconv" 2>&1 | tail -3 ) >> $O/smoke.txt 2>&1
