#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s22; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -m "gpu and not slow" 2>&1 | tail -6 ) > $O/ops.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_dist.py tests/test_golden.py -q -x -m "gpu and not slow" 2>&1 | tail -6 ) > $O/net.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
