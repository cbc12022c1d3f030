#!/bin/bash
# round 4, session 31: the whole GPU suite + smoke at HEAD
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s31; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/gpu_all.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
