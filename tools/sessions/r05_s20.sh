#!/bin/bash
# round 5, session 20: the whole GPU suite at HEAD (as the driver runs it) + smoke
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s20; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > $O/gputests.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) >> $O/gputests.txt 2>&1
