#!/bin/bash
# round 6, session 25: the whole GPU suite + smoke at the round's last commit
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s25; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 3000 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -16 ) > $O/gputests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gputests.txt 2>&1
