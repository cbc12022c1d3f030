#!/bin/bash
# round 6, session 17: the whole GPU suite + smoke + the default line at the round's last product change (band-check scratch reserved once; soak / metric tests added)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s17; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 3000 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -18 ) > $O/gputests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gputests.txt 2>&1
( time timeout 500 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt
