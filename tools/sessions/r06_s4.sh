#!/bin/bash
# round 6, session 4: the whole GPU suite at HEAD (watch as a small deferred band, hand-off restart, placement), then a 1000-step stream
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s4; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 300 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-robust --no-regimes --dump-steps > $O/bench_1000.json 2> $O/steps_1000.txt
( timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 ) > $O/gputests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gputests.txt 2>&1
