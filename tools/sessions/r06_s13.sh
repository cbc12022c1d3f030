#!/bin/bash
# round 6, session 13: a 1000-frame stream at HEAD: the watch frames' cost by layer
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s13; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 300 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-robust --no-regimes --dump-steps > $O/bench_1000.json 2> $O/steps_1000.txt
