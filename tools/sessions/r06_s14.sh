#!/bin/bash
# round 6, session 14: the 1000-frame stream again (band-check scratch reserved once), the watch / calibration tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s14; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 300 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-robust --no-regimes --dump-steps > $O/bench_1000.json 2> $O/steps_1000.txt
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -x -k "watch or calibrat or default_flow or boundary or zero_warmup or chain or handoff" 2>&1 | tail -6 ) > $O/tests.txt 2>&1
