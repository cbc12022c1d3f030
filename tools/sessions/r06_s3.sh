#!/bin/bash
# round 6, session 3: the deferred band form of the numerics watch: its tests, a 1000-step stream with per-step times, the default line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s3; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -x -k "watch or calibrat or handoff or default_flow or boundary or zero_warmup or chain" 2>&1 | tail -15 ) > $O/tests.txt 2>&1
timeout 300 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-robust --no-regimes --dump-steps > $O/bench_1000.json 2> $O/steps_1000.txt
timeout 400 python bench.py --layers > $O/bench.json 2> $O/layers.txt
