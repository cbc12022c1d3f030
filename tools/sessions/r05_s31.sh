#!/bin/bash
# round 5, session 30: the nontemporal choices as shipped (plain output transform always, chained kernel by M's size): chain / F(4x4) tests, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s31; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q -k "f4 or chain or wino or winograd or unfused" 2>&1 | tail -4 ) > $O/tests.txt 2>&1
timeout 300 python bench.py --steps 60 --warmup 10 --no-robust --layers > $O/bench.json 2> $O/layers.txt
