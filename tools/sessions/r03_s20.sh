#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s20; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "inner_product or f4 or wino" 2>&1 | tail -4 ) > $O/tests.txt 2>&1
timeout 600 python bench.py --steps 50 --warmup 10 --no-alt --no-robust --layers > $O/bench.json 2> $O/bench_layers.txt
