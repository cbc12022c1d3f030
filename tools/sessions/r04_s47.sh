#!/bin/bash
# round 4, session 47: conv1_2 on the ring kernel by default: op tests, the full-size chain / default-flow net tests, the default bench (full-size parity vs the reference inside)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s47; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "wconv or pool_only or kernel_selection or c3" 2>&1 | tail -5 ) > $O/tests_ops.txt 2>&1
( timeout 400 python -m pytest tests/test_gpu_net.py -m gpu -q -k "chains or default_flow or unfused" 2>&1 | tail -5 ) > $O/tests_net.txt 2>&1
timeout 600 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
