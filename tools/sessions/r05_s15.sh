#!/bin/bash
# round 5, session 15: the net with conv1_2 on the one-launch Winograd kernel: bench (full-size parity against the reference inside), net tests that involve conv1_2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s15; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 300 python bench.py --steps 40 --warmup 10 --layers > $O/bench.json 2> $O/bench_layers.txt
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -k "unfused or chains or default_flow or full_size_parity_vs_reference or vgg_like or partial_forward or batch_n" 2>&1 | tail -15 ) > $O/tests.txt 2>&1
