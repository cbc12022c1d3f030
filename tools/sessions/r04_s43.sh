#!/bin/bash
# round 4, session 43: greedy NMS scan with the chunk's kept set as a ballot fixed point: index-exact tests, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s43; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_golden.py -m gpu -q -k "nms or boxoutput or detection or cascade or decode" 2>&1 | tail -8 ) > $O/tests_ops.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -k "layerwise or cascade" 2>&1 | tail -6 ) > $O/tests_net.txt 2>&1
( timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-alt --no-robust --layers ) > $O/bench.json 2> $O/bench_layers.txt
