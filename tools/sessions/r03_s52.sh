#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s52; mkdir -p $O; export PYTHONUNBUFFERED=1
cd tools/micro
( for args in "25 512 512 480 4 200 1" "25 512 512 480 1 200 1" "25 512 512 480 260 200 1" "3 256 96 500 4 50 1" "36 512 512 270 4 200 1" "25 512 512 864 4 200 1" "25 512 512 864 1 200 1" "7 128 160 70 4 50 1"; do timeout 120 ./wgemm_bench $args 2>&1 | tail -3; done ) > ../../$O/wgemm_bench.txt 2>&1
cd ../..
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "wgemm or winograd or kernel_selection" 2>&1 | tail -6 ) > $O/ops.txt 2>&1
( timeout 600 python bench.py --no-alt --no-robust --no-cpu-baseline --layers 2>$O/bench.err | tail -1 ) > $O/bench.json
