#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s15; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_net.py -q -x -s -k "caffemodel_file or vgg_like" 2>&1 | grep -E "FULLSIZE|passed|failed|Error|assert" | tail -15 ) > $O/new_tests.txt 2>&1
timeout 600 python bench.py --steps 30 --warmup 10 --no-alt > $O/bench.json 2> $O/bench.err
