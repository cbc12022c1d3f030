#!/bin/bash
# round 3, GPU session 3: F(4x4,3x3) as the AUTO choice -- conv tests, bench, wg trace with shader clock, MFMA clock micro-benchmark
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s3; mkdir -p $O; export PYTHONUNBUFFERED=1
./tools/micro/mfma_clock > $O/mfma_clock.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "conv or wino" 2>&1 | tail -5 ) > $O/conv_tests.txt 2>&1
timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
( for L in conv4_2 conv3_2; do timeout 120 python tools/wg_trace.py --only $L --algo 3; done
  timeout 120 python tools/wg_trace.py --only conv4_2 --algo 3 --grid 256
  timeout 120 python tools/wg_trace.py --only conv4_2 --algo 3 --grid 750 ) > $O/trace.txt 2>&1
( timeout 300 python tools/bench_layers.py --ab algo=0,6 --only conv1_2 ) > $O/ab_conv1_2.txt 2>&1
echo done > $O/done
