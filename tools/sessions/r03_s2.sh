#!/bin/bash
# round 3, GPU session 2: per-workgroup phase timeline of the Winograd GEMM (debug library)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s2; mkdir -p $O; export PYTHONUNBUFFERED=1
( for L in conv4_2 conv3_2 conv5_1 conv2_2; do timeout 120 python tools/wg_trace.py --only $L --algo 3; done
  timeout 120 python tools/wg_trace.py --only conv4_2 --algo 6
  timeout 120 python tools/wg_trace.py --only conv4_2 --algo 3 --grid 512
  timeout 120 python tools/wg_trace.py --only conv4_2 --algo 3 --grid 256
  timeout 120 python tools/wg_trace.py --only conv1_2 ) > $O/trace.txt 2>&1
echo done > $O/done
