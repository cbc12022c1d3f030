#!/bin/bash
# round 4, session 15: same-box A/B of the side-stream prefetch of the ROI-pooling maps (MSCNN_NO_PREFETCH=1 = off), 200 timed steps each, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s15; mkdir -p $O; export PYTHONUNBUFFERED=1
for i in 1 2 3; do
  MSCNN_NO_PREFETCH=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt --no-robust 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('prefetch off', d['value'], d['step_ms'])"
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-alt --no-robust 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('prefetch on ', d['value'], d['step_ms'])"
done > $O/ab_prefetch.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_net.py -q -x -k "deferred or unfused or partial_forward or layerwise or dynamic_roi or test_net_" 2>&1 | tail -3 ) > $O/tests.txt 2>&1
