#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s9; mkdir -p $O; export PYTHONUNBUFFERED=1
for b in roipool_wino_check roipool_wino_check_no_store roipool_wino_check_no_pool roipool_wino_check_no_store_no_pool; do echo "=== $b"; timeout 60 tools/micro/$b 676 512 72 240 50 1 | tail -1; done > $O/ablate.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -q -x -k "roipool or deferred or unfused or partial_forward or layerwise or dynamic_roi or default_flow or test_net_" 2>&1 | tail -5 ) > $O/tests.txt 2>&1
timeout 600 python bench.py --layers --no-alt --no-robust > $O/bench.json 2> $O/bench_layers.txt
