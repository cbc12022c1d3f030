#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s7; mkdir -p $O; export PYTHONUNBUFFERED=1
C=tools/micro/roipool_wino_check
{ for b in $C ${C}_dbg_pooled; do echo "=== $b"; timeout 60 $b 37 64 24 40 5; timeout 60 $b 133 128 36 120 5; timeout 120 $b 700 512 72 240 30; timeout 60 $b 5 64 20 28 3; done; } > $O/check.txt 2>&1
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_roipool.py > $GRAFT_REPO_ROOT/$O/roipool.txt 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/roipool_kernel_stats.csv; rm -rf $O/prof
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -q -x -k "roipool or deferred or unfused or partial_forward or layerwise or dynamic_roi or default_flow or test_net_" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
