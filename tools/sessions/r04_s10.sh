#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s10; mkdir -p $O; export PYTHONUNBUFFERED=1
{ timeout 60 tools/micro/roipool_wino_check 37 64 24 40 5; timeout 60 tools/micro/roipool_wino_check 133 128 36 120 5; timeout 60 tools/micro/roipool_wino_check 5 64 20 28 3; timeout 60 tools/micro/roipool_wino_check 700 512 72 240 20; } > $O/check.txt 2>&1
for b in roipool_wino_check roipool_wino_check_no_store roipool_wino_check_no_pool roipool_wino_check_no_store_no_pool; do echo "=== $b"; timeout 60 tools/micro/$b 676 512 72 240 50 1 | tail -1; done > $O/ablate.txt 2>&1
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_roipool.py > $GRAFT_REPO_ROOT/$O/roipool.txt 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/roipool_kernel_stats.csv; rm -rf $O/prof
for v in 0 513 257 515 259; do timeout 60 tools/micro/wgemm_bench 25 512 512 120 $v 300 1 | tail -1; done > $O/conv6.txt 2>&1
