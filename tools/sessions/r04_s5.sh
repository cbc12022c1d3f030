#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s5; mkdir -p $O
for b in roipool_wino_check roipool_wino_check_dbg_pooled roipool_wino_check_no_launder roipool_wino_check_store_b32; do echo "=== $b"; timeout 60 tools/micro/$b 37 64 24 40 3; timeout 60 tools/micro/$b 9 64 20 28 3; done > $O/check.txt 2>&1
