#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/micro; O=$GRAFT_REPO_ROOT/gpurun_out/s7; mkdir -p $O
(
timeout 60 ./wgemm_bench 1 256 32 128 1 2 1 0
timeout 60 ./wgemm_bench 3 256 96 384 1 2 1 0
for v in 1 3; do timeout 60 ./wgemm_bench 25 512 512 1920 $v 20 1; done     # conv4_2 F3
for abl in 1 2 3 4 11 19 27; do timeout 60 ./wgemm_bench 25 512 512 1920 1 20 0 $abl; done
timeout 60 ./wgemm_bench 36 512 512 1080 1 20 1       # conv4_2 F4
timeout 60 ./wgemm_bench 25 256 256 7680 1 20 1       # conv3_2 F3
timeout 60 ./wgemm_bench 25 256 256 7680 2 20 1
timeout 60 ./wgemm_bench 25 128 128 30720 2 20 1      # conv2_2 F3
timeout 60 ./wgemm_bench 25 128 128 30720 3 20 1
timeout 60 ./wgemm_bench 25 512 512 400 1 20 1        # conv5 F3
timeout 60 ./wgemm_bench 25 512 1024 2800 1 20 1      # roi_c1
) > $O/wgemm.txt 2>&1
