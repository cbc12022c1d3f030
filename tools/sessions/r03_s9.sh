#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/micro; O=$GRAFT_REPO_ROOT/gpurun_out/s9; mkdir -p $O
(
timeout 60 ./wgemm_bench 3 256 96 384 1 2 1 0           # 9 tiles on 256 WGs: all stream-K
timeout 60 ./wgemm_bench 5 512 64 1000 1 2 1 0
timeout 60 ./wgemm_bench 7 96 160 700 2 2 1 0
timeout 60 ./wgemm_bench 25 512 512 1920 1 300 1       # conv4_2 F3
timeout 60 ./wgemm_bench 36 512 512 1080 1 300 1       # conv4_2 F4
timeout 60 ./wgemm_bench 36 256 256 4320 1 300 1       # conv3_2 F4
timeout 60 ./wgemm_bench 25 256 256 7680 1 300 1       # conv3_2 F3
timeout 60 ./wgemm_bench 36 256 128 4320 1 300 1       # conv3_1 F4
timeout 60 ./wgemm_bench 36 512 256 1080 1 300 1       # conv4_1 F4
timeout 60 ./wgemm_bench 36 128 128 17280 2 300 1      # conv2_2 F4
timeout 60 ./wgemm_bench 36 128 64 17280 2 300 1       # conv2_1 F4
timeout 60 ./wgemm_bench 25 512 512 400 1 300 1        # conv5 F3
timeout 60 ./wgemm_bench 36 512 512 270 1 300 1        # conv5 F4
timeout 60 ./wgemm_bench 25 512 512 100 1 300 1        # conv6_1 F3
timeout 60 ./wgemm_bench 25 512 1024 2800 1 100 1      # roi_c1
) > $O/wgemm.txt 2>&1
