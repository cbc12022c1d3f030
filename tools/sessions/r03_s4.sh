#!/bin/bash
# round 3, GPU session 4: new Winograd GEMM kernel, stand-alone bench
cd $GRAFT_REPO_ROOT/tools/micro; O=$GRAFT_REPO_ROOT/gpurun_out/s4; mkdir -p $O
(
for v in 1 3 4; do timeout 60 ./wgemm_bench 25 512 512 1920 $v 20 1; done     # conv4_2 F3
timeout 60 ./wgemm_bench 25 512 512 1920 1 20 0 1     # no loads
timeout 60 ./wgemm_bench 25 512 512 1920 1 20 0 2     # no stores
timeout 60 ./wgemm_bench 25 512 512 1920 1 20 0 4     # no MFMA
timeout 60 ./wgemm_bench 25 512 512 1920 1 20 0 3     # MFMA + LDS reads only
timeout 60 ./wgemm_bench 36 512 512 1080 1 20 1       # conv4_2 F4
timeout 60 ./wgemm_bench 25 256 256 7680 1 20 1       # conv3_2 F3
timeout 60 ./wgemm_bench 25 256 256 7680 2 20 1
timeout 60 ./wgemm_bench 25 128 128 30720 2 20 1      # conv2_2 F3
timeout 60 ./wgemm_bench 25 512 512 400 1 20 1        # conv5 F3
timeout 60 ./wgemm_bench 25 512 1024 2800 1 20 1      # roi_c1
) > $O/wgemm.txt 2>&1
