#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s13; mkdir -p $O; export PYTHONUNBUFFERED=1
( cd tools/micro
timeout 60 ./wgemm_bench 3 256 96 384 257 2 1 0
timeout 60 ./wgemm_bench 5 512 64 1000 257 2 1 0
timeout 60 ./wgemm_bench 7 96 160 700 258 2 1 0
timeout 60 ./wgemm_bench 1 256 512 17280 257 200 1       # LFCN_1_5x5 as GEMM over taps (135 tiles)
timeout 60 ./wgemm_bench 1 512 512 17280 257 200 1       # LFCN_1_7x7 (270 tiles)
for v in 257 513; do
timeout 60 ./wgemm_bench 25 512 512 1920 $v 300 1       # conv4_2 F3
timeout 60 ./wgemm_bench 36 512 512 1080 $v 300 1       # conv4_2 F4
timeout 60 ./wgemm_bench 36 256 256 4320 $v 300 1       # conv3_2 F4
timeout 60 ./wgemm_bench 36 128 128 17280 $((v+1)) 300 1      # conv2_2 F4
timeout 60 ./wgemm_bench 25 512 512 400 $v 300 1        # conv5 F3
timeout 60 ./wgemm_bench 25 512 512 100 $v 300 1        # conv6_1 F3
timeout 60 ./wgemm_bench 25 512 1024 2800 $v 100 1      # roi_c1
done
) > $O/wgemm.txt 2>&1
