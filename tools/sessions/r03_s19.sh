#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/micro; O=$GRAFT_REPO_ROOT/gpurun_out/s19; mkdir -p $O
(
for v in 1 5 6 7 8; do timeout 60 ./wgemm_bench 36 256 256 4320 $((v+512)) 300 1; done      # conv3_2 F4, whole tiles
for v in 1 5 6 7 8; do timeout 60 ./wgemm_bench 25 512 512 1920 $((v+512)) 300 0; done      # conv4_2 F3, whole tiles
for v in 1 5 6 7 8; do timeout 60 ./wgemm_bench 36 512 512 1080 $((v+256)) 300 1; done      # conv4_2 F4, split
for v in 1 5 6 7 8; do timeout 60 ./wgemm_bench 36 256 128 4320 $((v+512)) 300 0; done      # conv3_1 F4 (K=128)
) > $O/wgemm.txt 2>&1
