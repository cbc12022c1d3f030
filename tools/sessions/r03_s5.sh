#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/micro; O=$GRAFT_REPO_ROOT/gpurun_out/s5; mkdir -p $O
(
./lds_dma_probe
timeout 60 ./wgemm_bench 25 512 512 1920 1 20 1 0
for abl in 3 11 19 27 4; do timeout 60 ./wgemm_bench 25 512 512 1920 1 20 0 $abl; done
timeout 60 ./wgemm_bench 2 256 64 128 1 2 1 0
) > $O/wgemm.txt 2>&1
