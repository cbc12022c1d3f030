#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/micro; O=$GRAFT_REPO_ROOT/gpurun_out/s8; mkdir -p $O
(
for it in 20 200 1000; do timeout 60 ./wgemm_bench 25 512 512 1920 1 $it 0; done
for it in 200 1000; do timeout 60 ./wgemm_bench 25 512 512 1920 1 $it 0 27; done
timeout 60 ./wgemm_bench 25 512 512 1920 1 1000 0 3
timeout 60 ./wgemm_bench 25 256 256 7680 1 500 0
timeout 60 ./wgemm_bench 36 512 512 1080 1 500 0
) > $O/wgemm.txt 2>&1
cd $GRAFT_REPO_ROOT
( timeout 300 python tools/bench_layers.py --ab algo=3 --only conv4_2 --iters 20; timeout 300 python tools/bench_layers.py --ab algo=3 --only conv4_2 --iters 400 ) > $O/layers_iters.txt 2>&1
