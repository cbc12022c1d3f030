#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/micro; O=$GRAFT_REPO_ROOT/gpurun_out/s6; mkdir -p $O
( timeout 60 ./wgemm_bench 1 256 32 128 1 2 1 0 ) > $O/probe.txt 2>&1
