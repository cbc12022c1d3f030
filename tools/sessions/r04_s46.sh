#!/bin/bash
# round 4, session 46: in-net A/B of conv1_2 on the igemm kernel (flags 0) vs the ring kernel of wconv.hip (flags 32768), same process, alternating
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s46; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 400 python tools/ab_net_layer.py --layer conv1_2 --flags 0,32768 --iters 100 --rounds 4 ) > $O/ab.txt 2>&1
