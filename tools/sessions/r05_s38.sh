#!/bin/bash
# round 5, session 38: the proposal heads' stream-K split: workgroups x chunk size on every level (tune_grid now unclamped, tune_variant 501 = half chunks)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s38; mkdir -p $O; export PYTHONUNBUFFERED=1
B="timeout 120 python tools/bench_layers.py --iters 200"
{
$B --only LFCN_4_5x5 --ab grid=0,64
$B --only LFCN_4_5x5 --ab grid=64,128 --fixed variant=501
$B --only LFCN_3_5x5 --ab grid=0,256
$B --only LFCN_3_5x5 --ab grid=256,512 --fixed variant=501
$B --only LFCN_3_7x7 --ab grid=0,512
$B --only LFCN_3_7x7 --ab grid=512,1024 --fixed variant=501
$B --only LFCN_2_5x5 --ab grid=0,512,768
$B --only LFCN_2_5x5 --ab grid=512,768,1024,1536 --fixed variant=501
$B --only LFCN_2_7x7 --ab grid=0,768,1024,1536
$B --only LFCN_2_7x7 --ab grid=512,1024,1536 --fixed variant=501
$B --only LFCN_1_5x5 --ab grid=0,1024
$B --only LFCN_1_5x5 --ab grid=512,1024 --fixed variant=501
} > $O/heads.txt 2>&1
