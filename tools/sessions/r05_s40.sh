#!/bin/bash
# round 5, session 40: the heads' wide fix-up (8 slab groups x 32 lanes per tile quarter), quarter chunks (tune_variant 502), the igemm fix-ups' parallel list
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s40; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "head or stream_k or igemm or splitk or fixup" 2>&1 | tail -5 ) > $O/tests.txt 2>&1
B="timeout 120 python tools/bench_layers.py --iters 200"
{
$B --only LFCN_4_5x5 --ab grid=0,64
$B --only LFCN_4_5x5 --ab grid=64,128 --fixed variant=501
$B --only LFCN_4_5x5 --ab grid=128,256 --fixed variant=502
$B --only LFCN_3_5x5 --ab grid=0,256
$B --only LFCN_3_5x5 --ab grid=256,512 --fixed variant=501
$B --only LFCN_3_5x5 --ab grid=512,1024 --fixed variant=502
$B --only LFCN_3_7x7 --ab grid=0,512
$B --only LFCN_3_7x7 --ab grid=512,1024 --fixed variant=501
$B --only LFCN_2_5x5 --ab grid=0,256,768
$B --only LFCN_2_5x5 --ab grid=512,768 --fixed variant=501
$B --only LFCN_2_5x5 --ab grid=512,1024 --fixed variant=502
$B --only LFCN_2_7x7 --ab grid=0,256,768
$B --only LFCN_2_7x7 --ab grid=512,768,1024 --fixed variant=501
$B --only LFCN_1_5x5 --ab grid=0,256
$B --only LFCN_1_7x7
} > $O/heads.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python tools/bench_layers.py --iters 100 --only LFCN_ --ab grid=0 > /dev/null 2> $O/prof.err
find $O/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace.csv; rm -rf $O/stats
