#!/bin/bash
# round 5, session 41: heads with AUTO half chunks on maps of <= 16 tiles + the wide fix-up; BoxOutput writing its tops; tests, per-head A/B, the frame
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s41; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q -x -k "head or stream_k or boxoutput or inner_product or caffe_net_small or whole_net_batch or zero_warmup or handoff" 2>&1 | tail -5 ) > $O/tests.txt 2>&1
B="timeout 120 python tools/bench_layers.py --iters 200"
{
for L in LFCN_4_5x5 LFCN_3_5x5 LFCN_3_7x7 LFCN_2_5x5 LFCN_2_7x7 LFCN_1_5x5; do $B --only $L --ab variant=0,500; done
} > $O/heads.txt 2>&1
timeout 200 python bench.py --steps 100 --warmup 10 --no-robust --no-cpu-baseline --layers > $O/bench.json 2> $O/layers.txt
MSCNN_TUNE_VARIANT=500 timeout 200 python bench.py --steps 100 --warmup 10 --no-robust --no-cpu-baseline > $O/bench_full_chunks.json 2> /dev/null
timeout 200 python bench.py --steps 100 --warmup 10 --no-robust --no-cpu-baseline > $O/bench2.json 2> /dev/null
