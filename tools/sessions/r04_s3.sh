#!/bin/bash
# round 4, session 3: ROI pooling fused into roi_c1's input stage (op + net tests, timing), 256 x 160 wgemm tile stand-alone,
# whole tiles vs stream-K split per shape (re-fit of the plan's model), in-net bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s3; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -q -x -k "roipool or wgemm or deferred or unfused or partial_forward or layerwise or dynamic_roi or default_flow or numerics_watch or test_net_" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
timeout 300 python tools/bench_roipool.py > $O/roipool.txt 2>&1
N=tools/micro/wgemm_bench
run() { timeout 120 "$@" 2>&1 | grep -v "^ok map\|^row " ; }
{
echo "=== 256x160 (5), forced split (261), forced whole (517) vs 256x128 auto (1)"
for sh in "36 512 512 1080" "36 512 256 1080"; do for v in 1 5 261 517; do run $N $sh $v 200 1; done; done
for sh in "36 256 256 4320" "25 512 1024 2784" "25 512 512 480"; do for v in 0 5; do run $N $sh $v 100 1; done; done
echo "=== whole (512 +) vs split (256 +)"
for sh in "36 256 256 4320" "36 256 128 4320" "25 512 1024 2784"; do for v in 513 257; do run $N $sh $v 200 0; done; done
for sh in "36 128 128 17280" "36 128 64 17280"; do for v in 514 258; do run $N $sh $v 200 0; done; done
for v in 516 260 513 257 517 261; do run $N 25 512 512 480 $v 300 0; done
for v in 513 257 516 260 517 261 515 259; do run $N 25 512 512 120 $v 300 1; done
} > $O/wgemm_ab.txt 2>&1
timeout 600 python bench.py --layers --no-alt --no-robust > $O/bench.json 2> $O/bench_layers.txt
