#!/bin/bash
# round 4, session 6: fused ROI pooling on sliding-maximum maps -- stand-alone check + timing, op / net tests, wgemm A/Bs of s3, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s6; mkdir -p $O; export PYTHONUNBUFFERED=1
C=tools/micro/roipool_wino_check
{ for b in $C ${C}_dbg_pooled; do echo "=== $b"; timeout 60 $b 37 64 24 40 5; timeout 60 $b 133 128 36 120 5; timeout 120 $b 700 512 72 240 30; timeout 60 $b 5 64 20 28 3; done; } > $O/check.txt 2>&1
timeout 300 python tools/bench_roipool.py > $O/roipool.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -q -x -k "roipool or deferred or unfused or partial_forward or layerwise or dynamic_roi or default_flow or test_net_" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
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
