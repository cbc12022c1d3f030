#!/bin/bash
# round 4, session 2: wgemm with two accumulator sets (no copy of the finished tile) and the 256 x 160 tile, same-box A/B against the
# round-3 kernel (tools/micro/wgemm_bench_r3 = the harness built from commit 98b29bd); launch cost of a wgemm-shaped grid; in-net bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s2; mkdir -p $O; export PYTHONUNBUFFERED=1
tools/micro/launch_gap > $O/launch_gap.txt 2>&1
N=tools/micro/wgemm_bench; R=tools/micro/wgemm_bench_r3
run() { timeout 120 "$@" 2>&1 | grep -v "^ok map\|^row " ; }
{
for sh in "36 128 64 17280" "36 128 128 17280" "36 256 128 4320" "36 256 256 4320" "36 512 256 1080" "36 512 512 1080" "25 512 512 480" "25 512 1024 2784" "25 512 512 120"; do
  echo "=== $sh"
  run $R $sh 0 200 0; run $N $sh 0 200 1; run $R $sh 0 200 0; run $N $sh 0 200 0
done
echo "=== 256x160 forced: whole / split, conv4 shapes + ragged"
for v in 5 517 261; do run $N 36 512 512 1080 $v 100 1; run $N 36 512 256 1080 $v 100 1; done
run $N 25 512 1024 2784 5 100 1; run $N 36 256 256 4320 5 100 1; run $N 7 288 96 333 261 20 1; run $N 3 512 64 1000 517 20 1
} > $O/wgemm_ab.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "wgemm or winograd" 2>&1 | tail -5 ) > $O/tests.txt 2>&1
timeout 600 python bench.py --layers --no-alt --no-robust > $O/bench.json 2> $O/bench_layers.txt
