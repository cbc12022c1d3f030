#!/bin/bash
# round 4, session 1: the CPU-side work on hardware (self-launch refusal, safe-by-default numerics, ADVICE fixes), a baseline bench at
# HEAD, then the wgemm schedule A/B (spread flush FP / LATE wait) on the eight GEMM shapes of the 7s-576 net
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s1; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_net.py tests/test_gpu_ops.py -q -x -k "test_gpu_dist or default_flow or numerical_calibration or numerics_watch or unfused or caffemodel_file or publish_amax or wgemm or winograd_f4x4 or reference_style" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
timeout 600 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
B=tools/micro/wgemm_bench
run() { timeout 120 $B "$@" 2>&1 | grep -v "^ok map\|^row " ; }
{
for v in 2 18 66 50 82; do run 36 128 64 17280 $v 200 1; done
for v in 2 18 66 50 82; do run 36 128 128 17280 $v 200 1; done
for v in 1 17 33 49 65 97 113; do run 36 256 128 4320 $v 200 1; done
for v in 1 17 33 49 65 97 113; do run 36 256 256 4320 $v 200 1; done
for v in 1 17 33 49 65 97 113; do run 36 512 256 1080 $v 200 1; done
for v in 1 17 33 49 65 97 113; do run 36 512 512 1080 $v 200 1; done
for v in 4 20 52 1 17 49; do run 25 512 512 480 $v 300 1; done
for v in 1 17 33 49 97 113; do run 25 512 1024 2784 $v 100 1; done
for v in 1 17 49; do run 25 512 512 120 $v 300 1; done
} > $O/wgemm_ab.txt 2>&1
