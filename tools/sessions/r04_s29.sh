#!/bin/bash
# round 4, session 29: conv1_2 pool-only without issuing the y stores; epilogue priority on the other users of the direct kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s29; mkdir -p $O; export PYTHONUNBUFFERED=1
{
echo "== conv1_2, y + pooled"; timeout 300 python tools/bench_layers.py --ab flags=0,16384 --only conv1_2 --iters 60 --pool both
echo "== conv1_2, pooled only"; timeout 300 python tools/bench_layers.py --ab flags=0,16384 --only conv1_2 --iters 60 --pool only
echo "== LFCN_1_7x7 (kw-folded GEMM on the 64x256 k7x1 kernel)"; timeout 300 python tools/bench_layers.py --ab flags=0,16384 --only LFCN_1_7x7 --iters 200
echo "== conv2_1 / conv4_2 direct"; timeout 300 python tools/bench_layers.py --ab flags=0,16384 --fixed algo=1 --only conv2_1 --iters 60
timeout 300 python tools/bench_layers.py --ab flags=0,16384 --fixed algo=1 --only conv4_2 --iters 60
} > $O/ab_prio.txt 2>&1
( timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "pool or igemm or direct" 2>&1 | tail -5 ) > $O/tests.txt 2>&1
