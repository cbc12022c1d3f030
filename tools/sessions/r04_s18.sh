#!/bin/bash
# round 4, session 18: proposal heads on packed FMAs (headvalu.hip): parity tests, A/B per head against the M = 4 MFMA kernel / the kw-folded GEMM
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s18; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "head" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
( timeout 600 python tools/bench_layers.py --ab flags=0,512,8192,8704 --only LFCN --iters 200 ) > $O/ab_heads.txt 2>&1
