#!/bin/bash
# round 4, session 35: wconv parity (igemm reference on whole tiles), schedule A/B, in-net bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s35; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "wconv" 2>&1 | tail -30 ) > $O/tests.txt 2>&1
( timeout 300 python tools/bench_layers.py --ab variant=0,401,402 --only conv1_2 --iters 60 --pool only ) > $O/ab.txt 2>&1
for i in 1 2; do
  ( timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-robust ) > $O/bench_wconv_$i.json 2> $O/bench_wconv_$i.err
  ( MSCNN_TUNE_FLAGS=32768 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt --no-robust ) > $O/bench_igemm_$i.json 2> $O/bench_igemm_$i.err
done
