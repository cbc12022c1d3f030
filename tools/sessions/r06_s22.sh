#!/bin/bash
# round 6, session 22: per-layer table of config 5 in the fp16 mode at batch 8 and batch 1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s22; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --batch 8 --steps 40 --warmup 8 --no-robust --no-cpu-baseline --no-regimes --layers > $O/b8.json 2> $O/b8_layers.txt
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --batch 1 --steps 40 --warmup 8 --no-robust --no-cpu-baseline --no-regimes --layers > $O/b1.json 2> $O/b1_layers.txt
