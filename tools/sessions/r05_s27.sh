#!/bin/bash
# round 5, session 27: waves 0-3 close the unit before their output transform (split barrier): op tests x 3, A/B, timeline, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s27; mkdir -p $O; export PYTHONUNBUFFERED=1
for i in 1 2 3; do ( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wf2conv" 2>&1 | tail -2 ) >> $O/tests.txt 2>&1; done
( timeout 300 python tools/bench_layers.py --only conv1_2 --iters 20 --pool only --ab flags=0,65536 2>&1 | grep conv1_2 ) >> $O/ab_conv1_2.txt 2>&1
timeout 120 python tools/wf2_trace.py > $O/trace.txt 2>&1
timeout 300 python bench.py --steps 60 --warmup 10 --no-robust > $O/bench.json 2> $O/bench.err
