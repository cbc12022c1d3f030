#!/bin/bash
# round 4, session 28: conv1_2's epilogue at wave priority 3 (tune_flags bit 14): A/B + phase timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s28; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python tools/bench_layers.py --ab flags=0,16384 --only conv1_2 --iters 60 ) > $O/ab_conv1_2_prio.txt 2>&1
( timeout 600 python tools/bench_layers.py --ab flags=0,16384 --only conv1_2 --iters 60 ) >> $O/ab_conv1_2_prio.txt 2>&1
( timeout 300 python tools/wg_trace.py --only conv1_2 --algo 1 --flags 16384 ) > $O/wg_trace_conv1_2_prio.txt 2>&1
