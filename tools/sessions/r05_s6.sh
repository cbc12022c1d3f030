#!/bin/bash
# round 5, session 6: ablations of the one-launch Winograd kernel (bit 0 no MFMAs, 1 no LDS operand reads, 2 no transform, 3 no global loads, 4 no barrier)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s6; mkdir -p $O; export PYTHONUNBUFFERED=1
for n in 0 1 2 4 8 16 14 30; do
  lib=""; [ $n != 0 ] && lib="--lib tools/micro/libmscnn_hip_wf2abl$n.so"
  echo "== abl $n" >> $O/abl.txt
  ( timeout 120 python tools/bench_layers.py --only conv1_2 --iters 10 $lib 2>&1 | grep conv1_2 ) >> $O/abl.txt 2>&1
done
