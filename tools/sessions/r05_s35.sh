#!/bin/bash
# round 5, session 35: A/B nontemporal stores of conv1_1's output (283 MB, read back from HBM by conv1_2), stand-alone and in the net
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s35; mkdir -p $O; export PYTHONUNBUFFERED=1
for v in base nt base nt; do
  lib=""; pre=""; [ $v = nt ] && lib="--lib tools/micro/libmscnn_hip_c3nt.so" && pre="$GRAFT_REPO_ROOT/tools/micro/libmscnn_hip_c3nt.so"
  echo "== $v" >> $O/nt.txt
  ( timeout 120 python tools/bench_layers.py --only conv1_1 --iters 20 $lib 2>&1 | grep conv1_1 | cut -c1-120 ) >> $O/nt.txt
  LD_PRELOAD=$pre timeout 200 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline --layers 2> $O/layers.tmp | cut -c1-120 >> $O/nt.txt
  grep -E "^conv1_1|^conv1_2" $O/layers.tmp | cut -c1-120 >> $O/nt.txt
done
