#!/bin/bash
# round 5, session 33: A/B nontemporal V operand loads (LDS-DMA) in the plane GEMM
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s33; mkdir -p $O; export PYTHONUNBUFFERED=1
for v in base nt base nt; do
  pre=""; [ $v = nt ] && pre="$GRAFT_REPO_ROOT/tools/micro/libmscnn_hip_wgntv.so"
  echo "== $v" >> $O/nt.txt
  LD_PRELOAD=$pre timeout 200 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline --layers 2> $O/layers.tmp | cut -c1-160 >> $O/nt.txt
  grep -E "^conv2_1|^conv2_2|^conv3_1|^conv3_2|^conv3_3|^conv4_1|^conv4_2|^conv5_1|^roi_c1|^fc6" $O/layers.tmp | cut -c1-30,95-190 >> $O/nt.txt
done
