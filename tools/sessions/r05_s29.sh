#!/bin/bash
# round 5, session 29: nontemporal bits one by one (product library = bit 1; dev libraries 0 = none, 3 = + outin M loads, 5 = + outin V stores), per-layer stage times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s29; mkdir -p $O; export PYTHONUNBUFFERED=1
for v in 0 1 3 5 0 1; do
  pre=""; [ $v != 1 ] && pre="$GRAFT_REPO_ROOT/tools/micro/libmscnn_hip_nt$v.so"
  LD_PRELOAD=$pre timeout 200 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline --layers 2> $O/layers.tmp | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WINO_NT=$v', r['value'], r['ms_per_step'], r['winograd_stage_ms'])" >> $O/nt.txt
  grep -E "^conv2_1|^conv2_2|^conv3_2|^conv3_3|^conv4_2|^conv4_3|^loss1" $O/layers.tmp | cut -c1-30,95-190 >> $O/nt.txt
done
