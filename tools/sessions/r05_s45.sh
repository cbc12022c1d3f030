#!/bin/bash
# round 5, session 45: the other configs at HEAD (after the heads' fix-up change), GPU legs only (their CPU-reference / parity legs: r05_s34.sh, and the full-size tests of r05_s44.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s45; mkdir -p $O; export PYTHONUNBUFFERED=1
: > $O/models.jsonl
A="--steps 40 --warmup 8 --no-robust --no-cpu-baseline"
for m in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 200 python bench.py --model $m $A >> $O/models.jsonl 2>> $O/models.err
done
timeout 200 python bench.py --model caltech/mscnn-7s-480 --dtype f16 $A >> $O/models.jsonl 2>> $O/models.err
timeout 200 python bench.py --model caltech/mscnn-7s-480 --dtype f32 --batch 8 $A >> $O/models.jsonl 2>> $O/models.err
timeout 200 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --batch 8 $A >> $O/models.jsonl 2>> $O/models.err
timeout 200 python bench.py --batch 2 $A >> $O/models.jsonl 2>> $O/models.err
