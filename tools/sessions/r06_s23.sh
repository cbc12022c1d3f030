#!/bin/bash
# round 6, session 23: every BASELINE config at HEAD as the driver would run it (the default line incl. the reference leg and its parity), fp16 rows without the reference leg
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s23; mkdir -p $O; export PYTHONUNBUFFERED=1
: > $O/models.jsonl
for m in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 600 python bench.py --model $m --steps 60 --warmup 10 --no-robust >> $O/models.jsonl 2>> $O/models.err
done
A="--steps 60 --warmup 10 --no-robust --no-regimes"
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 $A >> $O/models.jsonl 2>> $O/models.err
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --batch 8 --no-cpu-baseline $A >> $O/models.jsonl 2>> $O/models.err
timeout 300 python bench.py --model caltech/mscnn-7s-480 --batch 8 --no-cpu-baseline $A >> $O/models.jsonl 2>> $O/models.err
timeout 300 python bench.py --batch 2 --no-cpu-baseline $A >> $O/models.jsonl 2>> $O/models.err
python tools/models_table.py $O/models.jsonl > $O/models.txt 2>&1
