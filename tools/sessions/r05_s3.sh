#!/bin/bash
# round 5, session 3: whole-net batch N against the reference + the input-reshape entry; every BASELINE config re-run at HEAD (fp32 and config 5 in fp16); caltech vs batch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s3; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q -k "batch_n or input_reshape" 2>&1 | tail -30 ) > $O/tests.txt 2>&1
: > $O/models.jsonl
for m in kitti_car/mscnn-7s-576 kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 300 python bench.py --model $m --steps 30 --warmup 8 --no-robust >> $O/models.jsonl 2>> $O/models.err
done
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --steps 30 --warmup 8 >> $O/models.jsonl 2>> $O/models.err
: > $O/batch.jsonl
for dt in f32 f16; do for b in 2 4 8; do
  timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype $dt --batch $b --steps 30 --warmup 8 >> $O/batch.jsonl 2>> $O/batch.err
done; done
timeout 300 python bench.py --batch 2 --steps 20 --warmup 5 >> $O/batch.jsonl 2>> $O/batch.err
python tools/models_table.py $O/models.jsonl $O/batch.jsonl > $O/models.txt 2>&1
