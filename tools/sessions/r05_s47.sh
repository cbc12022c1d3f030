#!/bin/bash
# round 5, session 47: half chunks for the 5-row head kernels on every map: head tests, full-size parity of every net against the reference, the frame, the other configs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s47; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "head" 2>&1 | tail -3 ) > $O/tests.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q -k "test_full_size_parity_vs_reference or test_full_size_f16_caltech or whole_net_batch or caffe_net_small" --durations=5 2>&1 | tail -12 ) >> $O/tests.txt 2>&1
timeout 300 python bench.py --layers > $O/bench_final.json 2> $O/layers_final.txt
: > $O/models.jsonl
A="--steps 40 --warmup 8 --no-robust --no-cpu-baseline"
for m in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 200 python bench.py --model $m $A >> $O/models.jsonl 2>> $O/models.err
done
timeout 200 python bench.py --model caltech/mscnn-7s-480 --dtype f16 $A >> $O/models.jsonl 2>> $O/models.err
