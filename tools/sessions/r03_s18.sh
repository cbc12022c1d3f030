#!/bin/bash
# round 3: the other BASELINE configurations through bench.py (the driver-runnable commands) + headline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s18; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 600 python bench.py --steps 50 --warmup 10 --layers > $O/bench.json 2> $O/bench_layers.txt
for M in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 600 python bench.py --model $M --steps 30 --warmup 8 --no-robust > $O/bench_$(basename $M).json 2> $O/bench_$(basename $M).err
done
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --steps 30 --warmup 8 --no-robust > $O/bench_caltech_f16.json 2> $O/bench_caltech_f16.err
