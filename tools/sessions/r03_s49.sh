#!/bin/bash
# round 3, final measurements at HEAD: headline (all legs), per-layer table, rocprofv3 kernel stats of the fp32 loop, the other
# BASELINE configurations, BoxOutput phase timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s49; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 900 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-robust > $O/bench_prof.json 2> $O/bench_prof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
for M in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 600 python bench.py --model $M --steps 30 --warmup 8 --no-robust > $O/bench_$(basename $M).json 2> $O/bench_$(basename $M).err
done
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --steps 30 --warmup 8 --no-robust > $O/bench_caltech_f16.json 2> $O/bench_caltech_f16.err
( timeout 300 python tools/bo_trace.py 2>&1 | grep -v amdgpu.ids | tail -8 ) > $O/bo_trace.txt 2>&1
