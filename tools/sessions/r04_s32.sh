#!/bin/bash
# round 4, session 32: final measurements at HEAD: default bench (+ per-layer tables), rocprofv3 kernel stats of the same command, PMC traffic of the
# dominant kernel (separate passes), the other configs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s32; mkdir -p $O; export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/tf -- python $R/tools/bench_layers.py --only conv4_2 --iters 6 > $R/$O/tf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/tw -- python $R/tools/bench_layers.py --only conv4_2 --iters 6 > $R/$O/tw.log 2>&1
cd $R
python tools/pmc_traffic.py $(find $O/tf -name '*counter_collection.csv' | head -1) $(find $O/tw -name '*counter_collection.csv' | head -1) $O/traffic_wgemm.json f4 1120 > $O/traffic.log 2>&1
cp $O/traffic_wgemm.json profiles/r04_traffic_wgemm.json
rm -rf $O/tf $O/tw
timeout 900 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-robust > $O/bench_prof.json 2> $O/bench_prof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-alt --no-robust > $O/bench_plain2.json 2> $O/bench_plain2.err
for M in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 600 python bench.py --model $M --steps 30 --warmup 8 --no-robust --no-alt > $O/bench_$(basename $M).json 2> $O/bench_$(basename $M).err
done
