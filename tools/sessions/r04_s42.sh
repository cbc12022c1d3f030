#!/bin/bash
# round 4, session 42 (final, after winograd.hip's last change): PMC traffic of the dominant kernel (separate passes), then the default bench with it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s42; mkdir -p $O; export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/tf -- python $R/tools/bench_layers.py --only conv4_2 --iters 6 > $R/$O/tf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/tw -- python $R/tools/bench_layers.py --only conv4_2 --iters 6 > $R/$O/tw.log 2>&1
cd $R
python tools/pmc_traffic.py $(find $O/tf -name '*counter_collection.csv' | head -1) $(find $O/tw -name '*counter_collection.csv' | head -1) $O/traffic_wgemm.json f4 1120 > $O/traffic.log 2>&1
cp $O/traffic_wgemm.json profiles/r04_traffic_wgemm.json
rm -rf $O/tf $O/tw
timeout 900 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
