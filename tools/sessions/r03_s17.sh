#!/bin/bash
# round 3: profiles at HEAD (kernel stats of the bench, PMC of the wgemm kernel, HBM traffic) + F(4x4) A/B on conv3_1 / conv4_1 / conv5_1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s17; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 300 python tools/bench_layers.py --ab algo=3,6 --only conv3_1 --iters 100; timeout 300 python tools/bench_layers.py --ab algo=3,6 --only conv4_1 --iters 100; timeout 300 python tools/bench_layers.py --ab algo=3,6 --only conv5_1 --iters 100 ) > $O/ab_f4_more.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-robust > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; python tools/kstats.py $f > $O/kernel_stats_summary.txt 2>&1; rm -rf $O/stats
for L in conv3_2 conv4_2 conv5_1; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
     --output-format csv -d $O/pmc_$L -- python tools/bench_layers.py --only $L --iters 8 > $O/pmc_$L.log 2>&1
  f=$(find $O/pmc_$L -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f > $O/pmc_sq_$L.txt 2>&1; rm -rf $O/pmc_$L
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tf -- python tools/bench_layers.py --only conv4_2 --iters 6 > $O/tf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tw -- python tools/bench_layers.py --only conv4_2 --iters 6 > $O/tw.log 2>&1
python tools/pmc_traffic.py $(find $O/tf -name '*counter_collection.csv' | head -1) $(find $O/tw -name '*counter_collection.csv' | head -1) $O/traffic_wgemm.json f4 > $O/traffic.log 2>&1
rm -rf $O/tf $O/tw
