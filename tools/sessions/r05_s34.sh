#!/bin/bash
# round 5, session 34: PMC counters again on the HEAD sources (winograd.hip changed: nontemporal M loads), then the final bench line + rocprofv3 kernel stats + the config table
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r5s34; mkdir -p $R/$O; export PYTHONUNBUFFERED=1
ARGS=""
: > $R/$O/pmc_gemm.txt
for L in conv2_1 conv3_2 conv4_2 conv5_1 roi_c1; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
     --output-format csv -d $R/$O/pmc_$L -- python $R/tools/bench_layers.py --only $L --iters 6 > $R/$O/pmc_$L.log 2>&1
  f=$(find $R/$O/pmc_$L -name '*counter_collection.csv' | head -1)
  cp $f $R/$O/cc_$L.csv
  echo "== $L (rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -- python tools/bench_layers.py --only $L --iters 6)" >> $R/$O/pmc_gemm.txt
  python $R/tools/pmc_summary.py $R/$O/cc_$L.csv >> $R/$O/pmc_gemm.txt 2>&1
  ARGS="$ARGS $L=$R/$O/cc_$L.csv"
  rm -rf $R/$O/pmc_$L
done
cd $R; python tools/pmc_mfma.py $O/mfma_busy.json $ARGS > $O/mfma_busy.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/tf -- python $R/tools/bench_layers.py --only conv4_2 --iters 6 > $R/$O/tf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/tw -- python $R/tools/bench_layers.py --only conv4_2 --iters 6 > $R/$O/tw.log 2>&1
cd $R
python tools/pmc_traffic.py $(find $O/tf -name '*counter_collection.csv' | head -1) $(find $O/tw -name '*counter_collection.csv' | head -1) $O/traffic_wgemm.json f4 1120 > $O/traffic.log 2>&1
rm -rf $O/tf $O/tw $O/cc_*.csv $O/pmc_*.log
cp $O/mfma_busy.json profiles/r05_mfma_busy.json; cp $O/traffic_wgemm.json profiles/r05_traffic_wgemm.json
timeout 400 python bench.py --layers > $O/bench_final.json 2> $O/layers_final.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-robust > $O/bench_prof.json 2> $O/bench_prof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
python tools/kstats.py $O/kernel_stats.csv > $O/kernel_stats_summary.txt 2>&1
: > $O/models.jsonl
for m in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 300 python bench.py --model $m --steps 30 --warmup 8 --no-robust >> $O/models.jsonl 2>> $O/models.err
done
timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --steps 30 --warmup 8 >> $O/models.jsonl 2>> $O/models.err
for dt in f32 f16; do for b in 2 4 8; do
  timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype $dt --batch $b --steps 30 --warmup 8 >> $O/models.jsonl 2>> $O/models.err
done; done
timeout 300 python bench.py --batch 2 --steps 20 --warmup 5 >> $O/models.jsonl 2>> $O/models.err
python tools/models_table.py $O/bench_final.json $O/models.jsonl > $O/models.txt 2>&1
