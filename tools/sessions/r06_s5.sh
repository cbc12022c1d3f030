#!/bin/bash
# round 6, session 5: the round's profiles at HEAD: rocprofv3 kernel table of the bench command, PMC MFMA-busy per layer and HBM traffic of conv4_2's plane GEMM
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r6s5; mkdir -p $R/$O; export PYTHONUNBUFFERED=1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-robust --no-regimes > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R; find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/kgaps.py {} 3 > $O/kernel_gaps.txt 2>&1; rm -rf $O/stats
python tools/kstats.py $O/kernel_stats.csv > $O/kernel_stats_summary.txt 2>&1
cd /tmp
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
rm -rf $O/tf $O/tw $O/cc_*.csv
# the other configs at HEAD (GPU legs only)
: > $O/models.jsonl
A="--steps 40 --warmup 8 --no-robust --no-cpu-baseline --no-regimes"
for m in kitti_car/mscnn-8s-768-trainval kitti_ped_cyc/mscnn-7s-576-2x caltech/mscnn-7s-480; do
  timeout 200 python bench.py --model $m $A >> $O/models.jsonl 2>> $O/models.err
done
timeout 200 python bench.py --model caltech/mscnn-7s-480 --dtype f16 $A >> $O/models.jsonl 2>> $O/models.err
timeout 200 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --batch 8 $A >> $O/models.jsonl 2>> $O/models.err
python tools/models_table.py $O/models.jsonl > $O/models.txt 2>&1
