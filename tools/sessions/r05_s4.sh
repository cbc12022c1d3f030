#!/bin/bash
# round 5, session 4: PMC counters of the HEAD plane-GEMM kernel (MFMA busy / waits per layer; HBM traffic of conv4_2), fp16 batch rows of the models table
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r5s4; mkdir -p $R/$O; export PYTHONUNBUFFERED=1
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
: > $O/batch.jsonl
for b in 2 4 8; do
  timeout 300 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --batch $b --steps 30 --warmup 8 >> $O/batch.jsonl 2>> $O/batch.err
done
python tools/models_table.py $O/batch.jsonl > $O/models_f16_batch.txt 2>&1
