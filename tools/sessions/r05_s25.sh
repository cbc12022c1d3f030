#!/bin/bash
# round 5, session 25: final evidence at HEAD (garbage collector out of the timed loops): the default bench (+ per-layer tables), rocprofv3 kernel stats of the same command, every config + the batch table, PMC counters of conv1_2's kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s25; mkdir -p $O; export PYTHONUNBUFFERED=1; R=$GRAFT_REPO_ROOT
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
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
   --output-format csv -d $R/$O/pmc_c12 -- python $R/tools/bench_layers.py --only conv1_2 --iters 6 --pool only > $R/$O/pmc_c12.log 2>&1
f=$(find $R/$O/pmc_c12 -name '*counter_collection.csv' | head -1); cp $f $R/$O/cc_conv1_2.csv; rm -rf $R/$O/pmc_c12
cd $R; python tools/pmc_summary.py $O/cc_conv1_2.csv wf2conv > $O/pmc_conv1_2.txt 2>&1
python tools/pmc_mfma.py $O/mfma_busy_conv1_2.json conv1_2=$O/cc_conv1_2.csv > $O/mfma_busy_conv1_2.log 2>&1; rm -f $O/cc_conv1_2.csv
