#!/bin/bash
# round 5, session 36: the four ROI-pooling maps in one pass: bit-identity test, the pooling tests, timing (tools/bench_roipool.py + the net)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s36; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "roipool" 2>&1 | tail -5 ) > $O/tests.txt 2>&1
timeout 200 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline --layers 2> $O/layers.txt | cut -c1-140 >> $O/tests.txt
grep -E "^roi_c1" $O/layers.txt >> $O/tests.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-robust > /dev/null 2> $O/prof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
python tools/kstats.py $O/kernel_stats.csv | grep -E "maps_fused|sliding_max|nchw_to_nhwc|roipool_wino33" >> $O/tests.txt
