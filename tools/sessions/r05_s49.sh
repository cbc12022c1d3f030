#!/bin/bash
# round 5, session 49: the rocprofv3 kernel table once more at HEAD (LFCN_1_5x5 on half chunks), short: whatever GPU time the round has left
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r5s49; mkdir -p $R/$O; export PYTHONUNBUFFERED=1
timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python $R/bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-robust > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err
cd $R; find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
python tools/kstats.py $O/kernel_stats.csv > $O/kernel_stats_summary.txt 2>&1
