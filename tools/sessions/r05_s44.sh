#!/bin/bash
# round 5, session 44: final evidence at HEAD -- the whole GPU suite + smoke, the default bench line with the per-layer table, its rocprofv3 kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s44; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1100 python -m pytest tests/ -q -m gpu 2>&1 | tail -25 ) > $O/gputests.txt 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) >> $O/gputests.txt 2>&1
timeout 300 python bench.py --layers > $O/bench_final.json 2> $O/layers_final.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-robust > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
python tools/kstats.py $O/kernel_stats.csv > $O/kernel_stats_summary.txt 2>&1
