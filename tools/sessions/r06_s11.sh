#!/bin/bash
# round 6, session 11: the whole GPU suite + smoke at HEAD (host round-trip changes, fc6's fall-back under whole tiles), the default line, the rocprofv3 kernel table of the bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s11; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 3000 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 ) > $O/gputests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gputests.txt 2>&1
timeout 500 python bench.py --layers > $O/bench.json 2> $O/layers.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-robust --no-regimes > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT; find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/kgaps.py {} 3 > $O/kernel_gaps.txt 2>&1; rm -rf $O/stats
python tools/kstats.py $O/kernel_stats.csv > $O/kernel_stats_summary.txt 2>&1
