#!/bin/bash
# round 4, session 48: rocprofv3 kernel stats of the bench at HEAD (conv1_2 on the ring kernel)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s48; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-robust > $O/bench_prof.json 2> $O/bench_prof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
