#!/bin/bash
# round 4, session 41 (final): the whole GPU suite + smoke at HEAD, then the default bench (+ per-layer tables) and its rocprofv3 kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s41; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/gpu_all.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
timeout 900 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-robust > $O/bench_prof.json 2> $O/bench_prof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
