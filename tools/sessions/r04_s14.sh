#!/bin/bash
# round 4, session 14: sliding-maximum maps prefetched on a side stream under BoxOutput; tests + bench + kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s14; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -q -x -k "roipool or inner_product_on or deferred or unfused or partial_forward or layerwise or dynamic_roi or default_flow or test_net_" 2>&1 | tail -8 ) > $O/tests.txt 2>&1
timeout 600 python bench.py --layers --no-alt --no-robust > $O/bench.json 2> $O/bench_layers.txt
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-robust > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err; cd $GRAFT_REPO_ROOT
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/stats
