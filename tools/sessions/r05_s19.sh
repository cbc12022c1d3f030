#!/bin/bash
# round 5, session 19: the small proposal heads on side streams: bit-identity test, net tests, A/B in one process, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s19; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -k "concurrent or unfused or partial_forward or handoff or dynamic_roi or layerwise or cascade_deploys or batch_n" 2>&1 | tail -15 ) > $O/tests.txt 2>&1
MSCNN_NO_SIDE_STREAMS=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline > $O/bench_seq.json 2> $O/bench_seq.err
timeout 300 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline > $O/bench_side.json 2> $O/bench_side.err
MSCNN_NO_SIDE_STREAMS=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline >> $O/bench_seq.json 2>> $O/bench_seq.err
timeout 300 python bench.py --steps 60 --warmup 10 --no-robust --no-cpu-baseline >> $O/bench_side.json 2>> $O/bench_side.err
