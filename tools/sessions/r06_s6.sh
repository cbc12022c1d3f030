#!/bin/bash
# round 6, session 6: ROI maps built under BoxOutput's host round trip + the watch's keep-top form: their tests, the line, pin / no-pin A/B, the gaps again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s6; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py -m gpu -q -x -k "watch or roi or deferred or chain or handoff or dynamic or batch_n or caffe_net_small or layerwise or default_flow or boundary or partial or range or materiali" 2>&1 | tail -12 ) > $O/tests.txt 2>&1
A="--steps 200 --warmup 20 --no-robust --no-cpu-baseline --no-regimes"
: > $O/ab.jsonl
for i in 1 2; do
  timeout 200 python bench.py $A >> $O/ab.jsonl 2>> $O/ab.err
  timeout 200 python bench.py $A --no-pin >> $O/ab.jsonl 2>> $O/ab.err
done
timeout 500 python bench.py --layers > $O/bench.json 2> $O/layers.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-robust --no-regimes > /dev/null 2> $GRAFT_REPO_ROOT/$O/kt.err
cd $GRAFT_REPO_ROOT; find $O/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/kgaps.py {} 3 > $O/kernel_gaps.txt 2>&1; rm -rf $O/kt
