#!/bin/bash
# round 6, session 7: BoxOutput's row count through host-coherent memory (no in-stream copy), the pipelined final stage (mscnn_net_detect_begin / _end): tests, the line, the gaps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s7; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py tests/test_gpu_dist.py -m gpu -q -x -k "detect or boxoutput or watch or roi or deferred or handoff or dynamic or batch_n or caffe_net_small or layerwise or default_flow or boundary or dist or gather or launcher" 2>&1 | tail -12 ) > $O/tests.txt 2>&1
timeout 500 python bench.py --layers > $O/bench.json 2> $O/layers.txt
timeout 200 python bench.py --steps 200 --warmup 20 --no-robust --no-cpu-baseline --no-regimes --detect-mode sync > $O/bench_sync.json 2>> $O/ab.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-robust --no-regimes > /dev/null 2> $GRAFT_REPO_ROOT/$O/kt.err
cd $GRAFT_REPO_ROOT; find $O/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/kgaps.py {} 3 > $O/kernel_gaps.txt 2>&1; rm -rf $O/kt
