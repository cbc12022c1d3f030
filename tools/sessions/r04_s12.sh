#!/bin/bash
# round 4, session 12: InnerProduct on the plane-GEMM kernel (op test, in-net), the F(4x4) test fix, PMC traffic of the wgemm kernel,
# the measured 1-core CPU row, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r4s12; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -q -x -s -k "inner_product or winograd_f4x4 or layerwise or unfused or test_net_ or deferred" 2>&1 | grep -v "^F(4x4\|^ip x3\|^$" | tail -30 ) > $O/tests.txt 2>&1
timeout 200 python tools/cpu_one_core.py --out $O/cpu_one_core.json > $O/cpu_one_core.txt 2>&1 &
timeout 600 python bench.py --layers --no-alt --no-robust > $O/bench.json 2> $O/bench_layers.txt
wait
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/tf -- python $GRAFT_REPO_ROOT/tools/bench_layers.py --only conv4_2 --iters 6 > $GRAFT_REPO_ROOT/$O/tf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/tw -- python $GRAFT_REPO_ROOT/tools/bench_layers.py --only conv4_2 --iters 6 > $GRAFT_REPO_ROOT/$O/tw.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $(find $O/tf -name '*counter_collection.csv' | head -1) $(find $O/tw -name '*counter_collection.csv' | head -1) $O/traffic_wgemm.json f4 1120 > $O/traffic.log 2>&1
rm -rf $O/tf $O/tw
