#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s41; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "c3 or test_conv or kernel_selection or fused_pool" 2>&1 | tail -6 ) > $O/ops.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_net.py -q -x -m "gpu and not slow" -k "layerwise or unfused or partial or cascade_style" 2>&1 | tail -6 ) > $O/net.txt 2>&1
( timeout 600 python bench.py --no-alt --no-robust --no-cpu-baseline --layers 2>$O/bench.err | tail -1 ) > $O/bench.json
