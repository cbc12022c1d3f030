#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s24; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "head or inner_product or boxoutput or nms or roi or wgemm or kernel_selection or test_conv" 2>&1 | tail -8 ) > $O/ops.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_net.py -q -x -k "watch or calibration or layerwise or dynamic_roi" 2>&1 | tail -8 ) > $O/net.txt 2>&1
( timeout 300 python -m pytest tests/test_gpu_dist.py -q -x -k world1 2>&1 | tail -5 ) > $O/dist.txt 2>&1
( timeout 600 python bench.py --no-alt --no-robust --no-cpu-baseline --layers 2>$O/bench.err | tail -1 ) > $O/bench.json
