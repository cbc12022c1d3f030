#!/bin/bash
# round 6, session 2: the line with host placement + the max_rois regime; hand-off restart / recover changes (net.cpp); full-size parity with the
# box-coordinate gate; the dist tests on hardware
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r6s2; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 500 python bench.py --layers > $O/bench.json 2> $O/layers.txt
( timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_dist.py -m gpu -q -k "handoff or dist or gather or caffe_net_small or partial or range" 2>&1 | tail -8 ) > $O/tests.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_net.py -m gpu -q -k "test_full_size_parity_vs_reference" -s 2>&1 | grep -E "FULLSIZE|passed|failed|Error" | tail -20 ) >> $O/tests.txt 2>&1
lscpu | grep -E "Model name|Socket|NUMA|^CPU\(s\)" > $O/host.txt; cat /sys/class/drm/card*/device/numa_node >> $O/host.txt 2>/dev/null
