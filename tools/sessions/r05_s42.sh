#!/bin/bash
# round 5, session 42: cls_pred / bbox_pred kernel at 8 rows per workgroup (A/B + tests), BoxOutput's finish inside the scan kernel, the frame
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s42; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q -x -k "inner_product or boxoutput or caffe_net_small or whole_net_batch or detections" 2>&1 | tail -5 ) > $O/tests.txt 2>&1
timeout 200 python tools/debug/ip_rows_ab.py > $O/ip_rows.txt 2>&1
timeout 200 python bench.py --steps 100 --warmup 10 --no-robust --no-cpu-baseline --layers > $O/bench.json 2> $O/layers.txt
timeout 200 python bench.py --steps 100 --warmup 10 --no-robust --no-cpu-baseline > $O/bench2.json 2> /dev/null
