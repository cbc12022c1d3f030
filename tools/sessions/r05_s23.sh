#!/bin/bash
# round 5, session 24: the same with the garbage collector off inside the timed loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s23; mkdir -p $O; export PYTHONUNBUFFERED=1
for i in 1 2; do timeout 200 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --batch 4 --steps 30 --warmup 8 --dump-steps 2>&1 | grep -E "step_ms|images/sec" | cut -c1-400 >> $O/spike.txt; done
timeout 200 python bench.py --model caltech/mscnn-7s-480 --dtype f16 --batch 4 --steps 30 --warmup 20 --dump-steps 2>&1 | grep -E "step_ms|images/sec" | cut -c1-400 >> $O/spike.txt
