#!/bin/bash
# round 5, session 1: the hand-off health tests (kernel + Net), the zero-frame check, config 5 at full size in fp16 (caltech + CityPersons 640x480) against oracle/_ref, a first bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s1; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_dist.py -m gpu -q -x -k "handoff or zero_warmup or plane_gemm or head_stream_k or full_size_f16 or citypersons_640 or test_gpu_dist or relu or unfused or inner_product_on" 2>&1 | tail -25 ) > $O/tests.txt 2>&1
timeout 300 python bench.py --steps 40 --warmup 10 --no-robust > $O/bench.json 2> $O/bench.err
tail -5 $O/bench.err >> $O/tests.txt
