#!/bin/bash
# round 5, session 2: the zero-frame check again, config 5 at full size in fp16 (caltech + CityPersons 640x480) against oracle/_ref, calibration / watch / default-flow tests after the re-arming change
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5s2; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q -k "handoff or zero_warmup or full_size_f16 or citypersons_640 or calibration or watch or default_flow or dynamic_roi or chains or deferred" 2>&1 | tail -40 ) > $O/tests.txt 2>&1
