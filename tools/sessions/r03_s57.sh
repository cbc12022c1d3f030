#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s57; mkdir -p $O; export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_net.py -q -k "fingerprints" 2>&1 | tail -5 ) > $O/fp.txt 2>&1
